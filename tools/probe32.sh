#!/bin/bash
# what bounds layer 1 of the 32-row forward kernel: probe 1 = no MMA (DMA + barriers only), 2 = no DMA refills (compute only)
for p in 0 1 2 3; do
RECNN_MLP_PROBE=$p timeout 200 python bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('probe $p: mlp_fwd_nets %.2f us' % ([l['ms'] for l in j['step_breakdown']['launches'] if l['name']=='mlp_fwd_nets'][0]*1e3))"
done
