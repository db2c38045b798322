#!/bin/bash
# what bounds the stream kernel: probe 1 = no MFMA work, 2 = no DMA, 3 = neither (synchronisation skeleton + epilogues)
for p in 0 1 2 3; do
RECNN_MLP_KERNEL=3 RECNN_MLP_PROBE=$p timeout 200 python bench.py --steps 600 --warmup 100 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('probe $p: mlp_fwd_nets %.2f us' % ([l['ms'] for l in j['step_breakdown']['launches'] if l['name']=='mlp_fwd_nets'][0]*1e3))"
done
