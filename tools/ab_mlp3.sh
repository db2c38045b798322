#!/bin/bash
export TMPDIR=/tmp
T=$1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "mlp_forward_64" 2>&1 | tail -25 > gpurun_out/${T}_mlp_test.log
grep -E "passed|failed|FAILED|^E  " gpurun_out/${T}_mlp_test.log | tail -12
timeout 120 python tools/mlp_trace.py 0 2 2>&1 | grep -v "^   setup\|L1 slab" | tail -36
for v in "0 0" "2 0" "2 2" "0 0" "2 0" "2 2"; do
  set -- $v
  RECNN_MLP_KERNEL=$1 RECNN_MLP_MAP=$2 timeout 200 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic > gpurun_out/${T}_probe.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("gpurun_out/${T}_probe.json").read().strip().splitlines()[-1])
print("kernel $1 map $2: %.2f us/step  " % (j["ms_per_step"]*1e3), [(l["name"], round(l["ms"]*1e3,2)) for l in j["step_breakdown"]["launches"]][:4])
PY
done
for k in 2 0; do
RECNN_MLP_KERNEL=$k timeout 200 python bench.py --algo td3 --rows 4096 --steps 1000 --warmup 100 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('td3 4096 kernel $k: %.2f us/step' % (j['ms_per_step']*1e3), [(l['name'], round(l['ms']*1e3,2)) for l in j['step_breakdown']['launches']][:5])"
done
