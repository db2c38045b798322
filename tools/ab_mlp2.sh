#!/bin/bash
export TMPDIR=/tmp
T=$1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_optim.py tests/test_gpu_parity_r2.py -m gpu -q 2>&1 | tail -40 > gpurun_out/${T}_sel_tests.log
grep -E "passed|failed|FAILED|^E  " gpurun_out/${T}_sel_tests.log | tail -20
for v in "32 0 0" "64 0 0" "64 2 0" "64 0 6" "64 0 2" "32 0 0" "64 0 0" "64 2 0"; do
  set -- $v
  RECNN_MLP_PANEL=$1 RECNN_MLP_MAP=$2 RECNN_MLP_PROBE=$3 timeout 200 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic > gpurun_out/${T}_probe.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("gpurun_out/${T}_probe.json").read().strip().splitlines()[-1])
print("panel $1 map $2 probe $3: %.2f us/step  " % (j["ms_per_step"]*1e3), [(l["name"], round(l["ms"]*1e3,2)) for l in j["step_breakdown"]["launches"]][:4])
PY
done
for p in 64 32; do
RECNN_MLP_PANEL=$p timeout 200 python bench.py --algo td3 --rows 4096 --steps 1000 --warmup 100 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('td3 4096 panel $p: %.2f us/step' % (j['ms_per_step']*1e3), [(l['name'], round(l['ms']*1e3,2)) for l in j['step_breakdown']['launches']][:5])"
done
