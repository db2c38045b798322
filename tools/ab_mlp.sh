#!/bin/bash
# A/B of the fused MLP forward variants inside one box (numbers differ by up to 25 % between boxes)
export TMPDIR=/tmp
T=$1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "mlp_forward_64" 2>&1 | tail -30 > gpurun_out/${T}_mlp64_test.log
cat gpurun_out/${T}_mlp64_test.log | tail -12
for v in "32 0" "64 0" "64 2" "32 0" "64 0" "64 2"; do
  set -- $v
  RECNN_MLP_PANEL=$1 RECNN_MLP_MAP=$2 timeout 200 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-traffic > gpurun_out/${T}_ab_p$1_m$2.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("gpurun_out/${T}_ab_p$1_m$2.json").read().strip().splitlines()[-1])
print("panel $1 map $2: %.2f us/step  " % (j["ms_per_step"]*1e3), [(l["name"], round(l["ms"]*1e3,2)) for l in j["step_breakdown"]["launches"]][:4])
PY
done
RECNN_MLP_PANEL=64 timeout 200 python bench.py --algo td3 --rows 4096 --steps 1000 --warmup 100 --no-cpu-baseline --no-traffic 2>/dev/null | tail -c 900 | head -c 400; echo
RECNN_MLP_PANEL=32 timeout 200 python bench.py --algo td3 --rows 4096 --steps 1000 --warmup 100 --no-cpu-baseline --no-traffic 2>/dev/null | tail -c 900 | head -c 400; echo
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -150 > gpurun_out/${T}_tests.log
grep -E "passed|failed|FAILED|Error|assert" gpurun_out/${T}_tests.log | tail -30
