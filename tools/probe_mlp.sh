#!/bin/bash
export TMPDIR=/tmp
T=$1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -120 > gpurun_out/${T}_tests.log
grep -E "passed|failed|FAILED|^E  " gpurun_out/${T}_tests.log | tail -40
for v in "64 0 0" "64 0 1" "64 0 2" "64 0 4" "64 0 6" "64 2 1" "32 0 0"; do
  set -- $v
  RECNN_MLP_PANEL=$1 RECNN_MLP_MAP=$2 RECNN_MLP_PROBE=$3 timeout 200 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-traffic > gpurun_out/${T}_probe.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("gpurun_out/${T}_probe.json").read().strip().splitlines()[-1])
print("panel $1 map $2 probe $3: %.2f us/step  " % (j["ms_per_step"]*1e3), [(l["name"], round(l["ms"]*1e3,2)) for l in j["step_breakdown"]["launches"]][:4])
PY
done
