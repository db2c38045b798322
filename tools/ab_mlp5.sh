#!/bin/bash
export TMPDIR=/tmp
T=$1
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_api.py tests/test_gpu_bench_shape.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/${T}_sel.log
grep -E "passed|failed|FAILED|^E  " gpurun_out/${T}_sel.log | tail -12
timeout 120 python tools/mlp_trace.py 0 0 2>&1 | grep -A14 "== target_actor" | head -16
for v in 0 0 0; do
  RECNN_MLP_KERNEL=$v timeout 200 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-traffic > gpurun_out/${T}_probe.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("gpurun_out/${T}_probe.json").read().strip().splitlines()[-1])
print("kernel $v: %.2f us/step  " % (j["ms_per_step"]*1e3), [(l["name"], round(l["ms"]*1e3,2)) for l in j["step_breakdown"]["launches"]][:4])
PY
done
