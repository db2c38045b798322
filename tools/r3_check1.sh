#!/bin/bash
# round 3, check 1: fused dW + optimizer (dwopt.hip) -- parity tests, then A/B bench inside one box
export TMPDIR=/tmp
T=${1:-r03a}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dwopt.py tests/test_gpu_engine.py tests/test_gpu_bench_shape.py tests/test_gpu_api.py tests/test_gpu_parity_r2.py -m gpu -q 2>&1 | tail -40 > gpurun_out/${T}_tests.log
tail -15 gpurun_out/${T}_tests.log
for v in 1 2 3 0 1; do
  RECNN_DW_FUSE=$v timeout 300 python bench.py --steps 2000 --warmup 200 --repeats 3 --no-cpu-baseline --no-traffic > gpurun_out/${T}_fuse$v.json 2>gpurun_out/${T}_fuse$v.err
  python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/${T}_fuse$v.json").read().strip().splitlines()[-1])
    print("fuse $v: %.2f us/step (samples %s)" % (j["ms_per_step"]*1e3, j["ms_per_step_samples"]), [(l["name"], round(l["ms"]*1e3,2)) for l in j["step_breakdown"]["launches"]])
except Exception as ex:
    print("fuse $v failed", ex); print(open("gpurun_out/${T}_fuse$v.err").read()[-2000:])
PY
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > gpurun_out/${T}_driver.json 2>gpurun_out/${T}_driver.err
tail -c 600 gpurun_out/${T}_driver.json
