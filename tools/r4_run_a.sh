#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bench_shape.py -x -q -k "td3_b4096_loss_curve" 2>&1 | tail -12 > gpurun_out/r4a_t1.log
timeout 300 python -m pytest tests/test_gpu_engine.py tests/test_gpu_split.py -x -q 2>&1 | tail -5 > gpurun_out/r4a_t2.log
cat gpurun_out/r4a_t1.log gpurun_out/r4a_t2.log
for i in 1 2 3; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-extras --no-traffic --no-cpu-baseline > gpurun_out/r4a_bench_$i.json 2> gpurun_out/r4a_bench_$i.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r4a_bench_$i.json").read().strip().splitlines()[-1])
    print("bench $i", d["value"], d["ms_per_step"])
except Exception as ex:
    print("bench $i failed", ex, open("gpurun_out/r4a_bench_$i.err").read()[-600:])
PY
done
