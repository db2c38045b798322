#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_x3.py -x -q 2>&1 | tail -8 > gpurun_out/x3direct_t1.log
cat gpurun_out/x3direct_t1.log
for d in 1 0 2 3; do
  RECNN_X3_DIRECT=$d timeout 200 python bench.py --dtype bf16x3 --steps 200 --warmup 20 --no-extras --no-traffic --no-cpu-baseline > gpurun_out/x3direct_bench_$d.json 2> gpurun_out/x3direct_bench_$d.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/x3direct_bench_$d.json").read().strip().splitlines()[-1])
    print("direct=$d", d["value"], d["ms_per_step"])
except Exception as ex:
    print("direct=$d failed", ex, open("gpurun_out/x3direct_bench_$d.err").read()[-600:])
PY
done
