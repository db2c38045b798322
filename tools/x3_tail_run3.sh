#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_x3.py tests/test_gpu_bench_shape.py -q -k "row_panel or bf16x3" 2>&1 | tail -8 > gpurun_out/x3tail_t3.log
BENCH_ARGS="--dtype bf16x3 --no-extras" bash tools/r3_stats.sh x3tail > gpurun_out/x3tail_stats.txt 2>&1
timeout 200 python bench.py --dtype bf16x3 --steps 200 --warmup 20 --no-extras > gpurun_out/x3tail_bench_1.json 2> gpurun_out/x3tail_bench_1.err
cat gpurun_out/x3tail_t3.log gpurun_out/x3tail_stats.txt
python - <<PY
import json
d = json.loads(open("gpurun_out/x3tail_bench_1.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"])
PY
