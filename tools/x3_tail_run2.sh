#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_bench_shape.py -q -k "bf16x3" 2>&1 | tail -15 > gpurun_out/x3tail_t2.log
BENCH_ARGS="--dtype bf16x3 --no-extras" bash tools/r3_stats.sh x3tail > gpurun_out/x3tail_stats.txt 2>&1
cat gpurun_out/x3tail_t2.log gpurun_out/x3tail_stats.txt
