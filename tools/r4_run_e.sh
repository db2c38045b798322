#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "transpose" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_reinforce.py -x -q 2>&1 | tail -5
timeout 200 python tools/reinforce_bench.py --dtype bf16 2>/dev/null | tail -1
timeout 200 python tools/reinforce_bench.py --dtype fp32 2>/dev/null | tail -1
bash tools/reinforce_stats.sh bf16 2>&1 | tail -16
