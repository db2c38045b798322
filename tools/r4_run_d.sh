#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_optim.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_reinforce.py tests/test_gpu_bcq.py tests/test_gpu_sac.py tests/test_gpu_api.py -x -q 2>&1 | tail -5
timeout 200 python tools/reinforce_bench.py --dtype bf16 2>/dev/null | tail -1
timeout 200 python tools/reinforce_bench.py --dtype fp32 2>/dev/null | tail -1
timeout 300 python tools/reinforce_ops.py 2>&1 | grep "aten::\|total device" | head -12
