#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_reinforce.py -x -q -k "beta or bf16" 2>&1 | tail -4
timeout 200 python tools/reinforce_bench.py --dtype bf16 2>/dev/null | tail -1
