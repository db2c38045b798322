#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5d
for v in 0 2; do
RECNN_X3_FWD=$v BENCH_ARGS="--dtype bf16x3" bash tools/r3_stats.sh r5d_x3_v$v > gpurun_out/r5d/stats_v$v.txt 2>&1
cat gpurun_out/r5d/stats_v$v.txt
done
