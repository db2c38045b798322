#!/bin/bash
# rocprofv3 kernel stats of the "next" rows (REINFORCE at a 100k catalogue, BCQ replayed from hipGraphs) -> gpurun_out/r03_*
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp
for cfg in "reinforce_100k_bf16:python $R/tools/reinforce_bench.py --dtype bf16" "reinforce_100k_fp32:python $R/tools/reinforce_bench.py --dtype fp32" "bcq_bf16_graphed:python $R/tools/bcq_bench.py --no-cpu --dtype bf16 --graphed"; do
  name=${cfg%%:*}; cmd=${cfg#*:}
  rm -rf /tmp/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p --output-format csv -- $cmd > $R/gpurun_out/r03_$name.json 2>/dev/null
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/r03_${name}_kernel_stats.csv
  tail -1 $R/gpurun_out/r03_$name.json | cut -c1-400
  head -6 "$f" | cut -c1-150
done
