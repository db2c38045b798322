#!/bin/bash
# rocprofv3 kernel stats of tools/reinforce_bench.py (100k items) and the ATen / Tensile share of its GPU time
# usage: bash tools/reinforce_stats.sh [bf16|fp32]
D=${1:-bf16}
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && rm -rf /tmp/prof_rf
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rf -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/reinforce_bench.py --dtype $D > /tmp/prof_rf.json 2>/dev/null
f=$(find /tmp/prof_rf -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r04_reinforce_100k_${D}_kernel_stats.csv
cd $GRAFT_REPO_ROOT
tail -1 /tmp/prof_rf.json | tee gpurun_out/r04_reinforce_100k_${D}_beta.json
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/r04_reinforce_100k_${D}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
aten = sum(float(r["TotalDurationNs"]) for r in rows if "at::native" in r["Name"] or r["Name"].startswith("Cijk"))
print("ATen + Tensile share of GPU time: %.1f %%" % (100 * aten / tot))
for r in rows[:14]:
    print("%-100s calls %5s avg %8.1f us %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
