#!/bin/bash
# clean kernel stats of the two bf16 schedules (no sub-records in the profiled run) + kernel stats of the 100k-item REINFORCE step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/r3_stats.sh r04_cycle > gpurun_out/r04_cycle_stats.txt 2>&1
RECNN_SPLIT_FWD=0 bash tools/r3_stats.sh r04_fused > gpurun_out/r04_fused_stats.txt 2>&1
cd /tmp && rm -rf /tmp/prof_rf
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_rf -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/reinforce_bench.py --dtype bf16 > /tmp/prof_rf.json 2>/dev/null
f=$(find /tmp/prof_rf -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r04_reinforce_100k_bf16_kernel_stats.csv
cd $GRAFT_REPO_ROOT
tail -1 /tmp/prof_rf.json
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/r04_reinforce_100k_bf16_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
aten = sum(float(r["TotalDurationNs"]) for r in rows if "at::native" in r["Name"] or r["Name"].startswith("Cijk"))
print("ATen + Tensile share of GPU time: %.1f %%" % (100 * aten / tot))
for r in rows[:22]:
    print("%-100s calls %5s avg %8.1f us %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
head -12 gpurun_out/r04_cycle_stats.txt; head -12 gpurun_out/r04_fused_stats.txt
