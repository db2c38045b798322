#!/bin/bash
# SQ / LDS counters of the split-bf16 step's kernels (one --pmc pass per group, kernel trace only)
export TMPDIR=/tmp
cd /tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
timeout 60 rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*\|TCP_[A-Z_0-9]*" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/avail_counters.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 240 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --repeats 1 --no-cpu-baseline --no-traffic --no-extras $BENCH_ARGS > /tmp/pmc_$i.json 2> /tmp/pmc_$i.err
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] || { echo "group $i: no counter file"; tail -5 /tmp/pmc_$i.err; continue; }
  python - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$f")):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    n = len(next(iter(cs.values())))
    if n < 30: continue
    print(k[:64].ljust(64), "n=%d" % n)
    print("   ", "  ".join("%s=%.3g" % (c.replace("SQ_", ""), sum(v) / len(v)) for c, v in cs.items()))
PY
done
