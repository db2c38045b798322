#!/bin/bash
# round 3, check 2: why is dw_opt_kernel slow?  probes + PMC counters (one pass, SQ counters only)
export TMPDIR=/tmp
T=${1:-r03c}
mkdir -p gpurun_out
for pr in 0 1 2 4 3 6; do
  RECNN_DW_PROBE=$pr timeout 300 python bench.py --steps 300 --warmup 50 --repeats 1 --no-cpu-baseline --no-traffic > gpurun_out/${T}_probe$pr.json 2>gpurun_out/${T}_probe$pr.err
  python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/${T}_probe$pr.json").read().strip().splitlines()[-1])
    print("probe $pr: %.2f us/step" % (j["ms_per_step"]*1e3), [(l["name"], round(l["ms"]*1e3,2)) for l in j["step_breakdown"]["launches"] if "dw" in l["name"]])
except Exception as ex:
    print("probe $pr failed", ex); print(open("gpurun_out/${T}_probe$pr.err").read()[-1500:])
PY
done
cd /tmp
for f in 1 0; do
  RECNN_DW_FUSE=$f timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/pmc$f -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 20 --repeats 1 --no-cpu-baseline --no-traffic > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc$f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        if "dw" in k or "apply" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("fuse $f", k, {c: round(sum(v)/len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
done
