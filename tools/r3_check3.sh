#!/bin/bash
# round 3, check 3: cycle mode (hoisted frozen networks + split forward) -- the bench-shape identities, then A/B timing
export TMPDIR=/tmp
T=${1:-r03d}
mkdir -p gpurun_out
RECNN_SPLIT_FWD=2 timeout 900 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_api.py tests/test_gpu_split.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/${T}_tests.log
tail -12 gpurun_out/${T}_tests.log
for v in 1 0 1; do
  RECNN_SPLIT_FWD=$v timeout 300 python bench.py --steps 2000 --warmup 200 --repeats 3 --no-cpu-baseline --no-traffic > gpurun_out/${T}_split$v.json 2>gpurun_out/${T}_split$v.err
  python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/${T}_split$v.json").read().strip().splitlines()[-1])
    print("split $v: %.2f us/step (samples %s)" % (j["ms_per_step"]*1e3, j["ms_per_step_samples"]), [(l["name"], round(l["ms"]*1e3,2)) for l in j["step_breakdown"]["launches"]])
except Exception as ex:
    print("split $v failed", ex); print(open("gpurun_out/${T}_split$v.err").read()[-2000:])
PY
done
cd /tmp
RECNN_SPLIT_FWD=2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2000 --warmup 200 --repeats 1 --no-cpu-baseline --no-traffic > /dev/null 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${T}_kernel_stats.csv 2>/dev/null
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:16]:
    print("%-60s calls %6s avg %8.2f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
