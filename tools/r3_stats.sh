#!/bin/bash
# kernel-time budget of one configuration: rocprofv3 --kernel-trace --stats over bench.py (env knobs pass through)
export TMPDIR=/tmp
T=${1:-stats}
mkdir -p gpurun_out
cd /tmp
rm -rf /tmp/prof_$T
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$T -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 4000 --warmup 200 --repeats 1 --no-cpu-baseline --no-traffic --no-extras $BENCH_ARGS > /tmp/prof_$T.json 2>/dev/null
f=$(find /tmp/prof_$T -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${T}_kernel_stats.csv 2>/dev/null
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$f")))
j = json.loads(open("/tmp/prof_$T.json").read().strip().splitlines()[-1])
print("$T: %.2f us/step under rocprof" % (j["ms_per_step"] * 1e3))
steps = 4200.0
tot = 0.0
for r in rows:
    t = float(r["TotalDurationNs"]) / 1e3
    tot += t
    if t / steps > 0.08:
        print("  %-58s calls %6s avg %8.2f us  %6.2f us/step" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, t / steps))
print("  sum of kernel time / step (eager profile launches included): %.2f us" % (tot / steps))
PY
