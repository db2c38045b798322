#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c; mkdir -p $O
for rep in 1 2; do
for v in 0 11 2; do
  RECNN_X3_FWD=$v timeout 300 python bench.py --dtype bf16x3 --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic --no-extras > $O/bench_v${v}_$rep.json 2>$O/err.txt
  python - <<PY
import json
d=json.load(open("$O/bench_v${v}_$rep.json"))
print("x3_fwd=$v rep $rep: 2000 steps", round(d["ms_per_step"]*1000,1), "us", [round(x*1000,1) for x in d["ms_per_step_samples"]])
PY
done; done 2>&1 | tee $O/ab2.log
