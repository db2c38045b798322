#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5i; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_x3.py -q -x 2>&1 | tail -3
VARIANTS=2 PROBES=0,512,16,528,1 timeout 120 python tools/x3_fwd_probe.py 2>&1 | grep "as is"
M=2048 VARIANTS=2 PROBES=0,512 timeout 120 python tools/x3_fwd_probe.py 2>&1 | grep "as is"
for rep in 1 2; do for pr in 512 0; do
  RECNN_X3_WS_PROBE=$pr timeout 300 python bench.py --dtype bf16x3 --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic --no-extras > $O/b.json 2>$O/b.err
  python - <<PY
import json
d=json.load(open("$O/b.json")); print("probe $pr", d.get("schedule"), round(d["ms_per_step"]*1000,2), "us", " ".join("%s=%.1f" % (l["name"], l["ms"]*1000) for l in d["step_breakdown"]["launches"]))
PY
done; done
