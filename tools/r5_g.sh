#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5g; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_x3.py -q 2>&1 | tail -3
VARIANTS=2 PROBES=0,256,16 timeout 300 python tools/x3_fwd_probe.py 2>&1 | grep variant
M=2048 VARIANTS=2 PROBES=0,256,16 timeout 300 python tools/x3_fwd_probe.py 2>&1 | grep variant
for rep in 1 2; do for pb in 256 0; do
  RECNN_X3_WS_PROBE=$pb timeout 300 python bench.py --dtype bf16x3 --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic --no-extras > $O/b_$pb_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/b_$pb_$rep.json")); print("probe=$pb (256 = no kernarg prefetch) rep $rep:", round(d["ms_per_step"]*1000,1), "us", " ".join("%s=%.1f" % (l["name"], l["ms"]*1000) for l in d["step_breakdown"]["launches"][:4]))
PY
done; done
