#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_x3.py tests/test_gpu_engine.py tests/test_gpu_kernels.py tests/test_gpu_optim.py -q 2>&1 | tail -3
for rep in 1 2; do for lib in base new; do cp recnn_amd/csrc/${lib}_lib.so recnn_amd/csrc/librecnn_hip.so; echo "== lib $lib rep $rep"; for cfg in "--dtype bf16x3" "" ; do
  timeout 300 python bench.py $cfg --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic --no-extras > $O/b.json 2>/dev/null
  RECNN_SPLIT_FWD=0 timeout 300 python bench.py $cfg --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic --no-extras > $O/bf.json 2>/dev/null
  python - <<PY
import json
for f in ("b","bf"):
    d=json.load(open("$O/%s.json" % f)); print("$cfg", d.get("schedule"), round(d["ms_per_step"]*1000,2), "us", " ".join("%s=%.1f" % (l["name"], l["ms"]*1000) for l in d["step_breakdown"]["launches"]))
PY
done
done; done
cp recnn_amd/csrc/new_lib.so recnn_amd/csrc/librecnn_hip.so
