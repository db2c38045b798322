#!/bin/bash
# round 5, GPU call A: the split-bf16 forward GEMM variants (gemm.hip x3_fwd_launch): correctness, the layer-1 shape, the step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5a; mkdir -p $O
for v in 1 2 6; do
  echo "== tests/test_gpu_x3.py variant $v"; RECNN_X3_FWD_DEBUG=$v RECNN_X3_FWD=$v timeout 400 python -m pytest tests/test_gpu_x3.py -q -x 2>&1 | tail -3
done 2>&1 | tee $O/tests.log
echo "== probe M=8192 (the grouped layer-1 launch: 256 tiles of 64 x 128)"
VARIANTS=0,2 OUT=$O/probe_8192.json timeout 300 python tools/x3_fwd_probe.py 2>&1 | grep variant | tee $O/probe_8192.log
echo "== probe M=2048 (one network's layer 1)"
M=2048 VARIANTS=0,2 OUT=$O/probe_2048.json timeout 300 python tools/x3_fwd_probe.py 2>&1 | grep variant | tee $O/probe_2048.log
for v in 0 1 2 5 6 22; do
  RECNN_X3_FWD=$v timeout 300 python bench.py --dtype bf16x3 --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic --no-extras > $O/bench_v$v.json 2>$O/bench_v$v.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_v$v.json"))
    print("x3_fwd=$v", round(d["value"]), "steps/s", round(d["ms_per_step"]*1000,1), "us;", " ".join("%s=%.1f" % (l["name"], l["ms"]*1000) for l in d["step_breakdown"]["launches"]))
except Exception as e:
    print("x3_fwd=$v FAILED", e)
PY
done 2>&1 | tee $O/bench.log
