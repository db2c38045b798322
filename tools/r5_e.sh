#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5e; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_engine.py tests/test_gpu_bench_shape.py tests/test_gpu_api.py tests/test_gpu_parity_r2.py -q -x -k "x3 or bf16x3 or tail" 2>&1 | tail -8 ) | tee $O/tests.log
for rep in 1 2; do
for hf in 0 1; do
  RECNN_X3_HEAD_DX=$hf timeout 300 python bench.py --dtype bf16x3 --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic --no-extras > $O/bench_hd${hf}_$rep.json 2>$O/err.txt
  python - <<PY
import json
d=json.load(open("$O/bench_hd${hf}_$rep.json"))
print("head_dx=$hf rep $rep: 2000 steps", round(d["ms_per_step"]*1000,1), "us", [round(x*1000,1) for x in d["ms_per_step_samples"]], " ".join("%s=%.1f" % (l["name"], l["ms"]*1000) for l in d["step_breakdown"]["launches"]))
PY
done; done 2>&1 | tee $O/ab_head_dx.log
tail -3 $O/err.txt
