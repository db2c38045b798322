#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5e; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_engine.py tests/test_gpu_bench_shape.py tests/test_gpu_api.py -q -x -k "x3 or bf16x3 or tail" 2>&1 | tail -6 ) | tee $O/tests.log
for rep in 1 2; do
for fk in 0 1; do
  RECNN_X3_FORK=$fk timeout 300 python bench.py --dtype bf16x3 --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic --no-extras > $O/bench_fk${fk}_$rep.json 2>$O/err.txt
  RECNN_X3_FORK=$fk timeout 300 python bench.py --dtype bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras > $O/drv_fk${fk}_$rep.json 2>>$O/err.txt
  python - <<PY
import json
d=json.load(open("$O/bench_fk${fk}_$rep.json")); e=json.load(open("$O/drv_fk${fk}_$rep.json"))
print("fork=$fk rep $rep: 2000 steps", round(d["ms_per_step"]*1000,1), "us", [round(x*1000,1) for x in d["ms_per_step_samples"]], "driver cmd", round(e["ms_per_step"]*1000,1), [round(x*1000,1) for x in e["ms_per_step_samples"]])
PY
done; done 2>&1 | tee $O/ab.log
tail -3 $O/err.txt
