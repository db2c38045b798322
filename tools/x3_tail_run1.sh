#!/bin/bash
# first GPU pass over the x3 row-panel tail: comparison test, the oracle tests, then the parity-mode bench with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_x3.py -x -q -k "row_panel" 2>&1 | tail -15 > gpurun_out/x3tail_t1.log
timeout 400 python -m pytest tests/test_gpu_engine.py tests/test_gpu_bench_shape.py -x -q -k "bf16x3" 2>&1 | tail -15 > gpurun_out/x3tail_t2.log
for t in 1 0; do
  RECNN_X3_TAIL=$t timeout 200 python bench.py --dtype bf16x3 --steps 200 --warmup 20 --no-extras > gpurun_out/x3tail_bench_$t.json 2> gpurun_out/x3tail_bench_$t.err
done
cat gpurun_out/x3tail_t1.log gpurun_out/x3tail_t2.log
for t in 1 0; do python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/x3tail_bench_$t.json").read().strip().splitlines()[-1])
    print("tail=$t", d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"))
except Exception as ex:
    print("tail=$t failed", ex, open("gpurun_out/x3tail_bench_$t.err").read()[-800:])
PY
done
