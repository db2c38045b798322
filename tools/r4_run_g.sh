#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver.json 2> gpurun_out/r04_bench_driver.err ) 2>&1 | grep real
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r04_bench_driver.json") if l.startswith("{")][-1])
print(round(d["value"], 1), round(d["ms_per_step"] * 1e3, 2), d.get("extras_error"))
print(json.dumps(d.get("other_configs"))[:900])
print(json.dumps(d.get("parity_mode"))[:300])
PY
tail -3 gpurun_out/r04_bench_driver.err
