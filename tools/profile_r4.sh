#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_r4.sh TAG
# Round-4 evidence run: the whole GPU suite, smoke(), the driver's bench line (--gpus 1 --steps 20 --warmup 5, complete: PMC child
# runs, CPU baseline, parity_mode / other_configs sub-records), the default line, the split-bf16 line, rocprofv3 kernel stats of the
# three schedules (every rocprofv3 call under `timeout`, csv output, kernel trace only) -> gpurun_out/TAG_*
set -u
T=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/measured_bounds.json
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/${T}_tests.log 2>&1
tail -6 gpurun_out/${T}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/${T}_smoke.log
cp gpurun_out/measured_bounds.json gpurun_out/${T}_measured_bounds.json 2>/dev/null
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver.json 2> gpurun_out/${T}_bench_driver.err
timeout 600 python bench.py --no-traffic --no-extras > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 600 python bench.py --dtype bf16x3 --gpus 1 --steps 20 --warmup 5 --no-extras > gpurun_out/${T}_bench_bf16x3_driver.json 2> gpurun_out/${T}_bench_bf16x3_driver.err
bash tools/r3_stats.sh ${T}_cycle > gpurun_out/${T}_cycle_stats.txt 2>&1
RECNN_SPLIT_FWD=0 bash tools/r3_stats.sh ${T}_fused > gpurun_out/${T}_fused_stats.txt 2>&1
BENCH_ARGS="--dtype bf16x3 --no-extras" bash tools/r3_stats.sh ${T}_x3 > gpurun_out/${T}_x3_stats.txt 2>&1
python - <<PY
import json
for n in ("bench_driver", "bench", "bench_bf16x3_driver"):
    try:
        d = json.loads([l for l in open("gpurun_out/${T}_%s.json" % n) if l.startswith("{")][-1])
        print(n, round(d["value"], 1), "steps/s", round(d["ms_per_step"] * 1e3, 2), "us/step", d.get("schedule"), d["dtype"],
              "roofline", d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"), "parity_mode", d.get("parity_mode"))
    except Exception as e:
        print(n, "FAILED", e)
PY
head -10 gpurun_out/${T}_cycle_stats.txt; head -8 gpurun_out/${T}_fused_stats.txt; head -10 gpurun_out/${T}_x3_stats.txt
