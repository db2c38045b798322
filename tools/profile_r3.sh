#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_r3.sh TAG
# Round-3 evidence run: the whole GPU suite, smoke(), the driver's bench line (--gpus 1 --steps 20 --warmup 5, complete: PMC child
# runs + CPU baseline) and the default one, rocprofv3 kernel stats of both schedules, forced data parallel with the device
# collective, PMC passes (HBM traffic per kernel; MFMA / CU busy cycles) -> gpurun_out/TAG_*
set -u
T=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/measured_bounds.json
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/${T}_tests.log
tail -3 gpurun_out/${T}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/${T}_smoke.log
cp gpurun_out/measured_bounds.json gpurun_out/${T}_measured_bounds.json 2>/dev/null
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver.json 2> gpurun_out/${T}_bench_driver.err
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 300 python bench.py --force-dp --no-traffic --no-cpu-baseline 2> gpurun_out/${T}_bench_dp1.err | grep '^{' > gpurun_out/${T}_bench_dp1.json
timeout 300 python bench.py --force-dp --collective rccl --no-traffic --no-cpu-baseline 2> /dev/null | grep '^{' > gpurun_out/${T}_bench_dp1_rccl.json
bash tools/r3_stats.sh ${T}_cycle > gpurun_out/${T}_cycle_stats.txt 2>&1
RECNN_SPLIT_FWD=0 bash tools/r3_stats.sh ${T}_fused > gpurun_out/${T}_fused_stats.txt 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$n
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$n -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 60 --repeats 1 --no-cpu-baseline --no-traffic > /dev/null 2>&1)
  find gpurun_out/pmc_$n -name "*kernel_trace.csv" -delete
done
python tools/pmc_r2.py $T > gpurun_out/${T}_pmc.txt 2>&1
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES
python - <<PY
import json
for n in ("bench_driver", "bench", "bench_dp1", "bench_dp1_rccl"):
    try:
        d = json.loads([l for l in open("gpurun_out/${T}_%s.json" % n) if l.startswith("{")][-1])
        print(n, round(d["value"], 1), "steps/s", round(d["ms_per_step"] * 1e3, 2), "us/step", d.get("schedule"), d["config"]["parallelism"],
              "roofline", d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(n, "FAILED", e)
PY
head -12 gpurun_out/${T}_cycle_stats.txt; head -8 gpurun_out/${T}_fused_stats.txt
