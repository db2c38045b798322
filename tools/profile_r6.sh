#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_r6.sh [part ...]     parts: bench stats pmc trace td3 reinforce suite (default: all but suite)
# Round-6 evidence run -> gpurun_out/r06_* (copy what is to be judged into profiles/).  Every rocprofv3 call: kernel trace only, csv,
# under `timeout`; counters in their own passes.
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=r06
mkdir -p gpurun_out
PARTS=${*:-bench stats pmc trace td3 reinforce}
has() { [[ " $PARTS " == *" $1 "* ]]; }

if has suite; then
  rm -f gpurun_out/measured_bounds.json
  ( time timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > gpurun_out/${T}_tests.log 2>&1
  tail -6 gpurun_out/${T}_tests.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/${T}_smoke.log
  cp gpurun_out/measured_bounds.json gpurun_out/${T}_measured_bounds.json 2>/dev/null
fi
if has bench; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver.json 2> gpurun_out/${T}_bench_driver.err
  timeout 900 python bench.py --no-traffic --no-extras > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
  timeout 600 python bench.py --dtype bf16x3 --gpus 1 --steps 20 --warmup 5 --no-extras --no-traffic > gpurun_out/${T}_bench_bf16x3_driver.json 2> /dev/null
  python - <<PY
import json
for n in ("bench_driver", "bench", "bench_bf16x3_driver"):
    try:
        d = json.loads([l for l in open("gpurun_out/${T}_%s.json" % n) if l.startswith("{")][-1])
        pm = d.get("parity_mode")
        r = d.get("roofline", {})
        print(n, round(d["value"], 1), "steps/s", round(d["ms_per_step"] * 1e3, 2), "us/step", d.get("schedule"), d["dtype"], "sustained",
              round((d.get("sustained") or {}).get("value", 0)), "| roofline", r.get("kernel"), round(r.get("frac") or 0, 4), "share",
              round(r.get("share_of_step_time") or 0, 3), "| e2e", round(d["roofline_end_to_end"]["frac"], 4),
              "| parity_mode", (round(pm["value"]), pm["loss_curve_deviation"]["within_bound"]) if pm else None)
    except Exception as e:
        print(n, "FAILED", e)
PY
fi
if has stats; then
  bash tools/r3_stats.sh ${T}_cycle > gpurun_out/${T}_cycle_stats.txt 2>&1
  RECNN_SPLIT_FWD=0 bash tools/r3_stats.sh ${T}_fused > gpurun_out/${T}_fused_stats.txt 2>&1
  RECNN_DW_FUSE=0 RECNN_TAIL_HALF=0 bash tools/r3_stats.sh ${T}_cycle_r5form > gpurun_out/${T}_cycle_r5form_stats.txt 2>&1
  BENCH_ARGS="--dtype bf16x3 --no-extras" bash tools/r3_stats.sh ${T}_x3 > gpurun_out/${T}_x3_stats.txt 2>&1
  head -12 gpurun_out/${T}_cycle_stats.txt; head -8 gpurun_out/${T}_cycle_r5form_stats.txt; head -6 gpurun_out/${T}_fused_stats.txt; head -8 gpurun_out/${T}_x3_stats.txt
fi
if has pmc; then
  rm -rf gpurun_out/pmc_*
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d' ' -f1)
    (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$n -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 60 --repeats 1 --child-trace --no-cpu-baseline --no-traffic --no-extras > /dev/null 2>&1)
    find gpurun_out/pmc_$n -name "*kernel_trace.csv" -delete
  done
  python tools/pmc_r2.py ${T} > gpurun_out/${T}_pmc.txt 2>&1
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES
  cut -c1-320 gpurun_out/${T}_pmc.txt | head -14
fi
if has trace; then
  timeout 200 python tools/chain_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_chain_trace.txt
  cat gpurun_out/${T}_chain_trace.txt
fi
if has td3; then
  timeout 600 python bench.py --algo td3 --rows 4096 --steps 20 --warmup 5 --no-extras --no-traffic --no-cpu-baseline > gpurun_out/${T}_bench_td3_b4096.json 2> /dev/null
  timeout 600 python bench.py --algo td3 --rows 4096 --steps 2000 --warmup 200 --repeats 3 --no-extras --no-traffic --no-cpu-baseline > gpurun_out/${T}_bench_td3_b4096_2000.json 2> /dev/null
  python - <<PY
import json
for n in ("bench_td3_b4096", "bench_td3_b4096_2000"):
    try:
        d = json.loads([l for l in open("gpurun_out/${T}_%s.json" % n) if l.startswith("{")][-1])
        print(n, round(d["value"], 1), "steps/s", round(d["ms_per_step"] * 1e3, 2), "us/step e2e", round(d["roofline_end_to_end"]["frac"], 4))
    except Exception as e:
        print(n, "FAILED", e)
PY
fi
if has reinforce; then
  timeout 300 python tools/reinforce_bench.py --dtype bf16 | tee gpurun_out/${T}_reinforce_100k_bf16.json | cut -c1-400
fi
