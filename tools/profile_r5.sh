#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_r5.sh [part ...]     parts: bench stats pmc dma reinforce bcq (default: all)
# Round-5 evidence run -> gpurun_out/r05_* (copy what is to be judged into profiles/).  Every rocprofv3 call: kernel trace only, csv,
# under `timeout`; counters in their own passes.
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=r05
mkdir -p gpurun_out
PARTS=${*:-bench stats pmc dma reinforce bcq}
has() { [[ " $PARTS " == *" $1 "* ]]; }

if has bench; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver.json 2> gpurun_out/${T}_bench_driver.err
  timeout 900 python bench.py --no-traffic --no-extras > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
  timeout 600 python bench.py --dtype bf16x3 --gpus 1 --steps 20 --warmup 5 --no-extras > gpurun_out/${T}_bench_bf16x3_driver.json 2> gpurun_out/${T}_bench_bf16x3_driver.err
  timeout 600 python bench.py --dtype bf16x3 --steps 2000 --warmup 200 --no-extras --no-traffic --no-cpu-baseline > gpurun_out/${T}_bench_bf16x3_2000.json 2> /dev/null
  timeout 600 python bench.py --dtype bf16x3 --algo td3 --rows 4096 --steps 20 --warmup 5 --no-extras --no-traffic --no-cpu-baseline > gpurun_out/${T}_bench_td3_b4096_bf16x3.json 2> /dev/null
  python - <<PY
import json
for n in ("bench_driver", "bench", "bench_bf16x3_driver", "bench_bf16x3_2000", "bench_td3_b4096_bf16x3"):
    try:
        d = json.loads([l for l in open("gpurun_out/${T}_%s.json" % n) if l.startswith("{")][-1])
        pm = d.get("parity_mode")
        print(n, round(d["value"], 1), "steps/s", round(d["ms_per_step"] * 1e3, 2), "us/step", d.get("schedule"), d["dtype"],
              "roofline", d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"),
              "| parity_mode", (round(pm["value"]), pm["loss_curve_deviation"]["within_bound"]) if pm else None)
    except Exception as e:
        print(n, "FAILED", e)
PY
fi
if has stats; then
  bash tools/r3_stats.sh ${T}_cycle > gpurun_out/${T}_cycle_stats.txt 2>&1
  RECNN_SPLIT_FWD=0 bash tools/r3_stats.sh ${T}_fused > gpurun_out/${T}_fused_stats.txt 2>&1
  BENCH_ARGS="--dtype bf16x3 --no-extras" bash tools/r3_stats.sh ${T}_x3 > gpurun_out/${T}_x3_stats.txt 2>&1
  head -10 gpurun_out/${T}_cycle_stats.txt; head -8 gpurun_out/${T}_fused_stats.txt; head -12 gpurun_out/${T}_x3_stats.txt
fi
if has pmc; then
  rm -rf gpurun_out/pmc_*
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d' ' -f1)
    (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$n -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 60 --repeats 1 --no-cpu-baseline --no-traffic --no-extras > /dev/null 2>&1)
    find gpurun_out/pmc_$n -name "*kernel_trace.csv" -delete
  done
  python tools/pmc_r2.py ${T} > gpurun_out/${T}_pmc.txt 2>&1
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES
  cut -c1-300 gpurun_out/${T}_pmc.txt | head -30
  BENCH_ARGS="--dtype bf16x3" bash tools/x3_pmc.sh > gpurun_out/${T}_x3_pmc.txt 2>&1
  head -30 gpurun_out/${T}_x3_pmc.txt
fi
if has dma; then
  mkdir -p tools/_build
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_build/dma_bw tools/dma_bw.hip 2>/dev/null
  timeout 120 tools/_build/dma_bw 256 gpurun_out/${T}_dma_paths_microbench.json
  VARIANTS=0,2 PROBES=0,512,1024,1,2,4,8,16 TRACE=1 OUT=gpurun_out/${T}_x3_fwd_probe_8192.json timeout 300 python tools/x3_fwd_probe.py 2>&1 | grep -v aliased | tee gpurun_out/${T}_x3_fwd_probe_8192.txt
  TRACE=1 TILES=1,2,3,4,5,0 PADS=0,128 PROBES=0 OUT=gpurun_out/${T}_wide_fwd_pitch.json timeout 300 python tools/wide_fwd_probe.py 2>&1 | tee gpurun_out/${T}_wide_fwd_probe.txt
  TILES=1 PADS=0 PROBES=16,1,17 timeout 300 python tools/wide_fwd_probe.py 2>&1 | tee -a gpurun_out/${T}_wide_fwd_probe.txt
  OUT=gpurun_out/${T}_ranger_bw.json timeout 200 python tools/ranger_bw.py | tee gpurun_out/${T}_ranger_bw.txt
fi
if has reinforce; then
  for D in bf16; do
    (cd /tmp && rm -rf /tmp/prof_rf && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_rf -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/reinforce_bench.py --dtype $D > /tmp/prof_rf.json 2>/dev/null)
    f=$(find /tmp/prof_rf -name "*kernel_stats.csv" | head -1)
    cp "$f" gpurun_out/${T}_reinforce_100k_${D}_kernel_stats.csv
    tail -1 /tmp/prof_rf.json | tee gpurun_out/${T}_reinforce_100k_${D}_under_rocprof.json
    python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${T}_reinforce_100k_${D}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
aten = sum(float(r["TotalDurationNs"]) for r in rows if "at::native" in r["Name"] or r["Name"].startswith("Cijk"))
print("ATen + Tensile share of GPU time: %.1f %%" % (100 * aten / tot))
for r in rows[:12]:
    print("%-100s calls %5s avg %8.1f us %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
  done > gpurun_out/${T}_reinforce_stats.txt 2>&1
  cat gpurun_out/${T}_reinforce_stats.txt | cut -c1-200
  timeout 300 python tools/reinforce_bench.py --dtype bf16 | tee gpurun_out/${T}_reinforce_100k_bf16.json
fi
if has bcq; then
  timeout 300 python tools/bcq_bench.py --dtype bf16 --graphed --no-cpu | tee gpurun_out/${T}_bcq_bf16_graphed.json
  (cd /tmp && rm -rf /tmp/prof_bcq && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_bcq -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/bcq_bench.py --dtype bf16 --graphed --no-cpu > /dev/null 2>&1)
  f=$(find /tmp/prof_bcq -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/${T}_bcq_bf16_graphed_kernel_stats.csv
  python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${T}_bcq_bf16_graphed_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
aten = sum(float(r["TotalDurationNs"]) for r in rows if "at::native" in r["Name"] or r["Name"].startswith("Cijk") or "rocclr" in r["Name"])
print("BCQ bf16 graphed: ATen / runtime copies share of GPU time: %.1f %%" % (100 * aten / tot))
for r in rows[:14]:
    print("%-100s calls %5s avg %8.1f us %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
fi 2>&1 | tee gpurun_out/${T}_bcq_stats.txt | cut -c1-200
