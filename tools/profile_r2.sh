#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_r2.sh TAG [quick]
# GPU tests, the driver's bench line (--steps 20 --warmup 5) and the default one, rocprofv3 kernel stats of the same
# command, PMC passes (HBM traffic; MFMA / CU busy cycles) -> gpurun_out/TAG_*
set -u
T=$1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/${T}_tests.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver.json 2> gpurun_out/${T}_bench_driver.err
timeout 400 python bench.py --no-cpu-baseline --no-traffic > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
if [ "${2:-}" != "quick" ]; then
rm -rf gpurun_out/${T}_prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof -o p --output-format csv -- python bench.py --steps 500 --warmup 100 --no-cpu-baseline --no-traffic > gpurun_out/${T}_bench_rocprof.json 2>/dev/null
cp gpurun_out/${T}_prof/p_kernel_stats.csv gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/${T}_prof
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$n
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$n -o p --output-format csv -- python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-traffic > /dev/null 2>&1
  rm -f gpurun_out/pmc_$n/p_kernel_trace.csv
done
python tools/pmc_r2.py $T
fi
cat gpurun_out/${T}_tests.log
tail -c 600 gpurun_out/${T}_bench_driver.json
tail -c 300 gpurun_out/${T}_bench.json
