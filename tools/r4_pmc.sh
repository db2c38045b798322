#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel trace only) and MFMA / CU busy cycles per kernel of the bf16
# step in both schedules -> gpurun_out/r04_pmc.json / r04_pmc.txt (aggregation: tools/pmc_r2.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_*
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$n -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 120 --warmup 60 --repeats 1 --no-cpu-baseline --no-traffic --no-extras > /dev/null 2>&1)
  find gpurun_out/pmc_$n -name "*kernel_trace.csv" -delete
done
python tools/pmc_r2.py r04 > gpurun_out/r04_pmc.txt 2>&1
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES
cat gpurun_out/r04_pmc.txt | cut -c1-330
