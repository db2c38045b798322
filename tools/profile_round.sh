#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh TAG
# bench line, rocprofv3 kernel stats of the same command, PMC traffic passes -> gpurun_out/TAG_*
set -u
T=$1
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/${T}_tests.log
timeout 300 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
rm -rf gpurun_out/${T}_prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof -o p --output-format csv -- python bench.py --steps 500 --warmup 100 --no-cpu-baseline > gpurun_out/${T}_bench_rocprof.json 2>/dev/null
cp gpurun_out/${T}_prof/p_kernel_stats.csv gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/${T}_prof
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p --output-format csv -- python bench.py --steps 60 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
  rm -f gpurun_out/pmc_$c/p_kernel_trace.csv
done
cat gpurun_out/${T}_tests.log
python tools/kstats.py gpurun_out/${T}_kernel_stats.csv
python tools/pmc.py
tail -c 400 gpurun_out/${T}_bench.json
