cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_x3.py -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_engine.py -q -k "x3" 2>&1 | tail -5
for mode in 0 2 3; do
  echo "== RECNN_X3_TILE=$mode"
  RECNN_X3_TILE=$mode timeout 600 python bench.py --dtype bf16x3 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline > gpurun_out/x3_b20_$mode.json 2> gpurun_out/x3_b20_$mode.err
  RECNN_X3_TILE=$mode python - <<PY
import json
d=json.load(open('gpurun_out/x3_b20_$mode.json'))
print(d['value'], d['ms_per_step'], d['ms_per_step_samples'])
print([(l['name'], round(l['ms']*1e3,1)) for l in d['step_breakdown']['launches']])
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/x3prof -o x3 -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16x3 --steps 600 --warmup 60 --repeats 2 --no-traffic --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/x3_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/x3_prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/x3prof -name "*kernel_stats*" | head
f=$(find gpurun_out/x3prof -name "*kernel_stats.csv" | head -1)
head -25 $f | cut -c1-200
