#!/bin/bash
# round-3 check 5: REINFORCE at N=100k vs the oracle, the mixed cycle/fused segments, optimizer scalar table A/B
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_reinforce.py tests/test_gpu_bench_shape.py tests/test_gpu_parity_r2.py tests/test_gpu_split.py -m gpu -q -x -s 2>&1 | tail -25 > gpurun_out/r03i_tests.log
tail -6 gpurun_out/r03i_tests.log
B="python bench.py --no-traffic --no-cpu-baseline"
for cfg in "base:" "minseg1:RECNN_CYCLE_MIN_SEG=1" "opttab:RECNN_OPT_TABLE=1" "fused:RECNN_SPLIT_FWD=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 $B > gpurun_out/r03i_$name.json 2> gpurun_out/r03i_$name.err
  python - "$name" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r03i_{sys.argv[1]}.json"))
print(sys.argv[1], d["value"], d["ms_per_step"], d["ms_per_step_samples"], d.get("schedule"))
PY
done
env timeout 300 $B --steps 20 --warmup 5 > gpurun_out/r03i_drv.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03i_drv.json')); print('drv', d['value'], d['ms_per_step_samples'])"
