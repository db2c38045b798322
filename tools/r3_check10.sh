#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_comm.py -m gpu -q -x 2>&1 | tail -5
B="python bench.py --no-traffic --no-cpu-baseline --steps 20 --warmup 5 --repeats 9"
for cfg in "base:" "cyc3:RECNN_CYCLE_MIN_LEN=20 RECNN_CYCLE_MIN_SEG=3" "cyc5:RECNN_CYCLE_MIN_LEN=20 RECNN_CYCLE_MIN_SEG=5" "cyc8:RECNN_CYCLE_MIN_LEN=20 RECNN_CYCLE_MIN_SEG=8" "base2:"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 $B 2>/dev/null | grep '^{' > gpurun_out/r03n_$name.json
  python - "$name" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/r03n_{sys.argv[1]}.json"))
print(sys.argv[1], round(d["value"]), d["ms_per_step"], sorted(d["ms_per_step_samples"]))
PY
done
timeout 300 python bench.py --no-traffic --no-cpu-baseline --force-dp 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dp1', round(d['value']), d['ms_per_step'], d['schedule'], d['config']['parallelism'])"
