cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_x3.py -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_engine.py -q -k "x3" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_gpu_bench_shape.py -q -s -k "bf16x3" 2>&1 | grep -n "parity\|passed\|failed\|Error\|assert" | head
timeout 600 python bench.py --dtype bf16x3 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline > gpurun_out/x3_bench20.json 2> gpurun_out/x3_bench20.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/x3_bench20.json'))
print(d['value'], d['ms_per_step'], d['ms_per_step_samples'])
for l in d['step_breakdown']['launches']: print(l)
print(d['step_breakdown']['sum_kernel_ms'], d['step_breakdown']['policy_step_sum_kernel_ms'])
PY
timeout 600 python bench.py --dtype bf16x3 --steps 2000 --warmup 100 --no-traffic --no-cpu-baseline > gpurun_out/x3_bench2000.json 2> gpurun_out/x3_bench2000.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/x3_bench2000.json'))
print(d['value'], d['ms_per_step'], d['ms_per_step_samples'])
PY
