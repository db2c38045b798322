cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_x3.py -q 2>&1 | tail -8
timeout 300 python -m pytest tests/test_gpu_engine.py -q -k "x3" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_bench_shape.py -q -s -k "bf16x3" 2>&1 | grep -n "parity\|passed\|failed\|Error\|assert" | head
timeout 300 python bench.py --dtype bf16x3 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline > gpurun_out/x3_b20.json 2> gpurun_out/x3_b20.err
python - <<PY
import json
d=json.load(open('gpurun_out/x3_b20.json'))
print(d['value'], d['ms_per_step'], d['ms_per_step_samples'])
print([(l['name'], round(l['ms']*1e3,1)) for l in d['step_breakdown']['launches']])
PY
