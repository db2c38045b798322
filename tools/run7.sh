cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sampler_dense.py -q -x 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_bcq.py -q -k "eager_forward_between or graphed" 2>&1 | tail -8
timeout 300 python bench.py --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-extras > gpurun_out/b20_dense.json 2> gpurun_out/b20_dense.err
python - <<PY
import json
d=json.load(open('gpurun_out/b20_dense.json'))
print(d['value'], d['ms_per_step'], d['ms_per_step_samples'], d['config']['workload'][-60:])
PY
tail -3 gpurun_out/b20_dense.err
