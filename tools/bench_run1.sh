cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err ) 2>&1 | tail -3
tail -3 gpurun_out/bench_driver.err
python - <<PY
import json
d=json.load(open('gpurun_out/bench_driver.json'))
for k in ('value','ms_per_step','schedule','roofline','roofline_end_to_end','roofline_gather','loss_curve_deviation','parity_mode','other_configs','extras_error','cpu_baseline'):
    print(k, json.dumps(d.get(k))[:900])
PY
