set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_x3.py -x -q 2>&1 | tail -30 > gpurun_out/x3_kernels.log
cat gpurun_out/x3_kernels.log
timeout 900 python -m pytest tests/test_gpu_engine.py -q -k "bf16x3" 2>&1 | tail -40 > gpurun_out/x3_engine.log
cat gpurun_out/x3_engine.log
timeout 600 python bench.py --dtype bf16x3 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline > gpurun_out/x3_bench20.json 2> gpurun_out/x3_bench20.err
tail -c 3000 gpurun_out/x3_bench20.json; tail -5 gpurun_out/x3_bench20.err
