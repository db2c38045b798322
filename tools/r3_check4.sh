#!/bin/bash
# round-3 check 4: new parity tests + bounds recorder + the new bench.py reporting
mkdir -p gpurun_out; rm -f gpurun_out/measured_bounds.json
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_bench_shape.py tests/test_gpu_parity_r2.py tests/test_gpu_reinforce.py -m gpu -q -x -s 2>&1 | tail -40 > gpurun_out/r03h_tests.log
tail -5 gpurun_out/r03h_tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03h_bench_driver.json 2> gpurun_out/r03h_bench_driver.err
cat gpurun_out/r03h_bench_driver.json; tail -3 gpurun_out/r03h_bench_driver.err
timeout 900 python bench.py > gpurun_out/r03h_bench_default.json 2> gpurun_out/r03h_bench_default.err
cat gpurun_out/r03h_bench_default.json; tail -3 gpurun_out/r03h_bench_default.err
