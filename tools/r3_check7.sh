#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_comm.py -m gpu -q -x -s 2>&1 | tail -30 > gpurun_out/r03k_tests.log
tail -12 gpurun_out/r03k_tests.log
RECNN_BENCH_SINGLE_DEVICE=1 timeout 100 python bench.py --no-traffic --no-cpu-baseline --gpus 2 --steps 20 --warmup 5 --repeats 2 > gpurun_out/r03k_dp2peer.json 2> gpurun_out/r03k_dp2peer.err
grep '^{' gpurun_out/r03k_dp2peer.json | cut -c1-400; tail -3 gpurun_out/r03k_dp2peer.err | cut -c1-200
