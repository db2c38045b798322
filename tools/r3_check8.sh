#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench_shape.py::test_reference_shaped_loop_on_planned_batches_equals_run tests/test_gpu_reinforce.py::test_vocab_parallel_head_on_hip_gemms tests/test_gpu_reinforce.py::test_reinforce_full_cycle_at_100k_catalogue_vs_oracle -m gpu -q -x -s 2>&1 | tail -30 > gpurun_out/r03l_tests.log
tail -12 gpurun_out/r03l_tests.log | cut -c1-1500
timeout 300 python tools/update_loop_rate.py bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03l_update_loop_rate.txt
RECNN_BENCH_SINGLE_DEVICE=1 timeout 100 python bench.py --no-traffic --no-cpu-baseline --gpus 2 --steps 20 --warmup 5 --repeats 2 > gpurun_out/r03l_dp2peer.json 2> gpurun_out/r03l_dp2peer.err
grep '^{' gpurun_out/r03l_dp2peer.json | cut -c1-700; tail -2 gpurun_out/r03l_dp2peer.err | cut -c1-200
