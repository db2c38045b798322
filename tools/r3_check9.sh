#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench_shape.py::test_reference_shaped_loop_on_planned_batches_equals_run tests/test_gpu_reinforce.py::test_vocab_parallel_head_on_hip_gemms "tests/test_gpu_reinforce.py::test_reinforce_full_cycle_at_100k_catalogue_vs_oracle" -m gpu -q -s 2>&1 | tail -30 > gpurun_out/r03m_tests.log
tail -12 gpurun_out/r03m_tests.log | cut -c1-1800
timeout 300 python tools/update_loop_rate.py bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03m_update_loop_rate.txt
