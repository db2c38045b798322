cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_x3.py -q 2>&1 | tail -30 > gpurun_out/x3_kernels.log
tail -5 gpurun_out/x3_kernels.log
timeout 900 python -m pytest tests/test_gpu_engine.py -q -s -k "x3 or fp32" 2>&1 | grep -v "^$" | tail -150 > gpurun_out/x3_engine.log
grep -n "Error\|passed\|failed\|unconditioned" gpurun_out/x3_engine.log | head -40
timeout 1200 python -m pytest tests/test_gpu_bench_shape.py -q -s -k "bf16x3" 2>&1 | tail -40 > gpurun_out/x3_shape.log
grep -n "parity\|passed\|failed\|Error" gpurun_out/x3_shape.log | head
