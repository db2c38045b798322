cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > gpurun_out/gpu_suite.log 2>&1
tail -30 gpurun_out/gpu_suite.log
