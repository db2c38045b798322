cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_comm.py -q -x 2>&1 | tail -8
RECNN_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --preflight > gpurun_out/preflight_2ranks_one_gpu.jsonl 2> gpurun_out/preflight.err
echo "rc=$?"
cat gpurun_out/preflight_2ranks_one_gpu.jsonl | cut -c1-400
tail -5 gpurun_out/preflight.err
