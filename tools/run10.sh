cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_reinforce.py -q -k "beta_net" 2>&1 | tail -8
timeout 300 python tools/reinforce_bench.py --dtype fp32 > gpurun_out/reinforce_100k_fp32_beta.json 2> gpurun_out/rb.err; cat gpurun_out/reinforce_100k_fp32_beta.json; tail -2 gpurun_out/rb.err
timeout 300 python tools/reinforce_bench.py --dtype bf16 > gpurun_out/reinforce_100k_bf16_beta.json 2>> gpurun_out/rb.err; cat gpurun_out/reinforce_100k_bf16_beta.json
timeout 300 python tools/reinforce_bench.py --dtype bf16 --beta frozen > gpurun_out/reinforce_100k_bf16_frozen.json 2>> gpurun_out/rb.err; cat gpurun_out/reinforce_100k_bf16_frozen.json
