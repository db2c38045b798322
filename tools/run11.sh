cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python tools/catalogue_gemm_probe.py | tee gpurun_out/catalogue_gemm.json
timeout 900 python -m pytest tests/test_gpu_reinforce.py -q 2>&1 | tail -5
timeout 300 python tools/reinforce_bench.py --dtype bf16 | tee gpurun_out/reinforce_100k_bf16_beta_v2.json
timeout 300 python tools/reinforce_bench.py --dtype fp32 | tee gpurun_out/reinforce_100k_fp32_beta_v2.json
