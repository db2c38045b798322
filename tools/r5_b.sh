#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5b; mkdir -p $O
for v in 7 8 9; do RECNN_X3_FWD_DEBUG=$v timeout 400 python -m pytest tests/test_gpu_x3.py -q -x -k gemm_fwd 2>&1 | tail -2; done
for M in 8192 10240 20480; do
M=$M VARIANTS=0,2,9,7,10,8 PROBES=0 OUT=$O/probe_big_$M.json timeout 300 python tools/x3_fwd_probe.py 2>&1 | grep variant | tee $O/probe_big_$M.log
done
M=20480 VARIANTS=7,8 PROBES=1,2,16 timeout 300 python tools/x3_fwd_probe.py 2>&1 | grep variant | tee $O/probe_big_bits.log
