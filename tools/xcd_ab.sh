#!/bin/bash
mkdir -p gpurun_out
RECNN_MLP_XCD=7 timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_bench_shape.py tests/test_gpu_parity_r2.py -m gpu -q -x 2>&1 | tail -4
B="python bench.py --no-traffic --no-cpu-baseline --steps 20 --warmup 5 --repeats 9"
for x in 0 5 7 0 5; do
  RECNN_MLP_XCD=$x timeout 300 $B 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('xcd $x', round(d['value']), d['ms_per_step'], sorted(d['ms_per_step_samples'])[:3], d['roofline']['avg_ms'])"
done
RECNN_MLP_XCD=5 timeout 400 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('traffic xcd5', d['roofline']['traffic'], d['roofline']['traffic_source'][-60:], d['roofline']['avg_ms'])"
RECNN_MLP_XCD=5 RECNN_SPLIT_FWD=0 bash tools/r3_stats.sh r03p_xcd5 2>&1 | grep "under rocprof\|mlps_fwd"
RECNN_SPLIT_FWD=0 bash tools/r3_stats.sh r03p_xcd0 2>&1 | grep "under rocprof\|mlps_fwd"
