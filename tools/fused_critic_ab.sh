#!/bin/bash
# cycle mode: the learning critic's per-step forward as one fused row-panel launch (default) vs layer-1 GEMM + tail
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_bench_shape.py tests/test_gpu_comm.py::test_world_one_collective_steps_equal_single_gpu_steps -m gpu -q -x 2>&1 | tail -4
B="python bench.py --no-traffic --no-cpu-baseline"
for x in 1 0; do
  RECNN_CYCLE_FUSED_CRITIC=$x timeout 300 $B 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused_critic $x', round(d['value']), d['ms_per_step'], d['ms_per_step_samples'], [(k['name'], k['ms']) for k in d['step_breakdown']['cycle_mode']['launches']])"
done
RECNN_CYCLE_FUSED_CRITIC=1 bash tools/r3_stats.sh r03q_fc1 2>&1 | head -8
