#!/bin/bash
# stream kernel (csrc/mlps.hip) vs the default: equality test, then A/B bench (usage: bash tools/quick_mlps.sh TAG)
export TMPDIR=/tmp
T=$1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "mlp_forward_64" 2>&1 | tail -25 > gpurun_out/${T}_ktest.log
tail -12 gpurun_out/${T}_ktest.log
for k in 0 3 0 3; do
RECNN_MLP_KERNEL=$k timeout 200 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-traffic > gpurun_out/${T}_q.json 2>gpurun_out/${T}_q.err || tail -3 gpurun_out/${T}_q.err
python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/${T}_q.json").read().strip().splitlines()[-1])
    print("kernel $k: %.2f us/step  " % (j["ms_per_step"]*1e3), [(l["name"], round(l["ms"]*1e3,2)) for l in j["step_breakdown"]["launches"]][:5], "frac %.4f" % j["roofline"]["frac"])
except Exception as e:
    print("kernel $k: no result", e)
PY
done
