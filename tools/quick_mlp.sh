#!/bin/bash
# quick check of a change to the fused MLP forward: kernel + engine tests, then the bench (usage: bash tools/quick_mlp.sh TAG)
export TMPDIR=/tmp
T=$1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/${T}_ktest.log
tail -8 gpurun_out/${T}_ktest.log
for i in 1 2; do
timeout 200 python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --no-traffic > gpurun_out/${T}_q.json 2>/dev/null
python - <<PY
import json
j=json.loads(open("gpurun_out/${T}_q.json").read().strip().splitlines()[-1])
print("%.2f us/step  " % (j["ms_per_step"]*1e3), [(l["name"], round(l["ms"]*1e3,2)) for l in j["step_breakdown"]["launches"]][:5], "frac %.4f" % j["roofline"]["frac"])
PY
done
