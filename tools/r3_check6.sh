#!/bin/bash
# round-3 check 6: the device-side collective (tests + forced-DP overhead + 2 ranks on one GPU)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_comm.py -m gpu -q -x -s 2>&1 | tail -30 > gpurun_out/r03j_tests.log
tail -4 gpurun_out/r03j_tests.log
B="python bench.py --no-traffic --no-cpu-baseline"
show() { python - "$1" <<'PY'
import json,sys
try:
    line=[l for l in open(f"gpurun_out/r03j_{sys.argv[1]}.json") if l.startswith("{")][-1]
    d=json.loads(line)
    print(sys.argv[1], round(d["value"],1), d["ms_per_step"], d["ms_per_step_samples"], d["config"]["parallelism"])
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(f"gpurun_out/r03j_{sys.argv[1]}.err").read()[-1500:])
PY
}
timeout 300 $B --force-dp > gpurun_out/r03j_dp1peer.json 2> gpurun_out/r03j_dp1peer.err; show dp1peer
BENCH_ARGS="--force-dp" bash tools/r3_stats.sh r03j_dp1
