#!/bin/bash
# usage: ab.sh "ENV=VAL ..." tag
env $1 timeout 200 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline > gpurun_out/ab_$2.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/ab_$2.json"))
print("$2", round(d["value"]), "steps/s", round(d["ms_per_step"]*1000,1), "us;", " ".join("%s=%.1f" % (l["name"], l["ms"]*1000) for l in d["step_breakdown"]["launches"]))
PY
