#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -m pytest tests/test_gpu_x3.py -x -q 2>&1 | tail -3
timeout 200 python bench.py --dtype bf16x3 --steps 200 --warmup 20 --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3', d['value'], d['ms_per_step'])
for l in d['step_breakdown']['launches']: print('  ', l['name'], round(l['ms']*1e3,2))"
