#!/bin/bash
# the whole GPU suite + smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/measured_bounds.json
( time timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > gpurun_out/r05_tests.log 2>&1
tail -15 gpurun_out/r05_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/r05_smoke.log
cp gpurun_out/measured_bounds.json gpurun_out/r05_measured_bounds.json 2>/dev/null
