#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5b; mkdir -p $O
echo "== ws probe bits: 3 barriers only, 19 + no epilogue, 32 exit at entry, 16 full loop without epilogue"
VARIANTS=2 PROBES=0,3,19,32,16 OUT=$O/probe_fixed_8192.json timeout 300 python tools/x3_fwd_probe.py 2>&1 | grep variant | tee $O/probe_fixed_8192.log
M=2048 VARIANTS=2,4 PROBES=0,3,19,32,16 OUT=$O/probe_fixed_2048.json timeout 300 python tools/x3_fwd_probe.py 2>&1 | grep variant | tee $O/probe_fixed_2048.log
