cd $GRAFT_REPO_ROOT
timeout 120 python tools/x3_fwd_probe.py
M=2048 timeout 120 python tools/x3_fwd_probe.py
