#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5f; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_x3.py -q -k "variants" 2>&1 | tail -3
run() { # name env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-traffic --no-extras $EXTRA > $O/$n.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.json")); print("$n", round(d["ms_per_step"]*1000,2), "us", d.get("schedule"))
except Exception as e: print("$n FAILED", e)
PY
}
echo "== bf16 (2000 steps: cycle schedule)"; EXTRA=""
run base A=1; run splits4 RECNN_DW_SPLITS=4; run splits2 RECNN_DW_SPLITS=2; run dwdma3 RECNN_DW_DMA=3; run dwdma4 RECNN_DW_DMA=4; run dwdma6 RECNN_DW_DMA=6; run base2 A=1
echo "== bf16 fused schedule"; 
run f_base RECNN_SPLIT_FWD=0; run f_splits4 RECNN_SPLIT_FWD=0 RECNN_DW_SPLITS=4; run f_dwdma3 RECNN_SPLIT_FWD=0 RECNN_DW_DMA=3; run f_dwdma4 RECNN_SPLIT_FWD=0 RECNN_DW_DMA=4; run f_base2 RECNN_SPLIT_FWD=0
echo "== bf16x3"; EXTRA="--dtype bf16x3"
run x_base A=1; run x_splits4 RECNN_DW_SPLITS=4; run x_splits2 RECNN_DW_SPLITS=2; run x_base2 A=1
