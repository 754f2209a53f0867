#!/bin/bash
cd "$GRAFT_REPO_ROOT"
SCAT_WG_ROWS=3 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad3x3_rows or conv_fwd_dgrad or fused_input" 2>&1 | tail -5
for v in 3 1; do echo "SCAT_WG_ROWS=$v"; SCAT_WG_ROWS=$v timeout -k 10 200 python tools/conv_bench.py --reps 10 --only wgrad --shapes 10,16,22,27,28,29 2>&1 | grep -E "wgrad"; done
