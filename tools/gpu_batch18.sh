#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for t in 64 128 256 512; do echo "target $t"; SCAT_WG_ROWS_TARGET=$t timeout -k 10 100 python tools/conv_bench.py --reps 20 --only wgrad --shapes 26 2>&1 | grep -E "wgrad"; done
