#!/bin/bash
cd "$GRAFT_REPO_ROOT"
PROBE_PC=1 timeout -k 10 120 tools/_bin/mfma_probe | grep -E "16x16|idle loop|\+ split \("
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "conv or bn or batchnorm" 2>&1 | tail -3
for v in 0 1; do echo "SCAT_PW_PLAIN=$v"; SCAT_PW_PLAIN=$v timeout -k 10 200 python tools/conv_bench.py --reps 10 --only fwd,dgrad 2>&1 | grep -E "k1 s1|TOTAL"; done
