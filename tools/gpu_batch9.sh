#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
SCAT_PC=7 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "conv1x1 or conv_fwd_dgrad or conv_fused" 2>&1 | tail -1
S=3,5,7,9,11,13,15,17,19,21,23,24,25
for pc in 1 7 0; do echo "== PC $pc"; SCAT_PC=$pc timeout -k 10 200 python tools/conv_bench.py --reps 10 --only fwd,dgrad --shapes $S 2>&1 | grep -v amdgpu | cut -c1-100; done > $O/r02_pc17.txt 2>&1
grep TOTAL $O/r02_pc17.txt
