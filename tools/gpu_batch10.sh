#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
SCAT_PC=8 timeout -k 10 120 python -m pytest tests/test_gpu_ops.py -x -q -k "conv1x1_pointwise or conv_fwd_dgrad" 2>&1 | tail -2
rc=${PIPESTATUS[0]}
S=3,7,11,13,15,17,19,21,23,24,25
for pc in 8 5 0; do echo "== PC $pc"; SCAT_PC=$pc timeout -k 10 120 python tools/conv_bench.py --reps 10 --only fwd,dgrad --shapes $S 2>&1 | grep -v amdgpu | cut -c1-100; done > $O/r02_pc8.txt 2>&1
grep TOTAL $O/r02_pc8.txt
