#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
for pc in 5 6; do SCAT_PC=$pc timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "conv1x1 or conv_fwd_dgrad or conv_fused" 2>&1 | tail -1; done
S=3,4,7,9,11,13,15,17,19,21,23,24,25
for pc in 0 5 6; do echo "== PC $pc"; SCAT_PC=$pc timeout -k 10 200 python tools/conv_bench.py --reps 10 --only fwd,dgrad --shapes $S 2>&1 | grep -v amdgpu | cut -c1-100; done > $O/r02_pc56.txt 2>&1
tail -3 $O/r02_pc56.txt
