#!/bin/bash
# one GPU-box visit: calibration probe, wide-form correctness + A/B, then the GPU test suite
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 120 tools/_bin/mfma_probe > $O/r02_mfma_probe.txt 2>&1
for pc in 3 4; do
  SCAT_PC=$pc timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "conv1x1 or conv_fwd or conv_fused" > $O/r02_t_pc$pc.txt 2>&1
  tail -2 $O/r02_t_pc$pc.txt
done
S=3,5,7,9,11,13,15,17,23,24
for pc in 0 1 3 4; do
  SCAT_PC=$pc timeout -k 10 200 python tools/conv_bench.py --reps 10 --only fwd,dgrad --shapes $S > $O/r02_w4_pc$pc.txt 2>&1
  tail -2 $O/r02_w4_pc$pc.txt
done
for d in 0 1 2 4 8 11 20; do
  echo "== DIAG $d"
  SCAT_PC=3 SCAT_TUNE=$((100+d)) timeout -k 10 200 python tools/conv_bench.py --reps 10 --only fwd --shapes 9,11,13,15,17 2>&1 | grep -v amdgpu.ids | cut -c1-100
done > $O/r02_diag_w4.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r02_gputests.txt 2>&1
tail -5 $O/r02_gputests.txt
