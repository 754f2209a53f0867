#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for m in 1 0; do
  SCAT_WG_PC=$m timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_fwd_dgrad_wgrad or stride2 or fused or bnb or batchnorm_backward" > $O/r02_t_wg$m.txt 2>&1
  tail -2 $O/r02_t_wg$m.txt
done
for m in 0 1; do
  SCAT_WG_PC=$m timeout -k 10 300 python tools/conv_bench.py --reps 10 --only wgrad > $O/r02_wg_pc$m.txt 2>&1
  tail -1 $O/r02_wg_pc$m.txt
done
for t in 256 384 512 768; do
  echo "== target $t"
  SCAT_WG_PC=1 SCAT_WG_TARGET=$t timeout -k 10 300 python tools/conv_bench.py --reps 10 --only wgrad --shapes 9,10,11,15,16,17 2>&1 | grep -v amdgpu | cut -c1-110
done > $O/r02_wg_target.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r02_gputests.txt 2>&1
tail -5 $O/r02_gputests.txt
for m in 0 1; do
  SCAT_WG_PC=$m timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 > $O/r02_bench_wg$m.txt 2>&1
  tail -1 $O/r02_bench_wg$m.txt | cut -c1-400
done
