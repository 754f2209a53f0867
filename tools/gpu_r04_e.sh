#!/bin/bash
# round-4, part E: staggered start of the co-resident workgroups of the pointwise kernel (tools build, SCAT_TUNE=300+t)
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
D=$GRAFT_REPO_ROOT/tools/_bin/libscat_hip_diag.so
for t in 0 300 303 306 310 315; do
  echo "== SCAT_TUNE=$t"
  SCAT_LIBPATH=$D SCAT_TUNE=$t timeout -k 10 200 python tools/conv_bench.py --shapes 7,9,11,13,17 --only fwd,dgrad --reps 10 2>/dev/null | grep "k1\|TOTAL"
done > $O/r04_stagger.txt 2>&1
cat $O/r04_stagger.txt
