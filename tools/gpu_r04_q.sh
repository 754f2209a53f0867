#!/bin/bash
# tile size of the stride-2 3x3 forward / data gradient (tools build: SCAT_TUNE forces 128x128 / 64x128 / 64x64)
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
python -m scat_amd.build --diag > /dev/null 2>&1
: > $O/r04_s2_tiles.txt
for t in 0 1 2 3; do
  echo "== SCAT_TUNE=$t" >> $O/r04_s2_tiles.txt
  SCAT_LIBPATH=tools/_bin/libscat_hip_diag.so SCAT_TUNE=$t timeout -k 10 300 python tools/conv_bench.py --reps 10 --shapes 6,12,18 2>/dev/null | grep "fwd\|dgrad" | cut -c1-130 >> $O/r04_s2_tiles.txt
done
cat $O/r04_s2_tiles.txt
