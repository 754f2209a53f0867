#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
for t in 0 201 202 203 204 208 212 216 219; do
  printf "TUNE=%s  " $t
  SCAT_TUNE=$t SCAT_LIBPATH=$GRAFT_REPO_ROOT/tools/_bin/libscat_hip_diag.so timeout -k 10 120 python tools/conv_bench.py --shapes 11,17 --reps 10 --only wgrad 2>/dev/null | grep "k1 s1" | awk '{printf "%s %s us   ", $1, $7}'
  echo
done
