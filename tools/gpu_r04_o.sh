#!/bin/bash
# fold bn3's backward at 14x14 too, now that its reduction rides in an epilogue? (same-box A/B)
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
: > $O/r04_ab_bnb14.txt
for rep in 1 2 3; do
  for v in "A=1" "SCAT_DIAG=1 SCAT_BNB_MIN_H=14"; do
    echo "== $v" >> $O/r04_ab_bnb14.txt
    env $v timeout -k 10 250 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O/r04_ab_bnb14.txt
  done
done
cat $O/r04_ab_bnb14.txt
