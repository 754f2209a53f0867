#!/bin/bash
# round-4, part D: non-temporal output stores (contraction epilogues + BatchNorm apply passes) by output size: whole step
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
D=$GRAFT_REPO_ROOT/tools/_bin/libscat_hip_diag.so
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("median_ms_per_step"))'
for rep in 1 2; do
  for v in "SCAT_STORE_AUX=0" "SCAT_STORE_AUX=2" "SCAT_STORE_AUX=2 SCAT_STORE_NT_MIN_MB=100" "SCAT_STORE_AUX=2 SCAT_STORE_NT_MIN_MB=200" "SCAT_STORE_AUX=2 SCAT_STORE_NT_MIN_MB=300" "SCAT_STORE_AUX=16"; do
    echo "== $v"; env SCAT_LIBPATH=$D $v timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 15 2>/dev/null | python -c "$P"
  done
done > $O/r04_ab_store_nt.txt 2>&1; cat $O/r04_ab_store_nt.txt
