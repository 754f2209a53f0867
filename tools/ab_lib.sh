#!/bin/bash
# same-box A/B of the shipped library against a variant build: tools/ab_lib.sh tools/_bin/libscat_hip_X.so [reps]
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
V=$GRAFT_REPO_ROOT/$1; N=${2:-2}
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("median_ms_per_step"))'
for rep in $(seq $N); do
  for lib in "" "$V"; do
    echo "== ${lib:-shipped}"
    if [ -z "$lib" ]; then timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 15 2>/dev/null | python -c "$P"
    else SCAT_LIBPATH=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 15 2>/dev/null | python -c "$P"; fi
  done
done
