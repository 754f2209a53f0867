#!/bin/bash
# round-4, part F: planes tests after the M0 change; HRNet-W32 step with the BatchNorm sums from the convolution epilogues
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "planes" 2>&1 | tail -2
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("median_ms_per_step"))'
for rep in 1 2; do
  for v in "A=1" "SCAT_DIAG=1 SCAT_HRNET_EPI=1" "SCAT_DIAG=1 SCAT_HRNET_CBR_EPI=0"; do
    echo "== $v"; env $v timeout -k 10 300 python bench.py --config hrnet_w32 --no-cpu-baseline --no-roofline --steps 20 --warmup 6 2>/dev/null | python -c "$P"
  done
done > $O/r04_ab_hrnet_epi.txt 2>&1; cat $O/r04_ab_hrnet_epi.txt
