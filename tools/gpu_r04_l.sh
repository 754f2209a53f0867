#!/bin/bash
# BatchNorm-backward sums in the data-gradient epilogue: tests, then a same-box A/B of the step
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "epilogue" 2>&1 | tail -8
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "trainstep or encoder_transformer_golden or against_oracle_config1 or bottleneck_golden" 2>&1 | tail -4
: > $O/r04_ab_epi_bnb.txt
for rep in 1 2 3; do
  for v in "A=1" "SCAT_DIAG=1 SCAT_EPI_BNB=0"; do
    echo "== $v" >> $O/r04_ab_epi_bnb.txt
    env $v timeout -k 10 250 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O/r04_ab_epi_bnb.txt
  done
done
cat $O/r04_ab_epi_bnb.txt
