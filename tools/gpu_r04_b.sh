#!/bin/bash
# round-4, part B: deferred-reduce test + A/B, 7x7-plane tile choices, WRITE_SIZE check of the 4-byte epilogue stores
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_dp.py tests/test_gpu_schedule.py -m gpu -q -x -k "deferred or golden or trainstep or rccl or switches or full_size_batch96_against_oracle or dropin" > $O/r04_b_tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/r04_b_tests.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("median_ms_per_step"))'
for rep in 1 2; do
  for v in "A=1" "SCAT_DIAG=1 SCAT_WG_DEFER=0"; do
    echo "== $v"; env $v timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 15 2>/dev/null | python -c "$P"
  done
done > $O/r04_ab_defer.txt 2>&1; cat $O/r04_ab_defer.txt
echo "[7x7 tiles]"
D=$GRAFT_REPO_ROOT/tools/_bin/libscat_hip_diag.so
for v in "A=1" "SCAT_PW_THIN=200" "SCAT_TUNE=2" "SCAT_TUNE=3"; do
  echo "== $v"; env SCAT_LIBPATH=$D $v timeout -k 10 200 python tools/conv_bench.py --shapes 19,21,25,17 --only fwd,dgrad --reps 10 2>/dev/null | grep "k1"
done > $O/r04_tiles_7x7.txt 2>&1; cat $O/r04_tiles_7x7.txt
echo "[write calib]"
tools/pmc_run.sh $O/pmc_w "WRITE_SIZE" -- python3 tools/conv_bench.py --shapes 3 --only fwd --reps 3 > $O/r04_write_calib.txt 2>&1; grep -A1 "split_kernel" $O/r04_write_calib.txt | head -6
rm -rf $O/pmc_w
