#!/bin/bash
# round-4, part C: cache policy of the contraction epilogue's stores (default / sc1 / nt): time and HBM reads
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
D=$GRAFT_REPO_ROOT/tools/_bin/libscat_hip_diag.so
for aux in 0 16 2; do
  echo "== SCAT_STORE_AUX=$aux (time)"
  SCAT_LIBPATH=$D SCAT_STORE_AUX=$aux timeout -k 10 200 python tools/conv_bench.py --shapes 3,4,7,9,11,13,15,17,19,21 --only fwd,dgrad --reps 10 2>/dev/null | grep "k1\|TOTAL"
done > $O/r04_store_policy.txt 2>&1
for aux in 0 16; do
  echo "== SCAT_STORE_AUX=$aux (FETCH_SIZE KiB per launch, second launch of each kernel)"
  SCAT_LIBPATH=$D SCAT_STORE_AUX=$aux tools/pmc_run.sh $O/pmc_st$aux "FETCH_SIZE" -- python3 tools/conv_bench.py --shapes 7,11,13,15,17 --only fwd,dgrad --reps 2 2>&1 | grep -A1 "split_kernel\|pc_kernel"
  python3 - $O/pmc_st$aux/run_counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE" and ("split_kernel" in r["Kernel_Name"] or "pc_kernel" in r["Kernel_Name"]) and "taps" not in r["Kernel_Name"]:
        acc.setdefault(r["Dispatch_Id"], (r["Kernel_Name"][:60], float(r["Counter_Value"])))
print("dispatch order (MB fetched, counter / 0.565 for the 4-byte-load kernels):")
for k, (n, v) in acc.items():
    print(f"  {n:60s} {v * 1024 / 0.565 / 1e6:8.1f}")
PY
  rm -rf $O/pmc_st$aux
done >> $O/r04_store_policy.txt 2>&1
cat $O/r04_store_policy.txt
