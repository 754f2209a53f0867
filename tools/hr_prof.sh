#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp
rm -rf $O/prof_hr
( cd /tmp && SCAT_DIAG=1 SCAT_SIDE_WGRAD=0 SCAT_OVERLAP_TOKENS=0 SCAT_EARLY_ADAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hr -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config hrnet_w32 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline > $O/prof_hr.log 2>&1 )
f=$(find $O/prof_hr -name 'run_kernel_stats.csv' | head -1)
python3 tools/prof_summary.py $f auto 70 > $O/r03_hrnet_kernel_summary_serialized.txt 2>&1
rm -rf $O/prof_hr
head -3 $O/r03_hrnet_kernel_summary_serialized.txt

