#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rm -rf $O/prof_x
( cd /tmp && env SCAT_DIAG=1 SCAT_SIDE_WGRAD=0 SCAT_OVERLAP_TOKENS=0 SCAT_EARLY_ADAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 6 --no-cpu-baseline --no-roofline > $O/prof_x.log 2>&1 )
python3 tools/trace_gaps.py $(find $O/prof_x -name 'run_kernel_trace.csv' | head -1) 6 45 > $O/r04_epi_bnb_serialized.txt 2>&1
grep "conv1x1_split_kernel\|bn_bwd\|launches" $O/r04_epi_bnb_serialized.txt | cut -c1-150
rm -rf $O/prof_x
