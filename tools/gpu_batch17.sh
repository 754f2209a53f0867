#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rm -rf $O/prof_rows
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rows -o run -- python3 $GRAFT_REPO_ROOT/tools/conv_bench.py --reps 20 --only wgrad --shapes 2,26 > $O/prof_rows.log 2>&1 )
f=$(find $O/prof_rows -name 'run_kernel_stats.csv' | head -1)
python3 tools/prof_summary.py $f 1 12
rm -rf $O/prof_rows
