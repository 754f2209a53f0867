#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
t=$(find $O/prof_hrnet -name run_kernel_trace.csv | head -1); python3 tools/trace_gaps.py $t 4 > $O/r02_hrnet_trace_gaps_serialized.txt 2>&1; head -12 $O/r02_hrnet_trace_gaps_serialized.txt; rm -rf $O/prof_hrnet
( cd /tmp && SCAT_SIDE_WGRAD=0 SCAT_OVERLAP_TOKENS=0 SCAT_EARLY_ADAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hrnet -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config hrnet_w32 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/prof_hrnet.log 2>&1 )
f=$(find $O/prof_hrnet -name 'run_kernel_stats.csv' | head -1)
python3 tools/prof_summary.py $f 9 70 > $O/r02_hrnet_kernel_summary_serialized.txt 2>&1
head -75 $O/r02_hrnet_kernel_summary_serialized.txt
t=$(find $O/prof_hrnet -name run_kernel_trace.csv | head -1); python3 tools/trace_gaps.py $t 4 > $O/r02_hrnet_trace_gaps_serialized.txt 2>&1; head -12 $O/r02_hrnet_trace_gaps_serialized.txt; rm -rf $O/prof_hrnet
