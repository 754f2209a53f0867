#!/bin/bash
# kernel trace of the bench step -> kernel summary, idle-gap summary and the timeline around the gaps
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
N=${1:-r03}
export TMPDIR=/tmp
rm -rf $O/prof_$N
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$N -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 14 --warmup 8 --no-cpu-baseline --no-roofline > $O/prof_$N.log 2>&1 )
f=$(find $O/prof_$N -name 'run_kernel_stats.csv' | head -1)
t=$(find $O/prof_$N -name 'run_kernel_trace.csv' | head -1)
python3 tools/prof_summary.py $f auto 90 > $O/${N}_bench_kernel_summary_default.txt 2>&1
python3 tools/trace_gaps.py $t 8 > $O/${N}_trace_gaps_default.txt 2>&1
python3 tools/step_timeline.py $t 10 12 > $O/${N}_step_timeline_gaps.txt 2>&1
python3 tools/step_timeline.py $t 10 0 > $O/${N}_step_timeline_full.txt 2>&1
rm -rf $O/prof_$N
head -3 $O/${N}_trace_gaps_default.txt
