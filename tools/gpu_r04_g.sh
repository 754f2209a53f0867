#!/bin/bash
# HRNet-W32 kernel summary of the round-4 tree (default streams and serialized)
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py"
prof() {
  name=$1; shift
  rm -rf $O/prof_$name
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o run -- $B --config hrnet_w32 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline > $O/prof_$name.log 2>&1 )
  f=$(find $O/prof_$name -name 'run_kernel_stats.csv' | head -1)
  python3 tools/prof_summary.py $f auto 70 > $O/r04_hrnet_kernel_summary_$name.txt 2>&1
  t=$(find $O/prof_$name -name 'run_kernel_trace.csv' | head -1)
  python3 tools/trace_gaps.py $t 8 > $O/r04_hrnet_trace_gaps_$name.txt 2>&1
  head -2 $O/r04_hrnet_kernel_summary_$name.txt
  rm -rf $O/prof_$name
}
prof default A=1
prof serialized SCAT_DIAG=1 SCAT_HRNET_PAR=0 SCAT_SIDE_WGRAD=0 SCAT_OVERLAP_TOKENS=0 SCAT_EARLY_ADAM=0
