#!/bin/bash
# fresh profile of the step at HEAD: kernel stats (default and serialized), idle gaps, timeline
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
prof() {   # name, env...
  name=$1; shift
  rm -rf $O/prof_$name
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 14 --warmup 8 --no-cpu-baseline --no-roofline > $O/prof_$name.log 2>&1 )
  f=$(find $O/prof_$name -name 'run_kernel_stats.csv' | head -1)
  python3 tools/prof_summary.py $f auto 70 > $O/r02_kernel_summary_$name.txt 2>&1
  t=$(find $O/prof_$name -name 'run_kernel_trace.csv' | head -1)
  python3 tools/trace_gaps.py $t 8 > $O/r02_trace_gaps_$name.txt 2>&1
  cp $f $O/r02_kernel_stats_$name.csv
  head -3 $O/r02_kernel_summary_$name.txt
}
prof default A=1
prof serialized SCAT_SIDE_WGRAD=0 SCAT_OVERLAP_TOKENS=0 SCAT_EARLY_ADAM=0
for e in "SCAT_WG_TARGET1=512" "SCAT_WG_TARGET1=512 SCAT_WG_TARGET9=1024" "SCAT_WG_PC=0" "A=1"; do
  echo "== $e"; env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['median_ms_per_step'])"
done > $O/r02_bench_ab.txt 2>&1
cat $O/r02_bench_ab.txt
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r02_gputests.txt 2>&1
tail -4 $O/r02_gputests.txt
