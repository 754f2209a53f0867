#!/bin/bash
# Per-round evidence: everything that ends up under profiles/rNN_* (run on the GPU box, outputs in gpurun_out/)
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
R=${1:-r03}
export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py"
# 1. bench lines
timeout -k 10 400 python bench.py --h2d > $O/${R}_bench_line.json 2> $O/${R}_bench_line.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --config hrnet_w32 --steps 20 --warmup 5 > $O/${R}_bench_line_hrnet_w32.json 2>/dev/null
timeout -k 10 300 python bench.py --config performer --steps 30 --warmup 8 > $O/${R}_bench_line_performer.json 2>/dev/null
SCAT_MATH=f32 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/${R}_bench_line_f32mfma.json 2>/dev/null
echo "bench lines done"; cut -c1-200 $O/${R}_bench_line.json
echo "[profiles] kernel stats"
prof() {
  name=$1; shift
  rm -rf $O/prof_$name
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o run -- $B --steps 14 --warmup 8 --no-cpu-baseline --no-roofline > $O/prof_$name.log 2>&1 )
  f=$(find $O/prof_$name -name 'run_kernel_stats.csv' | head -1)
  python3 tools/prof_summary.py $f auto 90 > $O/${R}_bench_kernel_summary_$name.txt 2>&1
  cp $f $O/${R}_bench_kernel_stats_$name.csv
  t=$(find $O/prof_$name -name 'run_kernel_trace.csv' | head -1)
  python3 tools/trace_gaps.py $t 8 60 > $O/${R}_trace_gaps_$name.txt 2>&1
  head -2 $O/${R}_bench_kernel_summary_$name.txt
  rm -rf $O/prof_$name
}
prof default A=1
prof serialized SCAT_DIAG=1 SCAT_SIDE_WGRAD=0 SCAT_OVERLAP_TOKENS=0 SCAT_EARLY_ADAM=0
echo "[profiles] hrnet kernel stats"
rm -rf $O/prof_hr
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hr -o run -- $B --config hrnet_w32 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline > $O/prof_hr.log 2>&1 )
python3 tools/trace_gaps.py $(find $O/prof_hr -name 'run_kernel_trace.csv' | head -1) 4 70 > $O/${R}_hrnet_trace_gaps_default.txt 2>&1
head -1 $O/${R}_hrnet_trace_gaps_default.txt
rm -rf $O/prof_hr
echo "[profiles] traffic"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- $B --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $O/pmc_$c.log 2>&1 )
done
python3 tools/traffic_json.py $(find $O/pmc_FETCH_SIZE -name run_counter_collection.csv | head -1) $(find $O/pmc_WRITE_SIZE -name run_counter_collection.csv | head -1) $O/${R}_traffic.json
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/pmc_$c; done
echo "[profiles] pmc util"
tools/pmc_run.sh $O/pmc_util "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 tools/conv_bench.py --shapes 9,10,15,16 --reps 3 > $O/${R}_pmc_util.txt 2>&1
tools/pmc_run.sh $O/pmc_wait "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" -- python3 tools/conv_bench.py --shapes 9,10,15,16 --reps 3 >> $O/${R}_pmc_util.txt 2>&1
rm -rf $O/pmc_util $O/pmc_wait
echo "[profiles] conv shapes"
timeout -k 10 300 python tools/conv_bench.py --reps 10 > $O/${R}_conv_shapes.txt 2>&1
timeout -k 10 300 python tools/conv_bench.py --reps 10 --shapes 26,27,28,29 > $O/${R}_conv_shapes_hrnet.txt 2>&1
tail -3 $O/${R}_conv_shapes.txt
timeout -k 10 200 python tools/ablate_step.py > $O/${R}_ablate.txt 2>&1
tail -10 $O/${R}_ablate.txt
