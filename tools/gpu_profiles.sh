#!/bin/bash
# Round-2 evidence: everything that ends up under profiles/r02_* (run on the GPU box, outputs in gpurun_out/)
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py"
# 1. bench lines
timeout -k 10 400 python bench.py --h2d > $O/r02_bench_line.json 2> $O/r02_bench_line.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --config hrnet_w32 --steps 20 --warmup 5 > $O/r02_bench_line_hrnet_w32.json 2>/dev/null
timeout -k 10 300 python bench.py --config performer --steps 30 --warmup 8 > $O/r02_bench_line_performer.json 2>/dev/null
SCAT_MATH=f32 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/r02_bench_line_f32mfma.json 2>/dev/null
echo "bench lines done"; cut -c1-200 $O/r02_bench_line.json
echo "[profiles] kernel stats"
# 2. kernel stats of the bench command, default and serialized
prof() {
  name=$1; shift
  rm -rf $O/prof_$name
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o run -- $B --steps 14 --warmup 8 --no-cpu-baseline --no-roofline > $O/prof_$name.log 2>&1 )
  f=$(find $O/prof_$name -name 'run_kernel_stats.csv' | head -1)
  python3 tools/prof_summary.py $f auto 90 > $O/r02_bench_kernel_summary_$name.txt 2>&1
  cp $f $O/r02_bench_kernel_stats_$name.csv
  t=$(find $O/prof_$name -name 'run_kernel_trace.csv' | head -1)
  python3 tools/trace_gaps.py $t 8 > $O/r02_trace_gaps_$name.txt 2>&1
  head -2 $O/r02_bench_kernel_summary_$name.txt
}
prof default A=1
prof serialized SCAT_SIDE_WGRAD=0 SCAT_OVERLAP_TOKENS=0 SCAT_EARLY_ADAM=0
echo "[profiles] traffic"
# 3. HBM traffic per launch: two PMC passes over the bench's own step
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- $B --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $O/pmc_$c.log 2>&1 )
done
python3 tools/traffic_json.py $(find $O/pmc_FETCH_SIZE -name run_counter_collection.csv | head -1) $(find $O/pmc_WRITE_SIZE -name run_counter_collection.csv | head -1) $O/r02_traffic.json
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/pmc_$c; done
echo "[profiles] pmc util"
# 4. matrix-pipe utilisation and clock of the contraction kernels alone on the GPU
tools/pmc_run.sh $O/pmc_util "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 tools/conv_bench.py --shapes 9,10,15,16 --reps 3 > $O/r02_pmc_util.txt 2>&1
tools/pmc_run.sh $O/pmc_wait "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" -- python3 tools/conv_bench.py --shapes 9,10,15,16 --reps 3 >> $O/r02_pmc_util.txt 2>&1
echo "[profiles] probes"
# 5. calibration probe, per-shape table, ViT tables
timeout -k 10 120 tools/_bin/mfma_probe > $O/r02_mfma_probe.txt 2>&1
PROBE_PC=1 timeout -k 10 120 tools/_bin/mfma_probe > $O/r02_mfma_probe_pc.txt 2>&1
echo "[profiles] conv shapes"
timeout -k 10 300 python tools/conv_bench.py --reps 10 > $O/r02_conv_shapes.txt 2>&1
timeout -k 10 300 python tools/conv_bench.py --reps 10 --shapes 26,27,28,29 > $O/r02_conv_shapes_hrnet.txt 2>&1
echo "[profiles] hrnet kernel summary"
rm -rf $O/prof_hrnet
( cd /tmp && SCAT_SIDE_WGRAD=0 SCAT_OVERLAP_TOKENS=0 SCAT_EARLY_ADAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hrnet -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config hrnet_w32 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/prof_hrnet.log 2>&1 )
python3 tools/prof_summary.py $(find $O/prof_hrnet -name 'run_kernel_stats.csv' | head -1) 9 70 > $O/r02_hrnet_kernel_summary_serialized.txt 2>&1
rm -rf $O/prof_hrnet
echo "[profiles] vit"
timeout -k 10 200 python tools/vit_fused_bench.py > $O/r02_vit_fused.txt 2>&1
timeout -k 10 300 python tools/vit_gemm_bench.py > $O/r02_vit_gemm.txt 2>&1
tail -2 $O/r02_conv_shapes.txt
rm -rf $O/prof_default $O/prof_serialized $O/pmc_util $O/pmc_wait
