#!/bin/bash
# round-4 evidence, part A: block-parity tests, planes-vs-split PMC rows, FETCH_SIZE calibration, attention timing
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "bottleneck_batch96" > $O/r04_block_tests.txt 2>&1; echo "block tests rc=$?"; tail -4 $O/r04_block_tests.txt
echo "[pmc] planes vs split"
tools/pmc_run.sh $O/pmc_pl_util "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 tools/planes_pmc.py > $O/r04_pmc_planes.txt 2>&1
tools/pmc_run.sh $O/pmc_pl_wait "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" -- python3 tools/planes_pmc.py >> $O/r04_pmc_planes.txt 2>&1
rm -rf $O/pmc_pl_util $O/pmc_pl_wait
tail -30 $O/r04_pmc_planes.txt
echo "[pmc] FETCH_SIZE calibration"
tools/pmc_run.sh $O/pmc_calib "FETCH_SIZE" -- python3 tools/fetch_calib.py > $O/r04_fetch_calib.log 2>&1
python3 tools/fetch_calib.py --report $O/pmc_calib/run_counter_collection.csv > $O/r04_fetch_calib.txt 2>&1; cat $O/r04_fetch_calib.txt
rm -rf $O/pmc_calib
echo "[hrnet bench]"
timeout -k 10 300 python bench.py --config hrnet_w32 --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_line_hrnet_w32_a.json 2>/dev/null; cut -c1-150 $O/r04_bench_line_hrnet_w32_a.json
