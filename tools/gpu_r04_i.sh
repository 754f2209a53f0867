#!/bin/bash
# new colsum-group / LayerNorm tests, host-issue time of the two steps, HRNet steady-state trace
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "colsum or layernorm" 2>&1 | tail -3
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 200 python tools/cpu_issue.py reg_transformer 8 > $O/r04_cpu_issue.txt 2>&1
timeout -k 10 200 python tools/cpu_issue.py hrnet_w32 6 >> $O/r04_cpu_issue.txt 2>&1
grep "ms/step" $O/r04_cpu_issue.txt
rm -rf $O/prof_h
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/prof_h -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config hrnet_w32 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline > $O/prof_h.log 2>&1 )
t=$(find $O/prof_h -name 'run_kernel_trace.csv' | head -1)
python3 tools/trace_gaps.py $t 4 70 > $O/r04_hrnet_trace_gaps_default.txt 2>&1
head -3 $O/r04_hrnet_trace_gaps_default.txt
rm -rf $O/prof_h
timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 | cut -c1-200
