#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "conv1x1 or conv_fwd_dgrad" 2>&1 | tail -1
for rep in 1 2; do for e in "A=1" "SCAT_PC=0"; do
  env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'], d['config']['median_ms_per_step'])"
done; done
