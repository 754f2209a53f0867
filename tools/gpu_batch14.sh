#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "conv or wgrad" 2>&1 | tail -3
timeout -k 10 200 python tools/conv_bench.py --reps 10 --only wgrad 2>&1 | grep -E "wgrad|TOTAL"
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['config']['median_ms_per_step'])"
done
