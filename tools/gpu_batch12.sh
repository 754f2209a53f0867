#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "stem or conv_fwd_dgrad_wgrad" 2>&1 | tail -3
for v in 1 0; do SCAT_STEM_SPLIT=$v timeout -k 10 100 python tools/conv_bench.py --reps 10 --only wgrad --shapes 0 2>&1 | grep "k7"; done
for v in 1 0 1 0; do
  SCAT_STEM_SPLIT=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stem split $v', d['value'], d['ms_per_step'], d['config']['median_ms_per_step'])"
done
