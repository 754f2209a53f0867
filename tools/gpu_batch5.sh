#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "stem or conv_fwd_dgrad" > $O/r02_t_stem.txt 2>&1
tail -3 $O/r02_t_stem.txt
for v in 1 0; do SCAT_STEM_SPLIT=$v timeout -k 10 100 python tools/conv_bench.py --reps 10 --only fwd --shapes 0 2>&1 | grep "k7"; done
for v in 1 0; do
  SCAT_STEM_SPLIT=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stem split $v', d['value'], d['ms_per_step'], d['config']['median_ms_per_step'])"
done
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_schedule.py -x -q > $O/r02_t_model.txt 2>&1
tail -3 $O/r02_t_model.txt
