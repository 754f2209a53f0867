#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad3x3_rows" 2>&1 | tail -4
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_schedule.py -x -q 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('resnet', d['value'], d['ms_per_step'], d['config']['median_ms_per_step'])"
done
