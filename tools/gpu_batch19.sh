#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q 2>&1 | tail -4
for v in 1 0; do
  SCAT_WG_ROWS=$v timeout -k 10 300 python bench.py --config hrnet_w32 --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hrnet rows=$v', d['value'], d['ms_per_step'])"
  SCAT_WG_ROWS=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('resnet rows=$v', d['value'], d['ms_per_step'], d['config']['median_ms_per_step'])"
done
