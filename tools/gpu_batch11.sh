#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3_halo" 2>&1 | tail -2
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -x -q -k "hrnet" 2>&1 | tail -2
for v in 1 0; do
  SCAT_C3_M32=$v timeout -k 10 300 python bench.py --config hrnet_w32 --steps 20 --warmup 5 --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('m32=$v', d['value'], d['ms_per_step'])"
done
