#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -x -q -k "hrnet" 2>&1 | tail -2
for v in 0 1 0 1; do
SCAT_HRNET_EPI=$v timeout -k 10 300 python bench.py --config hrnet_w32 --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hrnet epi=$v', d['value'], d['ms_per_step'])"
done
