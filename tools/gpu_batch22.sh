#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for v in 1 0; do
  SCAT_EPI_STATS=$v timeout -k 10 300 python bench.py --config hrnet_w32 --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hrnet epi=$v', d['value'], d['ms_per_step'])"
done
timeout -k 10 300 python bench.py --config performer --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('performer', d['value'], d['ms_per_step'])"
