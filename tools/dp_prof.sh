#!/bin/bash
# one-rank nccl group with the collectives forced (SCAT_DP_FORCE_COLLECTIVES=1): what does the data-parallel path add to a step?
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("median_ms_per_step"))'
for f in 1 0 1 0; do echo "forced collectives=$f:"; SCAT_DP_FORCE_COLLECTIVES=$f timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>/dev/null | python -c "$P"; done
SCAT_DP_FORCE_COLLECTIVES=1 timeout -k 10 200 python tools/dp_queues.py 2>&1 | grep -v "amdgpu.ids\|pretrained\|Warning\|warn"
SCAT_DP_FORCE_COLLECTIVES=0 timeout -k 10 200 python tools/dp_queues.py 2>&1 | grep -v "amdgpu.ids\|pretrained\|Warning\|warn"
