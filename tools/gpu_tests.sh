#!/bin/bash
# the whole -m gpu suite on the GPU box, output under gpurun_out/ (progress lines on stdout every test file)
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
N=${1:-r03}
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=15 "${@:2}" > $O/${N}_gputests.txt 2>&1
rc=$?
tail -25 $O/${N}_gputests.txt
exit $rc
