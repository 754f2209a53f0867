#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r02_gputests_final.txt 2>&1
tail -3 $O/r02_gputests_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
tools/gpu_profiles.sh
