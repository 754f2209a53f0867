#!/bin/bash
# what issues the copyBuffer launches of the HRNet-W32 step (hip-api trace of a short run)
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rm -rf $O/prof_hip
( cd /tmp && rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_hip -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config hrnet_w32 --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_hip.log 2>&1 )
ls $O/prof_hip/*/ 2>/dev/null | head; find $O/prof_hip -name '*.csv' | head
f=$(find $O/prof_hip -name 'run_hip_api_trace.csv' | head -1)
python3 tools/copy_origin.py $f > $O/r04_copy_origin.txt 2>&1
m=$(find $O/prof_hip -name 'run_memory_copy_trace.csv' | head -1)
if [ -n "$m" ]; then echo "memory copies:" >> $O/r04_copy_origin.txt; python3 - "$m" >> $O/r04_copy_origin.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
c = collections.Counter()
for r in rows:
    d = r.get("Direction", r.get("Name", "?"))
    c[d] += 1
print(dict(c))
print(rows[0].keys() if rows else None)
PY
fi
cat $O/r04_copy_origin.txt | cut -c1-400
rm -rf $O/prof_hip
