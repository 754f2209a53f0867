#!/bin/bash
# usage: tools/pmc_run.sh OUTDIR "COUNTERS" -- python3 tools/conv_bench.py ...   (on the GPU box; one --pmc pass)
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the copy of the repository there)}"
set -e
out=$1; shift; ctr=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out" -o run -- "$@" > "$out/stdout.txt" 2>&1
# rocprofv3 nests its output under a host/pid directory: flatten
f=$(find "$out" -name 'run_counter_collection.csv' | head -1)
if [ -n "$f" ] && [ "$(dirname "$f")" != "$out" ]; then cp "$(dirname "$f")"/run_*.csv "$out"/; fi
python3 tools/pmc_table.py "$out"
