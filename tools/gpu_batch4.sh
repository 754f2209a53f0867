#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r02_gputests.txt 2>&1
tail -4 $O/r02_gputests.txt
for c in hrnet_w32 performer; do
  timeout -k 10 400 python bench.py --config $c --steps 20 --warmup 5 > $O/r02_bench_$c.txt 2>$O/r02_bench_$c.err
  tail -1 $O/r02_bench_$c.txt | cut -c1-300
done
rm -rf $O/prof_hrnet
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hrnet -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config hrnet_w32 --steps 8 --warmup 4 --no-roofline > $O/prof_hrnet.log 2>&1 )
f=$(find $O/prof_hrnet -name 'run_kernel_stats.csv' | head -1)
python3 tools/prof_summary.py $f 12 60 > $O/r02_kernel_summary_hrnet.txt 2>&1
head -30 $O/r02_kernel_summary_hrnet.txt
