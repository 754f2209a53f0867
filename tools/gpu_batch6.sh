#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "vit_qkv or attention" > $O/r02_t_vit.txt 2>&1
tail -3 $O/r02_t_vit.txt
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q > $O/r02_t_model.txt 2>&1
tail -3 $O/r02_t_model.txt
timeout -k 10 200 python tools/vit_fused_bench.py > $O/r02_vit_fused.txt 2>&1
cat $O/r02_vit_fused.txt
for v in 1 0; do
  SCAT_VIT_FUSED=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('vit fused $v', d['value'], d['ms_per_step'], d['config']['median_ms_per_step'])"
done
