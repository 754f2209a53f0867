#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python tools/host_probe_cfg.py hrnet_w32 10 2>&1 | tail -1
timeout -k 10 200 python tools/host_probe_cfg.py resnet50 20 2>&1 | tail -1
timeout -k 10 300 python bench.py --config hrnet_w32 --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hrnet', d['value'], d['ms_per_step'])"
timeout -k 10 250 python tools/host_profile.py hrnet_w32 > gpurun_out/host_profile_hrnet.txt 2>&1
