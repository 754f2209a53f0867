#!/usr/bin/env python
"""Where does the HOST's time go in a train step?  cProfile over a few steps with autograd's backward on the calling thread
(so that the profile sees it), sorted by own time."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
net = bench.make_net(cfg, 1, dev)
step = bench.Step(cfg, net, dev)
u8, lab = bench.build_inputs(96, 100, dev)
torch.autograd.set_multithreading_enabled(False)
for _ in range(4):
    step(u8, lab)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step(u8, lab)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print(f"{cfg}: {n} steps; times below are totals over them (divide by {n})")
st.print_stats(45)
st.sort_stats("cumtime")
st.print_stats(35)
