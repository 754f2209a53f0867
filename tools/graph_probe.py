#!/usr/bin/env python
"""Feasibility probe: capture one whole train step (forward + backward + update) in a HIP graph and time its replay against
the eager step.  Timing only — the mask draw is switched off and Adam's bias correction is frozen at the captured step."""
import faulthandler
import os
import sys
import time

import torch

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from scat_amd import graphed  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
net = bench.make_net(cfg, 1, dev)
net.mask_rate = 0.0
step = bench.Step(cfg, net, dev)
u8, lab = bench.build_inputs(96, 100, dev)
gs = graphed.GraphedStep(lambda: step(u8, lab), warmup=4 + n)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for _ in range(4):
    gs()
print(f"{cfg}: eager (own stream) {timed(gs, n):.2f} ms/step", flush=True)
print("capturing", flush=True)
out = gs()
torch.cuda.synchronize()
print("captured + first replay; loss", float(out[0]), flush=True)
for _ in range(3):
    print(f"{cfg}: graph replay {timed(gs, n):.2f} ms/step", flush=True)
