#!/usr/bin/env python
"""Feasibility probe: capture one whole train step (forward + backward + update) in a HIP graph and time its replay against
the eager step.  Timing only — the mask draw is switched off and Adam's bias correction is frozen at the captured step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
net = bench.make_net(cfg, 1, dev)
net.mask_rate = 0.0
step = bench.Step(cfg, net, dev)
u8, lab = bench.build_inputs(96, 100, dev)
for _ in range(5):
    step(u8, lab)
torch.cuda.synchronize()


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print(f"{cfg}: eager {timed(lambda: step(u8, lab), n):.2f} ms/step", flush=True)
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    out = step(u8, lab)
print(f"captured in {time.perf_counter() - t0:.2f} s", flush=True)
g.replay()
torch.cuda.synchronize()
print("first replay done; loss", float(out[0]), flush=True)
for _ in range(3):
    print(f"{cfg}: graph replay {timed(g.replay, n):.2f} ms/step", flush=True)
print(f"{cfg}: eager again {timed(lambda: step(u8, lab), n):.2f} ms/step", flush=True)
