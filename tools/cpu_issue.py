#!/usr/bin/env python
"""Is a train step bound by the host?  Issues N steps without synchronising and reports the host's time to issue them next to
the time until the device has finished them (the host runs ahead when the device is the bound; equal times = host-bound)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "reg_transformer"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
net = bench.make_net(cfg, 1, dev)
step = bench.Step(cfg, net, dev)
u8, lab = bench.build_inputs(96, 100, dev)
for _ in range(4):
    step(u8, lab)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(n):
        step(u8, lab)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{cfg}: host issue {1e3 * (t1 - t0) / n:.2f} ms/step, device done {1e3 * (t2 - t0) / n:.2f} ms/step "
          f"(host ahead by {1e3 * (t2 - t1):.1f} ms after {n} steps)")
