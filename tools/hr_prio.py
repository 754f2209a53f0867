#!/usr/bin/env python
"""HRNet-W32 step with the caller's stream at high priority (the branch streams stay at normal priority): does the critical
chain (stem, layer1, branch 0, transitions, head: ~34 of the ~67 ms of kernel time) finish sooner when its kernels win the
dispatch?  Interleaved A/B in one process."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w32"
dev = torch.device("cuda", 0)
net = bench.make_net(cfg, 1, dev)
step = bench.Step(cfg, net, dev)
u8, lab = bench.build_inputs(96, 100, dev)
lo, hi = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)
print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")


def run(st, n):
    with torch.cuda.stream(st):
        for _ in range(3):
            step(u8, lab)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(u8, lab)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n


for _ in range(4):
    step(u8, lab)
for rep in range(3):
    print(f"{cfg}: default stream {run(torch.cuda.default_stream(), 12):.2f}  normal-priority stream {run(lo, 12):.2f}  "
          f"high-priority stream {run(hi, 12):.2f} ms/step", flush=True)
