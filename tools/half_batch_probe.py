#!/usr/bin/env python
"""Upper bound for a half-batch schedule: does the GPU finish two INDEPENDENT batch-48 train steps issued on two streams
sooner than one batch-96 step?  (Two separate networks: no shared BatchNorm statistics — what a staggered two-half
schedule of ONE network could at best approach.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, n):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


net = bench.make_net("resnet50", 1, dev)
step = bench.Step("resnet50", net, dev)
for B in (96, 48):
    u8, lab = bench.build_inputs(B, 100, dev)
    ms = timed(lambda: step(u8, lab), 20)
    print(f"one network, batch {B}: {ms:.2f} ms/step = {1e3 * B / ms:.0f} img/s", flush=True)
net2 = bench.make_net("resnet50", 2, dev)
step2 = bench.Step("resnet50", net2, dev)
u8a, laba = bench.build_inputs(48, 100, dev)
u8b, labb = bench.build_inputs(48, 101, dev)
s2 = torch.cuda.Stream()


def both():
    step(u8a, laba)
    with torch.cuda.stream(s2):
        step2(u8b, labb)


s2.wait_stream(torch.cuda.current_stream())
ms = timed(both, 20)
print(f"two networks, batch 48 each, two streams: {ms:.2f} ms per pair = {1e3 * 96 / ms:.0f} img/s", flush=True)
