#!/usr/bin/env python
"""Which roles of the train step share a hardware queue (scat_amd.streams.sharing) after a few steps, with and without
the data-parallel collectives (run with RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=... SCAT_DP_FORCE_COLLECTIVES=1
for the one-rank RCCL group)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from scat_amd import streams  # noqa: E402

from scat_amd.dp import init_distributed  # noqa: E402
rank, local, world = init_distributed()
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
net = bench.make_net("resnet50", 1, dev)
step = bench.Step("resnet50", net, dev)
u8, lab = bench.build_inputs(96, 100, dev)
for _ in range(4):
    step(u8, lab)
torch.cuda.synchronize()
print("collectives:", step.ts.buckets.collective, " roles:", list(streams.bound(dev)))
for a, b, why in streams.sharing(dev):
    print(f"  {a:>13s} + {b:<13s} {why}")
