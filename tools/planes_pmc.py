#!/usr/bin/env python
"""Workload for the PMC passes that compare the in-kernel-split pointwise kernel with the planes / LDS-DMA one on the same
shapes (tools/pmc_run.sh ... -- python3 tools/planes_pmc.py): every kernel is launched three times, tools/pmc_table.py
reports the second launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops  # noqa: E402

B = 96
for cin, cout, H in [(512, 256, 28), (256, 1024, 14), (1024, 512, 14), (128, 512, 28)]:
    x = torch.randn(B, cin, H, H, device="cuda")
    w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    y = torch.empty(B, cout, H, H, device="cuda")
    xp = ops.planes_from(x, sc, sh, True)
    wp = ops.WeightPrep()
    ops.conv2d_fwd(x, w, 1, 0, wp=wp)
    wp.run(True)
    for _ in range(3):
        ops.conv2d_fwd(x, w, 1, 0, out=y, wp=wp)
    for _ in range(3):
        ops.conv2d_fwd(x, w, 1, 0, sc, sh, True, out=y, wp=wp)
    for _ in range(3):
        ops.conv1x1_planes(xp, w, out=y, wp=wp, lds_stages=2)
    for _ in range(3):
        ops.planes_from(x, sc, sh, True, out=xp.buf)
    torch.cuda.synchronize()
    print(f"{cin}->{cout}@{H} done", flush=True)
