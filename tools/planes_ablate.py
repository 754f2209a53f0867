#!/usr/bin/env python
"""Timing ablation of the planes pointwise kernel (csrc/planes.hip pw_planes_kernel<4,128,2,DIAG>, tools build only:
SCAT_LIBPATH=tools/_bin/libscat_hip_diag.so): one ingredient removed at a time, results wrong, timing only.
DIAG bits: 1 no activation DMA, 2 no weight loads, 4 no MFMAs, 8 no output stores, 16 no stage barrier, 32 no LDS reads."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops  # noqa: E402
from tools.planes_bench import bench  # noqa: E402

SHAPES = [(512, 256, 28), (256, 1024, 14), (1024, 512, 14), (128, 512, 28), (2048, 512, 7)]
VARIANTS = [("full", 100), ("loads spread", 164), ("no DMA", 101), ("no weight loads", 102), ("no loads", 103), ("no MFMA", 104),
            ("no stores", 108), ("no loads/stores", 111), ("no barrier", 116), ("no LDS reads", 132),
            ("MFMA only (no loads/stores/barrier/LDS reads)", 159)]


def main():
    B = 96
    print("shape".ljust(20) + "".join(n[:16].rjust(17) for n, _ in VARIANTS))
    for cin, cout, H in SHAPES:
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
        y = torch.empty(B, cout, H, H, device="cuda")
        xp = ops.planes_from(x)
        wp = ops.WeightPrep()
        ops.conv2d_fwd(x, w, 1, 0, wp=wp)
        wp.run(True)
        y0 = ops.conv1x1_planes(xp, w, wp=wp, lds_stages=100).clone()
        same = torch.equal(ops.conv1x1_planes(xp, w, wp=wp, lds_stages=164), y0)
        fns = [(lambda c=c: ops.conv1x1_planes(xp, w, out=y, wp=wp, lds_stages=c)) for _, c in VARIANTS]
        us = bench(fns, 5, 5)
        print(f"{cin}->{cout}@{H}".ljust(20) + "".join(f"{u:17.1f}" for u in us) + f"   spread == full: {same}", flush=True)


if __name__ == "__main__":
    main()
