#!/usr/bin/env python
"""Diag build, SCAT_TUNE=232: every wavefront of the second-generation pointwise weight gradient reports the cycles of
its stage loop and how many of them it spent parked at the stage barrier — which side (consumers = MFMA, producers =
staging) the other one waits for."""
import os, sys
os.environ["SCAT_TUNE"] = "232"
os.environ.setdefault("SCAT_LIBPATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libscat_hip_diag.so"))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops
B = 96
for cin, cout, H in ((512, 256, 28), (1024, 512, 14), (128, 512, 28)):
    x = torch.randn(B, cin, H, H, device="cuda"); dy = torch.randn(B, cout, H, H, device="cuda")
    for _ in range(3):
        ops.conv2d_wgrad(dy, x, (cout, cin, 1, 1), 1, 0)
    torch.cuda.synchronize()
    lab = ops.lib().scat_last_kernel().decode()
    import re
    m = re.match(r"wgrad1x1_pw_(\d+)x(\d+)x32.*_split(\d+)", lab)
    RA, RB, splits = int(m.group(1)), int(m.group(2)), int(m.group(3))
    ws = ops.workspace(1, x.device).cpu().numpy().view(np.float32)[: splits * cout * cin].reshape(splits, cout, cin)
    nc = (RA // 64) * (RB // 64)
    cons, prod = [], []
    for z in range(splits):
        for i0 in range(0, cout, RA):
            for j0 in range(0, cin, RB):
                for w in range(nc + 4):
                    v = ws[z, i0 + w, j0:j0 + 4].copy().view(np.uint64)
                    (cons if w < nc else prod).append((int(v[0]), int(v[1])))
    c, p = np.array(cons, dtype=np.float64), np.array(prod, dtype=np.float64)
    c, p = c[(c[:, 0] > 0) & (c[:, 0] < 1e7)], p[(p[:, 0] > 0) & (p[:, 0] < 1e7)]
    print(f"{cin}->{cout} @{H}: {lab}")
    print(f"   consumers: loop {np.median(c[:, 0]) / 100:.1f} us (100 MHz ticks), parked at the barrier {np.median(c[:, 1] / c[:, 0]):.2f} of it")
    print(f"   producers: loop {np.median(p[:, 0]) / 100:.1f} us, parked at the barrier {np.median(p[:, 1] / p[:, 0]):.2f} of it")
