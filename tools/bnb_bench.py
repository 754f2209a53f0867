#!/usr/bin/env python
"""The folded bn3 backward's three kernels per bottleneck shape, alone on the GPU: bn_bwd_pre (masking reduce),
conv1x1_dgrad_bnb and conv1x1_wgrad_bnb (dual-source operands), with the HBM time of the bytes they must move."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops  # noqa: E402
from tools.conv_bench import timeit  # noqa: E402

B = 96
print(f"{'shape':22s} {'pre us':>8s} {'(hbm)':>7s} {'dgrad us':>9s} {'(hbm)':>7s} {'wgrad us':>9s} {'(hbm)':>7s}")
for cin, cout, H in ((64, 256, 56), (128, 512, 28), (256, 1024, 14)):      # (7x7 planes: HW % 4 != 0, unfolded path)
    dcur = torch.randn(B, cout, H, H, device="cuda")
    c3 = torch.randn(B, cout, H, H, device="cuda")
    c2 = torch.randn(B, cin, H, H, device="cuda")
    w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
    sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    mean, invstd, gamma = torch.randn(cout, device="cuda"), torch.rand(cout, device="cuda") + 0.5, torch.rand(cout, device="cuda")
    sc2, sh2 = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    out = torch.relu(c3 * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    _, mask = ops.bn_apply(c3, sc, sh, None, True, want_mask=True)
    g = dcur.clone()
    coef3, dg, db = ops.bn_bwd_pre(g, c3, True, sc, sh, mean, invstd, gamma, None, None, y_mask=mask)
    T = dcur.numel() * 4
    t = c2.numel() * 4
    us_pre = timeit(lambda: ops.bn_bwd_pre(g, c3, True, sc, sh, mean, invstd, gamma, None, None, y_mask=mask), 10)
    da = torch.empty_like(c2)
    us_d = timeit(lambda: ops.conv1x1_dgrad_bnb(g, c3, coef3, w, tuple(c2.shape), out=da), 10)
    dw = torch.empty_like(w)
    us_w = timeit(lambda: ops.conv1x1_wgrad_bnb(g, c3, coef3, c2, tuple(w.shape), sc2, sh2, True, out=dw), 10)
    hb = lambda nbytes: nbytes / 4.5e6      # us at 4.5 TB/s
    print(f"{cout:4d}->{cin:4d} @{H:2d}x{H:<2d}      {us_pre:8.1f} {hb(3 * T + T / 32):7.1f} {us_d:9.1f} {hb(2 * T + t):7.1f} "
          f"{us_w:9.1f} {hb(2 * T + t):7.1f}", flush=True)
