#!/usr/bin/env python
"""Probe the engine: square GEMM and 1x1-conv efficiency as a function of the contraction length."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops
from scat_amd._lib import lib
from tools.conv_bench import timeit

L = lib()
for n in (2048, 4096):
    x = torch.randn(n, n, device="cuda"); w = torch.randn(n, n, device="cuda"); y = torch.empty(n, n, device="cuda")
    us = timeit(lambda: ops.linear_fwd(x, w, out=y), 5)
    print(f"gemm {n}^3 {L.scat_last_kernel().decode():36s} {us:9.1f} us {2*n**3/us/1e6:7.1f} TF", flush=True)
B, H = 96, 28
for cout in (128, 256):
    for cin in (64, 128, 256, 512, 1024, 2048, 4096):
        x = torch.randn(B, cin, H, H, device="cuda"); w = torch.randn(cout, cin, 1, 1, device="cuda") * .05
        y = ops.conv2d_fwd(x, w, 1, 0)
        us = timeit(lambda: ops.conv2d_fwd(x, w, 1, 0, out=y), 5)
        fl = 2.0 * y.numel() * cin
        print(f"conv1x1 {cin:5d}->{cout} @28 {L.scat_last_kernel().decode():30s} {us:9.1f} us {fl/us/1e6:7.1f} TF  (mfma {fl/157.3e6:7.1f} us, hbm {(x.numel()+y.numel())*4/6.3e6:6.1f} us)", flush=True)
