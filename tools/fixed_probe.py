#!/usr/bin/env python
"""Fixed-cost probe: 1x1 conv with a tiny contraction (time ~ prologue + epilogue)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops
from scat_amd._lib import lib
from tools.conv_bench import timeit
L = lib()
B = 96
for H, cout in ((56, 256), (56, 128), (28, 256), (14, 256), (14, 1024)):
    for cin in (16, 64, 256):
        x = torch.randn(B, cin, H, H, device="cuda"); w = torch.randn(cout, cin, 1, 1, device="cuda") * .05
        y = ops.conv2d_fwd(x, w, 1, 0)
        us = timeit(lambda: ops.conv2d_fwd(x, w, 1, 0, out=y), 10)
        fl = 2.0 * y.numel() * cin
        print(f"conv1x1 {cin:4d}->{cout:4d} @{H:2d} {L.scat_last_kernel().decode():32s} {us:8.1f} us  out {y.numel()*4/1e6:6.1f} MB -> {y.numel()*4/us/1e6:5.2f} TB/s-out  mfma {fl/157.3e6:6.1f} us", flush=True)
# plain device copy of the same size for reference
y = torch.empty(96 * 256 * 56 * 56, device="cuda"); z = torch.empty_like(y)
us = timeit(lambda: z.copy_(y), 10)
print(f"torch copy 308 MB: {us:.1f} us = {2*y.numel()*4/us/1e6:.2f} TB/s (r+w)")
us = timeit(lambda: z.fill_(1.0), 10)
print(f"torch fill 308 MB: {us:.1f} us = {y.numel()*4/us/1e6:.2f} TB/s (w)")
