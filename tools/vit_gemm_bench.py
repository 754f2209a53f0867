#!/usr/bin/env python
"""The dense projections of vision_transformer.Transformer(784, 3, 8, 64, 392) at batch 96 (M = 2016 tokens; SURVEY
appendix A): forward, data gradient and weight gradient of every Linear, timed per call on the split-operand kernel
(scat_gemm_split) and on the fp32-MFMA engine (scat_gemm), against the fp32 matrix peak (157.3 TF) and the
split-product peak (2500/6 = 416.7 TF).  This is the "ViT attention GEMM" roofline of BASELINE.json's north_star."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops  # noqa: E402
from scat_amd._lib import lib  # noqa: E402
from tools.conv_bench import timeit  # noqa: E402

L = lib()
M = 2016
LAYERS = [("qkv.0", 784, 1536), ("out.0", 512, 784), ("ff1.0", 784, 588), ("ff2.0", 588, 392),
          ("qkv.1", 392, 1536), ("out.1", 512, 392), ("ff1.1", 392, 294), ("ff2.1", 294, 196),
          ("qkv.2", 196, 1536), ("out.2", 512, 196)]
print(f"{'layer':7s} {'op':6s} {'K':>5s} {'N':>5s}  {'split us':>9s} {'TF':>6s} {'%fp32pk':>8s} {'%splitpk':>9s}   {'fp32 us':>8s} {'TF':>6s}  kernel")
tot = {True: 0.0, False: 0.0}
flops_attn = {True: 0.0, False: 0.0}
for name, K, N in LAYERS:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    dy = torch.randn(M, N, device="cuda")
    y, dx, dw = torch.empty(M, N, device="cuda"), torch.empty(M, K, device="cuda"), torch.empty(N, K, device="cuda")
    for op, fn in (("fwd", lambda: ops.linear_fwd(x, w, out=y)), ("dgrad", lambda: ops.linear_dgrad(dy, w, out=dx)),
                   ("wgrad", lambda: ops.linear_wgrad(dy, x, out=dw))):
        res = {}
        for split in (True, False):
            ops.GEMM_SPLIT = split
            fn()
            lab = L.scat_last_kernel().decode()
            res[split] = (timeit(fn, 20), lab)
            tot[split] += res[split][0]
        ops.GEMM_SPLIT = True
        fl = 2.0 * M * N * K
        us, lab = res[True]
        us0, _ = res[False]
        print(f"{name:7s} {op:6s} {K:5d} {N:5d}  {us:9.1f} {fl / us / 1e6:6.1f} {100 * fl / us / 1e6 / 157.3:7.1f}% "
              f"{100 * fl / us / 1e6 / 416.7:8.1f}%   {us0:8.1f} {fl / us0 / 1e6:6.1f}  {lab}", flush=True)
print(f"total per step (one fwd + dgrad + wgrad of each): split {tot[True]:.0f} us, fp32 engine {tot[False]:.0f} us")
