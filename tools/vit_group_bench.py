#!/usr/bin/env python
"""The twelve weight-gradient contractions of the token mixer's backward at batch 96 (2016 tokens): one by one
(scat_gemm each + its split-K reduce) against ONE grouped launch (scat_gemm_group), fp32 MFMA engine, peak 157.3 TF."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops  # noqa: E402

M = 2016
DIMS = []
for dim in (784, 392, 196):
    last = dim == 196
    DIMS += [("qkv", 1536, dim), ("out", dim, 512), ("ff1", dim * 3 // 4, dim), ("ff2", 3 if last else dim // 2, dim * 3 // 4)]


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


pairs = [(torch.randn(M, N, device="cuda"), torch.randn(M, K, device="cuda")) for _, N, K in DIMS]
flops = [2.0 * M * N * K for _, N, K in DIMS]
tot = 0.0
print(f"{'gemm':6s} {'N':>5s} {'K':>5s} {'us':>8s} {'TF':>7s}  kernel")
for (name, N, K), (dy, x), fl in zip(DIMS, pairs, flops):
    us = timeit(lambda: ops.linear_wgrad(dy, x))
    tot += us
    print(f"{name:6s} {N:5d} {K:5d} {us:8.1f} {fl / us / 1e6:7.1f}  {ops.lib().scat_last_kernel().decode()}")
print(f"one by one: {tot:.1f} us, {sum(flops) / tot / 1e6:.1f} TF aggregate = {sum(flops) / tot / 1e6 / 157.3:.2f} of the fp32 MFMA peak")
us = timeit(lambda: ops.linear_wgrad_group(pairs))
print(f"grouped   : {us:.1f} us, {sum(flops) / us / 1e6:.1f} TF aggregate = {sum(flops) / us / 1e6 / 157.3:.2f} of the fp32 MFMA peak  "
      f"({ops.lib().scat_last_kernel().decode()})")
