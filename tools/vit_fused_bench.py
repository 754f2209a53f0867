#!/usr/bin/env python
"""The fused qkv + attention launch of a transformer layer against the three launches it replaces (qkv projection on the
contraction engine, attention core), batch 96, 21 tokens, the three layer widths of vision_transformer.Transformer
(784, 392, 196); HIP-event timed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops  # noqa: E402


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


B, n, heads = 96, 21, 8
print(f"{'dim':>5s} {'fused us':>9s} {'TF':>6s} | {'qkv us':>7s} {'attn us':>8s} {'sum':>7s} | speed-up")
for dim in (784, 392, 196):
    h = torch.randn(B * n, dim, device="cuda")
    w = torch.randn(1536, dim, device="cuda") * dim ** -0.5
    fl = 2.0 * B * n * 1536 * dim
    tf = timeit(lambda: ops.qkv_attention_fwd(h, w, B, n, heads, 0.125))
    q = ops.linear_fwd(h, w)
    t1 = timeit(lambda: ops.linear_fwd(h, w, out=q))
    t2 = timeit(lambda: ops.attention_fwd(q.view(B, n, 1536), heads, 64, 0.125))
    print(f"{dim:5d} {tf:9.1f} {fl / tf / 1e6:6.1f} | {t1:7.1f} {t2:8.1f} {t1 + t2:7.1f} | {(t1 + t2) / tf:.2f}x")
