#!/usr/bin/env python
"""Per-shape micro-benchmark of the contraction engine on the ResNet-50 conv geometries (batch 96):
forward / data-gradient / weight-gradient, HIP-event timed, against the two rooflines
(fp32 MFMA 157.3 TF; HBM ~6.3 TB/s achievable on algorithmic bytes)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops  # noqa: E402
from scat_amd._lib import lib  # noqa: E402

SHAPES = [  # Cin, Cout, k, s, p, H
    (3, 64, 7, 2, 3, 224),
    (64, 64, 1, 1, 0, 56), (64, 64, 3, 1, 1, 56), (64, 256, 1, 1, 0, 56), (256, 64, 1, 1, 0, 56),
    (256, 128, 1, 1, 0, 56), (128, 128, 3, 2, 1, 56), (128, 512, 1, 1, 0, 28), (256, 512, 1, 2, 0, 56),
    (512, 128, 1, 1, 0, 28), (128, 128, 3, 1, 1, 28), (512, 256, 1, 1, 0, 28), (256, 256, 3, 2, 1, 28),
    (256, 1024, 1, 1, 0, 14), (512, 1024, 1, 2, 0, 28), (1024, 256, 1, 1, 0, 14), (256, 256, 3, 1, 1, 14),
    (1024, 512, 1, 1, 0, 14), (512, 512, 3, 2, 1, 14), (512, 2048, 1, 1, 0, 7), (1024, 2048, 1, 2, 0, 14),
    (2048, 512, 1, 1, 0, 7), (512, 512, 3, 1, 1, 7),
    # 23-25: the stride-2 shortcut convolutions as the step runs their forward and weight gradient: stride-1 on the input
    # packed by scat_subsample2 (rows 8, 14, 20 are the same layers called unpacked: only their data gradient is on the step)
    (256, 512, 1, 1, 0, 28), (512, 1024, 1, 1, 0, 14), (1024, 2048, 1, 1, 0, 7),
    # 26-29: HRNet-W32's branch convolutions (models/hrnet.py:38-63)
    (32, 32, 3, 1, 1, 56), (64, 64, 3, 1, 1, 28), (128, 128, 3, 1, 1, 14), (256, 256, 3, 1, 1, 7),
]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=96)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="", help="comma list of op kinds: fwd,dgrad,wgrad")
    ap.add_argument("--shapes", default="", help="comma list of shape indices")
    a = ap.parse_args()
    kinds = a.only.split(",") if a.only else ["fwd", "dgrad", "wgrad"]
    idx = [int(i) for i in a.shapes.split(",")] if a.shapes else range(26)
    B = a.batch
    L = lib()
    print(f"{'shape':34s} {'op':6s} {'kernel':42s} {'us':>8s} {'TF':>7s} {'mfma_us':>8s} {'hbm_us':>7s}")
    tot = {k: [0.0, 0.0] for k in kinds}
    for i in idx:
        cin, cout, k, s, p, H = SHAPES[i]
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        y = ops.conv2d_fwd(x, w, s, p)
        dy = torch.randn_like(y)
        wt = ops.conv2d_wt(w) if k != 7 else None
        flops = 2.0 * y.numel() * cin * k * k
        byts = 4.0 * (x.numel() + y.numel() + w.numel())
        name = f"{cin}->{cout} k{k} s{s} {H}x{H}"
        for kind in kinds:
            # what a ResNet-50 step launches: the 1x1 / stride-2 shortcuts run forward and weight gradient packed (rows
            # 23-25) and only the data gradient in this geometry; the packed rows have no data gradient of their own
            on_step = not ((k == 1 and s == 2 and kind != "dgrad") or (i in (23, 24, 25) and kind == "dgrad"))
            if kind == "fwd":
                fn = lambda: ops.conv2d_fwd(x, w, s, p, out=y)
            elif kind == "dgrad":
                if k == 7:
                    continue
                dx = torch.empty_like(x)
                fn = lambda: ops.conv2d_dgrad_w(dy, w, tuple(x.shape), s, p, out=dx)
            else:
                dw = torch.empty_like(w)
                fn = lambda: ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p, out=dw)
            us = timeit(fn, a.reps)
            label = L.scat_last_kernel().decode()
            print(f"{name:34s} {kind:6s} {label:42s} {us:8.1f} {flops / us / 1e6:7.1f} {flops / 157.3e6:8.1f} "
                  f"{byts / 6.3e6:7.1f}{'' if on_step else '   (not on the step)'}", flush=True)
            if on_step:
                tot[kind][0] += us
                tot[kind][1] += flops
    for kind, (us, fl) in tot.items():
        if us:
            print(f"TOTAL {kind} (rows on the step, one call each): {us / 1e3:.2f} ms, {fl / us / 1e6:.1f} TF")


if __name__ == "__main__":
    main()
