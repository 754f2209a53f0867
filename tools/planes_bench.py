#!/usr/bin/env python
"""Pointwise contraction from pre-split bf16 planes (csrc/planes.hip) against the in-kernel split (csrc/conv1x1.hip) on
the ResNet-50 pointwise geometries at batch 96: forward with the fused BatchNorm + ReLU (conv3 of a block: the operand
is relu(bn2(c2))) and the plain forward / data gradient (conv1, both directions), interleaved in ONE process
(cdna_hip_programming.md 5.4 rule 24), HIP-event timed; plus the producing pass (scat_planes_from_f32)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops  # noqa: E402
from scat_amd._lib import lib  # noqa: E402

SHAPES = [  # Cin, Cout, H
    (64, 64, 56), (64, 256, 56), (256, 64, 56), (256, 128, 56), (128, 512, 28), (512, 128, 28), (512, 256, 28),
    (256, 1024, 14), (1024, 256, 14), (1024, 512, 14), (512, 2048, 7), (2048, 512, 7),
    (256, 512, 28), (512, 1024, 14), (1024, 2048, 7),
]


def bench(fns, reps, rounds):
    """interleaved rounds; -> median us per variant"""
    for f in fns:
        f()
    torch.cuda.synchronize()
    ts = [[] for _ in fns]
    for _ in range(rounds):
        for k, f in enumerate(fns):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / reps * 1e3)
    return [sorted(t)[len(t) // 2] for t in ts]


def flops_of(B, H, cin, cout):
    return 2.0 * B * H * H * cin * cout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=96)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--shapes", default="")
    ap.add_argument("--s16", action="store_true", help="also the 16x16x32 experiment (diag build: SCAT_LIBPATH)")
    a = ap.parse_args()
    B = a.batch
    idx = [int(i) for i in a.shapes.split(",")] if a.shapes else range(len(SHAPES))
    print(f"{'shape':24s} {'split':>8s} {'split_tf':>8s} {'pl_nb2':>8s} {'pl_nb3':>8s} {'TF_nb2':>7s} {'TF_nb3':>7s} "
          f"{'make':>7s} {'make_tf':>7s} {'mk TB/s':>7s}   (us; TF = algorithmic fp32 TFLOP/s)")
    tot = [0.0] * 6
    for i in idx:
        cin, cout, H = SHAPES[i]
        x = torch.randn(B, cin, H, H, device="cuda")
        w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
        y = torch.empty(B, cout, H, H, device="cuda")
        xp = ops.planes_from(x, sc, sh, True)
        wp = ops.WeightPrep()
        ops.conv2d_fwd(x, w, 1, 0, wp=wp)
        wp.run(True)
        fns = [
            lambda: ops.conv2d_fwd(x, w, 1, 0, out=y, wp=wp),
            lambda: ops.conv2d_fwd(x, w, 1, 0, sc, sh, True, out=y, wp=wp),
            lambda: ops.conv1x1_planes(xp, w, out=y, wp=wp, lds_stages=2),
            lambda: ops.conv1x1_planes(xp, w, out=y, wp=wp, lds_stages=3),
            lambda: ops.planes_from(x, out=xp.buf),
            lambda: ops.planes_from(x, sc, sh, True, out=xp.buf),
        ]
        if a.s16 and cout > 64:
            y2 = torch.empty_like(y)
            ops.conv1x1_planes(xp, w, out=y, wp=wp, lds_stages=2)
            ops.conv1x1_planes(xp, w, out=y2, wp=wp, lds_stages=12)
            err = float((y - y2).abs().max() / y.abs().max())
            fns += [lambda: ops.conv1x1_planes(xp, w, out=y, wp=wp, lds_stages=12),
                    lambda: ops.conv1x1_planes(xp, w, out=y, wp=wp, lds_stages=13)]
        us = bench(fns, a.reps, a.rounds)
        if len(us) > 6:
            print(f"      16x16x32: nb2 {us[6]:8.1f} us ({flops_of(B, H, cin, cout) / us[6] / 1e6:6.1f} TF)  nb3 {us[7]:8.1f} us "
                  f"({flops_of(B, H, cin, cout) / us[7] / 1e6:6.1f} TF)   max rel diff vs 32x32x16: {err:.2e}")
        flops = 2.0 * B * H * H * cin * cout
        print(f"{cin:5d}->{cout:<5d}@{H:<3d}        {us[0]:8.1f} {us[1]:8.1f} {us[2]:8.1f} {us[3]:8.1f} "
              f"{flops / us[2] / 1e6:7.1f} {flops / us[3] / 1e6:7.1f} {us[4]:7.1f} {us[5]:7.1f} "
              f"{10.0 * x.numel() / us[5] / 1e6:7.2f}", flush=True)
        for k in range(6):
            tot[k] += us[k]
    print("TOTAL (us)               " + " ".join(f"{t:8.1f}" for t in tot[:4]) + " " * 16 + " ".join(f"{t:7.1f}" for t in tot[4:]))
    print("kernel labels:", lib().scat_last_kernel().decode())


if __name__ == "__main__":
    main()
