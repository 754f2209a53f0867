#!/usr/bin/env python
"""Eval-mode latency of EncoderTransformer (eval.py runs frame by frame and prints FPS, eval.py:632-649):
BatchNorm uses running statistics folded into the consumers' operand loads, no autograd, batch 1 / 8 / 32.
Eager launches vs the same forward captured once into a HIP graph (torch.cuda.CUDAGraph) and replayed — at batch 1 the
~250 launches of a forward are pure launch latency, which is what the graph removes.
(mask_rate 0: the reference draws the mask on the host with python `random` on every call, eval included —
hand_net.py:369-373 — which cannot be part of a captured graph.)"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import metrics, synth  # noqa: E402

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,8,32")
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    from scat_amd.models.hand_net import EncoderTransformer

    opt = SimpleNamespace(vit_heads=8, pl_reg=False, iteration=3, pos_embed=True, mask_rate=0.0, vit_depth=3)
    net = EncoderTransformer(opt, T(synth.mean_params(1)))
    net.load_state_dict(synth.to_torch(synth.encoder_transformer_state(1, 8)), strict=True)
    net.cuda().eval()
    for B in [int(b) for b in a.batches.split(",")]:
        x = T(synth.images(200, B)).cuda()
        with torch.no_grad():
            ref = net(x)[0].clone()
            eager = timeit(lambda: net(x), a.reps)
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    net(x)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                out = net(x)[0]
            graph = timeit(g.replay, a.reps)
            assert torch.equal(out, ref), "graph replay differs from the eager forward"
        gt = T(synth.normal_like(201, "gt", (B, 63), 0.03)).cuda()
        print(f"batch {B:3d}: eager {eager * 1e3:7.3f} ms ({B / eager:8.1f} img/s)   hipGraph replay {graph * 1e3:7.3f} ms "
              f"({B / graph:8.1f} img/s)   [MPJPE vs synthetic gt {metrics.mpjpe_mm(out, gt).item():.1f} mm, "
              f"PA-MPJPE {metrics.pa_mpjpe_mm(out.double(), gt.double()).item():.1f} mm]")


if __name__ == "__main__":
    main()
