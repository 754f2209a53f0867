#!/usr/bin/env python
"""How much of the train step does each kernel class really cost?  The step runs on five streams, so a class's
serialized kernel time says little about what removing it would buy.  This runs the bench step with ONE class of
library calls turned into no-ops (results are garbage — timing only) and reports the step time next to the full step:
the difference is the most any optimisation of that class can give.
    python tools/ablate_step.py [--steps 40] [--classes wgrad,bn_bwd,...]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from scat_amd import ops  # noqa: E402
from scat_amd._lib import lib  # noqa: E402

CLASSES = {
    "wgrad": ["scat_conv2d_wgrad", "scat_conv1x1_wgrad_bnb", "scat_conv7x7_s2_wgrad_split"],
    "wgrad1x1_only": None,      # handled below (needs the shape)
    "pointwise": ["scat_conv1x1_s1", "scat_conv1x1_s1_bnb"],
    "conv3x3": ["scat_conv3x3_s1"],
    "stride2": ["scat_conv2d_fwd_split", "scat_conv2d_dgrad_s2"],
    "stem": ["scat_conv7x7_s2_fwd_split", "scat_conv7x7_s2_wgrad_split", "scat_maxpool3x3s2_fwd", "scat_maxpool3x3s2_bwd"],
    "bn_fwd": ["scat_bn_apply", "scat_bn_train_stats", "scat_bn_train_stats_partials"],
    "bn_bwd": ["scat_bn_bwd", "scat_bn_bwd_pre"],
    "epi_fin": ["scat_bn_train_stats_partials", "scat_bn_train_stats_partials_shifted"],   # the finalize launches only
    "bn_apply": ["scat_bn_apply"],
    "bn_bwd_pre": ["scat_bn_bwd_pre"],
    "tokens": ["scat_gemm", "scat_attention_fwd", "scat_attention_bwd", "scat_layernorm_fwd", "scat_layernorm_bwd",
               "scat_gelu_fwd", "scat_gelu_bwd", "scat_colsum"],
    "adam_wprep": ["scat_adam", "scat_wprep_run"],
}


def run(steps, warmup):
    dev = torch.device("cuda", 0)
    net = bench.make_net("resnet50", 1, dev)
    step = bench.Step("resnet50", net, dev)
    u8, lab = bench.build_inputs(96, 2, dev)
    for _ in range(warmup):
        step(u8, lab)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        step(u8, lab)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    return statistics.median(ms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--classes", default=",".join(k for k in CLASSES if CLASSES[k]))
    a = ap.parse_args()
    L = lib()
    base = run(a.steps, a.warmup)
    print(f"{'full step':16s} {base:7.2f} ms")
    for name in a.classes.split(","):
        saved = {}
        for fn in CLASSES[name]:
            saved[fn] = getattr(L, fn)
            setattr(L, fn, lambda *args, **kw: 0)
        try:
            ms = run(a.steps, a.warmup)
        finally:
            for fn, f in saved.items():
                setattr(L, fn, f)
        print(f"without {name:12s} {ms:7.2f} ms   (the class costs the step {base - ms:.2f} ms)", flush=True)


if __name__ == "__main__":
    main()
