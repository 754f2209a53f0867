#!/usr/bin/env python
"""Headline benchmark: images/sec of one full reg_transformer train step (train.py:136-209 of the
reference: forward, loss, backward, [gradient all-reduce], Adam) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: ResNet-50 backbone + dim-halving transformer (8 heads) +
3 regressor iterations, pl_reg, mask_rate 0.2, positional encoding, batch 96 per GPU, synthetic
256x256 RGB source images resampled to the network's only legal geometry 224x224
BEFORE the timed region; fp32 end to end.  One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# fp32 work per image of one train step (SURVEY §8d / BASELINE.md §2, torch flop counter on the reference)
GF_TRAIN_PER_IMG = 24.987
PEAK_F32_MFMA_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TF = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16, 32 clk)
PEAK_HBM_GBS = 8000.0


def build_inputs(batch, seed, device, src=256):
    """Synthetic uint8 source images -> [-1,1] -> bilinear 224x224 (dataset/load_STB.py:55 Resize(224))."""
    from scat_amd import synth

    from scat_amd import ops

    u8 = torch.from_numpy(synth.randint_u8(seed, "bench_images", (batch, 3, src, src))).to(device)
    x = ops.preprocess_u8(u8, (224, 224))      # normalise + bilinear resize in one HIP kernel
    lab = torch.from_numpy(synth.labels(seed + 1, batch)).to(device)
    return x.contiguous(), lab


def make_net(seed, device):
    from scat_amd import synth
    from scat_amd.models.hand_net import EncoderTransformer

    opt = SimpleNamespace(vit_heads=8, pl_reg=True, iteration=3, pos_embed=True, mask_rate=0.2, vit_depth=3)
    net = EncoderTransformer(opt, torch.from_numpy(synth.mean_params(seed)))
    net.load_state_dict(synth.to_torch(synth.encoder_transformer_state(seed, 8)), strict=True)
    return net.to(device).train()


def cpu_baseline(batch=32, steps=5):
    """The CPU oracle (a torch-CPU restatement of the reference path, pinned to the reference's outputs by
    tests/golden) on a bounded sample of the same workload, all host cores."""
    from oracle import scat_oracle as O
    from scat_amd import synth

    # the box's CPU share, not the host's core count: oversubscribed OpenMP threads spin
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("SCAT_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    sd = synth.to_torch(synth.encoder_transformer_state(1, 8))
    mp = torch.from_numpy(synth.mean_params(1))
    x = torch.from_numpy(synth.images(2, batch))
    lab = torch.from_numpy(synth.labels(3, batch))
    st = {}
    random.seed(3)
    O.train_step(sd, mp, x, lab, st, 1)      # warm-up
    t0 = time.perf_counter()
    for s in range(steps):
        O.train_step(sd, mp, x, lab, st, s + 2)
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{steps} train steps at batch {batch} (224x224, heads 8, iteration 3, pl_reg), "
                      f"{dt:.1f} s of CPU work, torch {torch.__version__} CPU"}


def instantiation_of(label):
    """kernel label (scat_last_kernel) -> the template instantiation it launches, as tools/traffic_json.py names it"""
    import re
    tf = "t" if label.endswith("_tf") or "_tf_" in label else "f"
    ds = "t" if "_bnb" in label else "f"      # dual-source operand (folded BatchNorm backward)
    m = re.search(r"_split_(\d+)x(\d+)x32", label)
    if m:       # pointwise / taps kernel: WM = rows / 32
        return f"conv1x1_split_kernel<{int(m.group(1)) // 32},{m.group(2)},{tf},{ds}>"
    m = re.match(r"conv3x3_split_(\d+)x(\d+)x16", label)
    if m:
        return f"conv3x3_split_kernel<{int(m.group(1)) // 32},{m.group(2)},{tf}>"
    m = re.match(r"wgrad(1x1|3x3)(_s2)?_split_(\d+)x(\d+)x16", label)
    if m:
        return (f"wgrad_split_kernel<{9 if m.group(1) == '3x3' else 1},{int(m.group(3)) // 64},{int(m.group(4)) // 64},"
                f"{tf},{'t' if m.group(2) else 'f'},{ds}>")
    return label


def dominant_kernel_roofline(ts, x, lab):
    """One extra instrumented step (outside the timed region): HIP events around every launch of the
    contraction engine, grouped by kernel instantiation; report the one with the most total time."""
    from scat_amd import ops
    from scat_amd.models import resnet as resnet_mod

    # per-kernel durations are only meaningful when kernels do not share the GPU: this one step runs the
    # weight gradients on the main stream instead of the side stream (the timed steps overlap them)
    from scat_amd.models import hand_net as hand_net_mod

    side, resnet_mod.SIDE_WGRAD = resnet_mod.SIDE_WGRAD, False
    overlap, hand_net_mod.OVERLAP_TOKENS = hand_net_mod.OVERLAP_TOKENS, False     # (and the token path in line)
    ops.PROFILE = []
    ts(x, lab)
    torch.cuda.synchronize()
    rec, ops.PROFILE = ops.PROFILE, None
    resnet_mod.SIDE_WGRAD = side
    hand_net_mod.OVERLAP_TOKENS = overlap
    agg = {}
    for name, flops, e0, e1 in rec:
        ms = e0.elapsed_time(e1)
        a = agg.setdefault(name, [0.0, 0.0, 0])
        a[0] += ms
        a[1] += flops
        a[2] += 1
    if not agg:
        return None, {}
    name, (ms, flops, n) = max(agg.items(), key=lambda kv: kv[1][0])
    tf = flops / (ms * 1e-3) / 1e12
    table = {k: {"launches": v[2], "ms": round(v[0], 3), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2)}
             for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}
    # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE
    # collected separately, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM); null if this kernel was not profiled
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
            traffic = json.load(fh).get(instantiation_of(name), {}).get("hbm_bytes_per_launch")
    except OSError:
        pass
    # kernels whose label says "split" form each fp32 product from six bf16 MFMA terms: their
    # ceiling for ALGORITHMIC fp32 FLOPs is the dense bf16 MFMA peak / 6; the others use the fp32 MFMA
    split = "_split" in name
    peak = round(PEAK_BF16_MFMA_TF / 6.0, 1) if split else PEAK_F32_MFMA_TF
    roof = {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(tf / peak, 4), "traffic": traffic, "kernel": name, "launches_per_step": n,
            "peak_basis": ("2500 TF dense bf16 MFMA / 6 bf16 terms per fp32 product" if split
                           else "157.3 TF fp32 MFMA"),
            "frac_of_fp32_mfma_peak": round(tf / PEAK_F32_MFMA_TF, 4),
            "avg_launch_ms": round(ms / n, 4), "algorithmic_gflop_per_launch": round(flops / n / 1e9, 3)}
    return roof, table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=96, help="per-GPU batch (reference: 96, script/ablation_pose.sh:11)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="also print the per-kernel table to stderr")
    a = ap.parse_args()

    from scat_amd.dp import init_distributed

    rank, local, world = init_distributed()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    import torch.distributed as dist
    from scat_amd._lib import lib
    from scat_amd.trainer import TrainStep

    lib().scat_check_device()
    dev = torch.device("cuda", local)
    net = make_net(1, dev)                      # identical weights on every rank
    ts = TrainStep(net, lr=5e-4)
    x, lab = build_inputs(a.batch, 100 + rank, dev)   # each rank its own shard (weak scaling)
    random.seed(3 + rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"model + data ready on {torch.cuda.get_device_name(dev)}; warm-up {a.warmup} steps")
    for _ in range(a.warmup):
        ts(x, lab)
    sync()
    note(f"timing {a.steps} steps")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, parts, lpl, pred = ts(x, lab)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    final_loss = float(loss.item())
    note(f"{a.steps} steps in {dt:.3f} s = {a.batch * world * a.steps / dt:.1f} img/s; roofline + CPU baseline legs")

    roof, table = (None, {})
    if not a.no_roofline:
        # every rank runs the instrumented step (it contains the gradient all-reduces); rank 0 reports
        roof, table = dominant_kernel_roofline(ts, x, lab)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        imgs = a.batch * world * a.steps
        step_tf = GF_TRAIN_PER_IMG * a.batch / (dt / a.steps) / 1e3
        out = {
            "metric": "images/sec (train step, 256x256 source -> 224x224, reg_transformer)",
            "value": round(imgs / dt, 2), "unit": "images/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ResNet-50 + dim-halving transformer (8 heads) + 3 regressor iterations, "
                                   "pl_reg, mask_rate 0.2, pos_embed, Adam; full train step",
                       "batch_per_gpu": a.batch, "global_batch": a.batch * world, "input": "3x224x224 fp32 "
                       "(from 256x256 synthetic uint8, resized before the timed region)",
                       "parallelism": f"dp{world}", "final_loss": final_loss,
                       "products": ("fp32 operands as 3 bf16 terms, 6 bf16 MFMA terms per product, fp32 accumulate "
                                    "(error <= 2^-25 per product; DESIGN.md 3.0)" if lib().scat_get_math_mode() == 1
                                    else "fp32 MFMA"),
                       "whole_step_tflops_per_gpu": round(step_tf, 2)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
        if a.kernel_table:
            print(json.dumps(table, indent=1), file=sys.stderr)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
