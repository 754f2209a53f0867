#!/usr/bin/env python
"""Headline benchmark: images/sec of one full reg_transformer train step (train.py:136-209 of the
reference: [input pipeline,] forward, loss, backward, [gradient all-reduce], Adam) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config resnet50|hrnet_w32|performer]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json configs[1]: ResNet-50 backbone + dim-halving transformer (8 heads) +
3 regressor iterations, pl_reg, mask_rate 0.2, positional encoding, batch 96 per GPU; fp32 end to end.
A step starts from the synthetic 256x256 uint8 source batch resident in HBM: normalise + bilinear resample to the
network's only legal geometry 224x224 (dataset/load_STB.py:48-67) runs INSIDE the timed step.
``value`` = images / wall time of the K steps between the two barriers; the median of the per-step HIP-event
times is reported beside it.  ``--config hrnet_w32`` / ``performer`` time BASELINE configs[3] / configs[4] on the same
contract.  One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import random
import statistics
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# fp32 work per image of one train step (SURVEY §8d / BASELINE.md §2, torch flop counter on the reference)
GF_TRAIN_PER_IMG = {"resnet50": 24.987, "hrnet_w32": 3 * (15.63 + 0.527), "performer": 3 * (8.179 + 0.017 + 3 * 0.240)}
PEAK_F32_MFMA_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TF = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16, 32 clk)
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured with a float4 copy)


def opt_ns(**kw):
    d = dict(vit_heads=8, pl_reg=True, iteration=3, pos_embed=True, mask_rate=0.2, vit_depth=3, hrnet_width=32)
    d.update(kw)
    return SimpleNamespace(**d)


def make_net(config, seed, device):
    """Random-init weights of the named architecture from the in-repo counter hash (identical on every rank)."""
    from scat_amd import synth
    from scat_amd.models import hand_net as H

    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    if config == "resnet50":
        net = H.EncoderTransformer(opt_ns(), T(synth.mean_params(seed)))
        net.load_state_dict(synth.to_torch(synth.encoder_transformer_state(seed, 8)), strict=True)
    elif config == "hrnet_w32":
        # BASELINE configs[3]: HRNet(c=32, nof_joints=128) -> view(512,28,28) -> conv3x3/2 -> [B,128,196] ->
        # vit.Transformer(196,3,8,64,392) -> 3 x Linear(257->61)   (hand_net.py:150-213 with models.vit swapped in)
        net = H.EncoderTransformerHRNet(opt_ns(pl_reg=False), T(synth.mean_params(seed, 61)))
        net.load_state_dict(synth.to_torch(synth.hrnet_wrapper_state(seed + 1, net.state_dict())), strict=True)
    else:
        # BASELINE configs[4]: ResNet-50 tokens -> 3 x performer_attn_block(49, 16) -> iteration 5, mask 0.2
        torch.manual_seed(seed)
        net = H.EncoderPerformer(opt_ns(vit_heads=16, iteration=5, pl_reg=False), T(synth.mean_params(seed)))
        sd = synth.to_torch(synth.encoder_transformer_state(seed, 8))
        net.main_encoder.load_state_dict({k[len("main_encoder."):]: v for k, v in sd.items()
                                          if k.startswith("main_encoder.")}, strict=True)
    return net.to(device).train()


class Step:
    """One train.py inner iteration starting from the uint8 source batch in HBM."""

    def __init__(self, config, net, dev):
        from scat_amd import ops
        from scat_amd.trainer import TrainStep

        self.ops, self.config = ops, config
        if config == "hrnet_w32":
            # the reference has no trainer for this wrapper (train.py builds EncoderTransformer only, train.py:49-57)
            # and its 61 outputs are MANO parameters, not joints: the step is forward + a fixed linear functional of
            # the outputs + backward + Adam.  The update is the library's fused Adam over the flat buckets (torch.optim.Adam's
            # defaults, one launch; the gradients autograd left in p.grad are gathered into the flat buffer by one
            # multi-tensor copy — and all-reduced there when there is more than one rank): torch.optim.Adam's foreach path
            # costs this 900-parameter model ~250 small launches and 3.5 ms of GPU time per step (SCAT_BENCH_TORCH_ADAM=1)
            self.net = net
            if os.environ.get("SCAT_BENCH_TORCH_ADAM", "0") != "0":
                self.opt = torch.optim.Adam(net.parameters(), lr=1e-5)
            else:
                from scat_amd.dp import GradBuckets
                from scat_amd.trainer import FusedAdam
                self.opt = FusedAdam(GradBuckets(net), lr=1e-5)
            self.cot = None
            self.ts = None
        else:
            self.ts = TrainStep(net, lr=5e-4)

    def __call__(self, u8, lab):
        x = self.ops.preprocess_u8(u8, (224, 224))      # normalise + bilinear resize, one HIP kernel, in the step
        if self.ts is not None:
            return self.ts(x, lab)
        if isinstance(self.opt, torch.optim.Optimizer):
            self.opt.zero_grad(set_to_none=True)
        else:
            self.opt.zero_grad()
        pred = self.net(x)
        pred = pred[0] if isinstance(pred, tuple) else pred
        if self.cot is None:
            self.cot = torch.full_like(pred, 1e-3)
        loss = (pred * self.cot).sum()
        loss.backward()
        self.opt.step()
        return loss.detach(), None, None, pred.detach()


def build_inputs(batch, seed, device, src=256):
    from scat_amd import synth

    u8 = torch.from_numpy(synth.randint_u8(seed, "bench_images", (batch, 3, src, src))).to(device)
    lab = torch.from_numpy(synth.labels(seed + 1, batch)).to(device)
    return u8, lab


def h2d_pipeline_rate(step, u8, lab, steps):
    """PCIe-inclusive rate (never ``value``): the uint8 source batch starts in PINNED HOST memory every step and is
    copied by a side stream into one of two device buffers while the previous step computes (what
    train.py:152-153 does synchronously with a 57.8 MB fp32 batch is an 18.9 MB uint8 copy here, because the
    normalise + resize runs on the device)."""
    dev = u8.device
    host = [u8.cpu().pin_memory(), u8.cpu().pin_memory()]
    devb = [torch.empty_like(u8), torch.empty_like(u8)]
    copy = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream()
    ready = [None, None]
    freed = [None, None]

    def issue(i):
        b = i & 1
        with torch.cuda.stream(copy):
            if freed[b] is not None:
                copy.wait_event(freed[b])          # the step that read this buffer two steps ago has finished
            devb[b].copy_(host[b], non_blocking=True)
            ready[b] = copy.record_event()

    issue(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if i + 1 < steps:
            issue(i + 1)
        b = i & 1
        main.wait_event(ready[b])
        step(devb[b], lab)
        freed[b] = main.record_event()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"images_per_s": round(u8.shape[0] * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "h2d_mb_per_step": round(u8.numel() / 1e6, 1),
            "note": "uint8 source batch from pinned host memory, double-buffered on a copy stream under the previous step"}


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline():
    """The CPU oracle (a torch-CPU restatement of the reference path, pinned to the reference's outputs by
    tests/golden) on a bounded sample of the same workload: batch 96 and batch 8, on the cores the
    box gives this process (16 per GPU; SCAT_CPU_THREADS overrides)."""
    from oracle import scat_oracle as O
    from scat_amd import synth

    # the box's CPU SHARE, not the host's core count: a 1-GPU box gives this process 16 cores of a much larger host
    # (sched_getaffinity lists them all), and OpenMP threads beyond the share spin against each other for minutes
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("SCAT_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    sd = synth.to_torch(synth.encoder_transformer_state(1, 8))
    mp = torch.from_numpy(synth.mean_params(1))
    res = {}
    for batch, warm, steps in ((8, 2, 5), (96, 1, 2)):
        x = torch.from_numpy(synth.images(2, batch))
        lab = torch.from_numpy(synth.labels(3, batch))
        st = {}
        random.seed(3)
        for s in range(warm):
            O.train_step(sd, mp, x, lab, st, s + 1)
        ts = []
        for s in range(steps):
            t0 = time.perf_counter()
            O.train_step(sd, mp, x, lab, st, warm + s + 1)
            ts.append(time.perf_counter() - t0)
        res[batch] = (batch / statistics.median(ts), sum(ts), steps)
    return {"value": round(res[96][0], 2), "unit": "images/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "batch8_images_per_s": round(res[8][0], 2),
            "sample": f"median of {res[96][2]} train steps at batch 96 after 1 warm-up ({res[96][1]:.1f} s) and of "
                      f"{res[8][2]} steps at batch 8 after 2 ({res[8][1]:.1f} s): 224x224, heads 8, iteration 3, "
                      f"pl_reg, Adam; torch {torch.__version__} CPU, {cores} threads"}


def instantiation_of(label, traffic=None):
    """kernel label (scat_last_kernel) -> the template instantiation it launches, as tools/traffic_json.py names it"""
    import re
    tf = "t" if label.endswith("_tf") or "_tf_" in label else "f"
    ds = "t" if "_bnb" in label else "f"      # dual-source operand (folded BatchNorm backward)
    epi = "t" if label.endswith("_epibn") else "f"      # BatchNorm-backward sums in the epilogue (scat_epilogue_bnb_arm)
    m = re.search(r"_split_pc(\d+)x128x32", label)
    if m:
        return f"conv1x1_pc_kernel<{int(m.group(1)) // 128},{tf},0>"
    m = re.search(r"_split_(\d+)x(\d+)x32", label)
    if m:       # pointwise / taps kernel: WM = rows / 32; the stem is the same kernel with the STEM staging
        stem = "t" if label.startswith("conv7x7_s2_split") else "f"
        base = f"conv1x1_split_kernel<{int(m.group(1)) // 32},{m.group(2)},{tf},{ds},{stem}"
        # one tap and C % 32 == 0 (every pointwise layer of the networks) launch the mask-free staging variant
        cands = ([f"{base},t,{epi}>"] if stem == "f" else []) + [f"{base},f,{epi}>"]
        for c in cands:
            if traffic is not None and c in traffic:
                return c
        return cands[0]
    m = re.match(r"conv3x3_split_(\d+)x(\d+)x16", label)
    if m:
        return f"conv3x3_split_kernel<{int(m.group(1)) // 32},{m.group(2)},{tf}>"
    m = re.match(r"wgrad1x1_pw_(\d+)x(\d+)x32", label)
    if m:       # second-generation pointwise weight gradient: WA x WB consumer wavefronts of 64 x 64
        rag = "t" if "_rag" in label else "f"
        return f"wgrad_pw_kernel<{int(m.group(1)) // 64},{int(m.group(2)) // 64},{tf},{ds},{rag},0,4>"
    m = re.match(r"wgrad3x3_rows_64x576x16.*_r(\d)", label)
    if m:
        return f"wgrad3x3_rows64_kernel<{m.group(1)}>"
    if label.startswith("wgrad3x3_rows"):
        return "wgrad3x3_rows_kernel"
    if label.startswith("wgrad7x7_s2_split"):
        return "stem_wgrad_split_kernel"
    m = re.match(r"wgrad(1x1|3x3)(_s2)?_split_pc128x128x16", label)
    if m:
        return f"wgrad_pc_kernel<{9 if m.group(1) == '3x3' else 1},{tf},{'t' if m.group(2) else 'f'},{ds}>"
    m = re.match(r"wgrad(1x1|3x3)(_s2)?_split_(\d+)x(\d+)x16", label)
    if m:
        return (f"wgrad_split_kernel<{9 if m.group(1) == '3x3' else 1},{int(m.group(3)) // 64},{int(m.group(4)) // 64},"
                f"{tf},{'t' if m.group(2) else 'f'},{ds}>")
    return label


def latest_traffic():
    """HBM bytes per launch from the newest committed rocprofv3 --pmc passes over this bench (tools/traffic_json.py)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return {}, None
    try:
        return json.load(open(files[-1])), os.path.basename(files[-1])
    except (OSError, ValueError):
        return {}, None


def kernel_rooflines(step, u8, lab):
    """One extra instrumented step (outside the timed region): HIP events on the launch stream around every launch of
    the contraction kernels (algorithmic FLOPs) and of the BatchNorm passes (algorithmic bytes), grouped by kernel;
    report the contraction kernel and the BatchNorm pass with the most total time."""
    from scat_amd import ops
    from scat_amd.models import hand_net as hand_net_mod
    from scat_amd.models import resnet as resnet_mod

    # per-kernel durations are only meaningful when kernels do not share the GPU: this one step runs the
    # weight gradients on the main stream instead of the side stream (the timed steps overlap them)
    side, resnet_mod.SIDE_WGRAD = resnet_mod.SIDE_WGRAD, False
    overlap, hand_net_mod.OVERLAP_TOKENS = hand_net_mod.OVERLAP_TOKENS, False     # (and the token path in line)
    ops.PROFILE, ops.PROFILE_HBM = [], []
    step(u8, lab)
    torch.cuda.synchronize()
    rec, ops.PROFILE = ops.PROFILE, None
    rec_hbm, ops.PROFILE_HBM = ops.PROFILE_HBM, None
    resnet_mod.SIDE_WGRAD = side
    hand_net_mod.OVERLAP_TOKENS = overlap

    def aggregate(records):
        agg = {}
        for name, work, e0, e1 in records:
            a = agg.setdefault(name, [0.0, 0.0, 0])
            a[0] += e0.elapsed_time(e1)
            a[1] += work
            a[2] += 1
        return agg

    agg = aggregate(rec)
    roof, table = None, {}
    if agg:
        name, (ms, flops, n) = max(agg.items(), key=lambda kv: kv[1][0])
        tf = flops / (ms * 1e-3) / 1e12
        table = {k: {"launches": v[2], "ms": round(v[0], 3), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2)}
                 for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}
        # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE
        # collected separately, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM); null if this kernel was not profiled
        tj, tj_file = latest_traffic()
        traffic = tj.get(instantiation_of(name, tj), {}).get("hbm_bytes_per_launch")
        # kernels whose label says "split" form each fp32 product from six bf16 MFMA terms: their
        # ceiling for ALGORITHMIC fp32 FLOPs is the dense bf16 MFMA peak / 6; the others use the fp32 MFMA
        split = "_split" in name
        peak = round(PEAK_BF16_MFMA_TF / 6.0, 1) if split else PEAK_F32_MFMA_TF
        roof = {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(tf / peak, 4), "traffic": traffic, "traffic_source": tj_file, "kernel": name,
                "launches_per_step": n,
                "peak_basis": ("2500 TF dense bf16 MFMA / 6 bf16 terms per fp32 product" if split
                               else "157.3 TF fp32 MFMA"),
                "avg_launch_ms": round(ms / n, 4), "algorithmic_gflop_per_launch": round(flops / n / 1e9, 3)}
    roof_hbm = None
    agg = aggregate(rec_hbm)
    if agg:
        name, (ms, nbytes, n) = max(agg.items(), key=lambda kv: kv[1][0])
        gbs = nbytes / (ms * 1e-3) / 1e9
        roof_hbm = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "kernel": name, "launches_per_step": n,
                    "avg_launch_ms": round(ms / n, 4), "algorithmic_mb_per_launch": round(nbytes / n / 1e6, 2),
                    "all_passes": {k: {"launches": v[2], "ms": round(v[0], 3),
                                       "gb_per_s": round(v[1] / (v[0] * 1e-3) / 1e9, 1)}
                                   for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}}
    return roof, roof_hbm, table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=96, help="per-GPU batch (reference: 96, script/ablation_pose.sh:11)")
    ap.add_argument("--config", default="resnet50", choices=sorted(GF_TRAIN_PER_IMG),
                    help="resnet50 = BASELINE configs[1]/[2] (the headline); hrnet_w32 = configs[3]; performer = configs[4]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="also print the per-kernel table to stderr")
    ap.add_argument("--h2d", action="store_true", help="also time the steps with the source batch coming from pinned "
                    "host memory (double-buffered H2D; reported as config.pcie_inclusive, never as value)")
    a = ap.parse_args()

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to stdout when its first communicator
    # comes up (seen with a one-rank nccl group, profiles/r03_rccl_single_rank.txt): keep the real stdout for the line
    # and point file descriptor 1 at stderr for everything else, native libraries included.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # distributed first: RANK / LOCAL_RANK / WORLD_SIZE from torch.distributed.run, the device is chosen and the
    # process group created before anything touches the GPU
    from scat_amd.dp import init_distributed

    rank, local, world = init_distributed()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    import torch.distributed as dist
    from scat_amd._lib import lib

    lib().scat_check_device()
    dev = torch.device("cuda", local)
    net = make_net(a.config, 1, dev)            # identical weights on every rank
    step = Step(a.config, net, dev)
    u8, lab = build_inputs(a.batch, 100 + rank, dev)   # each rank its own shard (weak scaling)
    random.seed(3 + rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"{a.config}: model + data ready on {torch.cuda.get_device_name(dev)}; warm-up {a.warmup} steps")
    for _ in range(a.warmup):
        step(u8, lab)
    sync()
    note(f"timing {a.steps} steps")
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        marks[i].record()
        loss, parts, lpl, pred = step(u8, lab)
    marks[a.steps].record()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    med = statistics.median(per_step)
    final_loss = float(loss.item())
    note(f"{a.steps} steps in {dt:.3f} s = {a.batch * world * a.steps / dt:.1f} img/s (median step {med:.3f} ms); "
         "roofline + CPU baseline legs")

    pcie = h2d_pipeline_rate(step, u8, lab, a.steps) if a.h2d else None
    roof, roof_hbm, table = None, None, {}
    if not a.no_roofline:
        # every rank runs the instrumented step (it contains the gradient all-reduces); rank 0 reports
        roof, roof_hbm, table = kernel_rooflines(step, u8, lab)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.config == "resnet50":
        cpu = cpu_baseline()

    # which side streams of the step share a hardware queue on this rank (scat_amd/streams.py; ~20 ms of spin kernels,
    # after everything that is timed): evidence that the queue plan held where the step actually ran
    from scat_amd import streams
    queue_mates = ["+".join((x, y)) + (" (one stream)" if why == "same stream" else "") for x, y, why in streams.sharing(dev)]
    # data-parallel evidence, so that a scaling record can be audited for "did the collectives see N ranks, on which
    # queues": per rank the queue-mates and what the queue-plan verification found; the backend, its version, whether
    # ReduceOp.AVG is native, and the bytes of every gradient bucket
    dp_info = None
    buckets = step.ts.buckets if step.ts is not None else getattr(step.opt, "buckets", None)
    if world > 1 or dist.is_initialized():
        mine = {"rank": rank, "hw_queue_mates": queue_mates, "queue_plan": streams.plan_report(dev)}
        per_rank = [None] * world
        if world > 1:
            dist.all_gather_object(per_rank, mine)
        else:
            per_rank = [mine]
        backend = dist.get_backend()
        ver = None
        if backend == "nccl":
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:   # noqa: BLE001 - a missing version query must not cost the line
                ver = None
        dp_info = {"world": world, "backend": backend, "rccl_version": ver,
                   "reduce_op_avg_native": bool(getattr(buckets, "_avg_ok", False)),
                   "collectives_per_step": len(buckets.ranges) if buckets is not None else None,
                   "bucket_bytes": ({b: 4 * (e - a0) for b, (a0, e) in buckets.ranges.items()} if buckets is not None else None),
                   "per_rank": per_rank}

    if rank == 0:
        imgs = a.batch * world * a.steps
        step_tf = GF_TRAIN_PER_IMG[a.config] * a.batch / (dt / a.steps) / 1e3
        workloads = {
            "resnet50": "ResNet-50 + dim-halving transformer (8 heads) + 3 regressor iterations, pl_reg, mask_rate 0.2, "
                        "pos_embed, Adam; full train step (BASELINE configs[1])",
            "hrnet_w32": "HRNet-W32 + vit.Transformer(196,3,8,64,392) + 3 x Linear(257->61), Adam; forward + linear "
                         "functional + backward + update (BASELINE configs[3])",
            "performer": "ResNet-50 tokens + 3 x performer_attn_block(49, heads 16) + 5 regressor iterations, mask_rate "
                         "0.2, Adam; full train step (BASELINE configs[4])"}
        out = {
            "metric": "images/sec (train step, 256x256 source -> 224x224, reg_transformer)",
            "value": round(imgs / dt, 2), "unit": "images/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workloads[a.config],
                       "batch_per_gpu": a.batch, "global_batch": a.batch * world,
                       "input": "uint8 3x256x256 synthetic source batch resident in HBM; normalise + bilinear resize to "
                                "3x224x224 fp32 inside the timed step",
                       "parallelism": f"dp{world}", "final_loss": final_loss,
                       "median_ms_per_step": round(med, 3),
                       "median_images_per_s": round(a.batch * world / (med * 1e-3), 2),
                       "products": ("fp32 operands as 3 bf16 terms, 6 bf16 MFMA terms per product, fp32 accumulate "
                                    "(error <= 2^-25 per product; DESIGN.md 3.0)" if lib().scat_get_math_mode() == 1
                                    else "fp32 MFMA"),
                       "whole_step_tflops_per_gpu": round(step_tf, 2), "pcie_inclusive": pcie,
                       "hw_queue_mates": queue_mates, "data_parallel": dp_info},
            "roofline": roof, "roofline_hbm": roof_hbm, "cpu_baseline": cpu,
        }
        print(json.dumps(out), file=json_out, flush=True)
        if a.kernel_table:
            print(json.dumps(table, indent=1), file=sys.stderr)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
