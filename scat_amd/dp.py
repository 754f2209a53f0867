"""Flat gradient buckets + data-parallel all-reduce (one process per GPU, RCCL over xGMI).

The reference has no distributed code at all (DistributedDataParallel is imported and never
used, train.py:18), so this layer is new.  Design for MI355X:

* All trainable parameters and their gradients live in ONE flat fp32 buffer each, ordered by the
  order gradients become final in backward: head (regressor, transformer, conv1x1, mask_token)
  -> fc1 -> layer4 -> layer3 -> layer2 -> layer1 -> stem.  ``p.data`` / ``p.grad`` are views.
* The fused backbone backward writes weight gradients straight into the flat buffer and calls
  ``ready(bucket)`` when a stage is done; the bucket's all-reduce is issued immediately with
  ``async_op=True`` — torch.distributed runs it on the process group's own HIP stream (RCCL), so
  it overlaps the remaining backward kernels; ``finish()`` makes the compute stream wait before
  Adam reads the gradients.  6 large buckets (0.9-60 MB) rather than many small ones: xGMI is
  point-to-point, and RCCL's per-collective latency, not bandwidth, is what small buckets pay.
* With world_size == 1 the same object is just the flat storage for the fused Adam.

Device-agnostic (plain torch tensors + torch.distributed), so the N>1 logic is covered on CPU
with the gloo backend (tests/test_dp_gloo.py).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, List

import torch
import torch.distributed as dist

# SCAT_DP_CHECK=1: stream-ordering self-check.  zero_grad() poisons the flat gradient buffer with NaN; every bucket is
# tested for NaN ON THE STREAM ITS COLLECTIVE IS ORDERED AFTER, immediately in front of the collective; finish() raises
# if a bucket went out with a slice nobody had written yet (a producing stream that was not joined).
def _dp_check():
    return os.environ.get("SCAT_DP_CHECK", "0") != "0"


# SCAT_DP_FORCE_COLLECTIVES=1: take the world > 1 code path (replica broadcast, ReduceOp.AVG probe, asynchronous
# all-reduce per bucket on the ordering stream, work.wait() on the optimiser stream) even in a ONE-rank process group —
# how the RCCL branch is exercised on a one-GPU box (tests/test_gpu_dp.py::test_rccl_single_rank_collectives).
def _force_collectives():
    return os.environ.get("SCAT_DP_FORCE_COLLECTIVES", "0") != "0"


# The host driver of this pool only supports dmabuf IPC: RCCL's peer mappings over xGMI fail with "hipIpcGetMemHandle:
# invalid argument" without this, and it has to be in the environment BEFORE HIP initialises — i.e. before the model's
# constructor calls .cuda() (hand_net.py:321), which in the unattended path (auto_attach, from the first forward) is long
# past.  Importing the package is the earliest point this library controls.  Only multi-process runs need it (a launcher
# has set WORLD_SIZE, or the one-rank collective test forces the path): a single-GPU user's environment is left alone.
if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1 or _force_collectives():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

_PRODUCERS: List = []   # streams other than the caller's on which gradient kernels run (process-wide)


def register_producer(stream):
    """Declare a stream on which gradient kernels run besides the caller's (the weight-gradient side streams, the
    token path's stream).  Every bucket collective is ordered after everything queued on these streams so far —
    explicitly, by events — and not only after the stream that happens to be current when the bucket is declared
    ready (GradBuckets._ordered_stream)."""
    if stream is not None and all(s is not stream for s in _PRODUCERS):
        _PRODUCERS.append(stream)


BACKBONE_BUCKETS = ("fc1", "layer4", "layer3", "layer2", "layer1", "stem")


def _bucket_of(name: str) -> str:
    if name.startswith("main_encoder."):
        rest = name[len("main_encoder."):]
        for b in ("layer4", "layer3", "layer2", "layer1", "fc1"):
            if rest.startswith(b + "."):
                return b
        return "stem"
    return "head"


class GradBuckets:
    """Flat parameter / gradient storage for a model with a ``main_encoder`` backbone."""

    def __init__(self, model: torch.nn.Module, process_group=None):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        order = ("head",) + BACKBONE_BUCKETS
        groups: Dict[str, List] = OrderedDict((b, []) for b in order)
        for n, p in named:
            groups[_bucket_of(n)].append((n, p))
        ALIGN = 64   # floats (256 B): every parameter view stays 16-B aligned for the kernels' dwordx4 loads
        pad = lambda k: (k + ALIGN - 1) // ALIGN * ALIGN
        total = sum(pad(p.numel()) for _, p in named)
        dev = named[0][1].device
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)   # padding stays 0 under Adam
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.ranges: Dict[str, tuple] = {}
        self.slot: Dict[torch.nn.Parameter, tuple] = {}
        self.names: Dict[torch.nn.Parameter, str] = {}
        off = 0
        for b, items in groups.items():
            start = off
            for n, p in items:
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + k].view_as(p)   # parameters now alias the flat buffer
                self.slot[p] = (off, k)
                self.names[p] = n
                off += pad(k)
            self.ranges[b] = (start, off)
        self.head_params = [p for _, p in groups["head"]]
        self.bucket_params = {b: [p for _, p in items] for b, items in groups.items()}
        self.pg = process_group
        have_pg = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if have_pg else 1
        # collectives are issued when there is somebody to talk to — or when asked to rehearse them in a 1-rank group
        self.collective = self.world > 1 or (have_pg and _force_collectives())
        self._pending = []
        self._head_sent = False
        self._sent = set()
        self._comm_stream = None   # the stream every collective is ordered after (device runs only)
        self._checks = []
        self._pad_idx = None
        self._host_stage = {}
        self.check = _dp_check()
        self._avg_ok = False
        if self.collective:
            if dev.type == "cuda":
                # hardware queues (scat_amd.streams): normally planned by init_distributed() before the process group came
                # up; a group the caller created itself is met here, too late for the collective stream but not for ours
                from . import streams
                streams.bind(dev, ["wgrad", "tokens"])
            self._sync_replicas(model)
            self._avg_ok = self._probe_avg()
            if dev.type == "cuda":
                streams.bind(dev, ["comm"])
                streams.alias(dev, "opt", "comm")
        backbone = getattr(model, "main_encoder", None)
        if backbone is not None:
            backbone._grad_sink = self
        self.model = model
        self._auto = False
        self._armed = False
        self.tail_hook = None      # called once layer1's bucket has been issued (only the stem is left): see
                                   # trainer.FusedAdam.early

    # ---- replica start state (what DistributedDataParallel does at construction)
    def _sync_replicas(self, model):
        """Every rank starts from rank 0's parameters and buffers.  The reference's train.py never seeds
        (kaiming_normal_ / randn / default Linear init), so without this each rank would average gradients into a
        different model.  The flat buffer makes the parameters ONE broadcast; buffers (BatchNorm running statistics,
        num_batches_tracked, the positional table) go per dtype as one flat broadcast each."""
        dist.broadcast(self.flat_param, src=self._src_rank(), group=self.pg)
        by_dtype: Dict[torch.dtype, List[torch.Tensor]] = {}
        for b in model.buffers():
            by_dtype.setdefault(b.dtype, []).append(b)
        for dt, bufs in by_dtype.items():
            flat = torch.cat([b.detach().reshape(-1) for b in bufs])
            dist.broadcast(flat, src=self._src_rank(), group=self.pg)
            off = 0
            for b in bufs:
                k = b.numel()
                b.detach().copy_(flat[off:off + k].view_as(b))
                off += k

    def _src_rank(self):
        return dist.get_global_rank(self.pg, 0) if self.pg is not None else 0

    def _probe_avg(self):
        """ReduceOp.AVG exists in RCCL >= 2.10 only and a missing op surfaces asynchronously: ask once, on a
        one-element tensor, synchronously."""
        if dist.get_backend(self.pg) != "nccl":
            return False
        try:
            probe = torch.ones(1, device=self.flat_grad.device)
            dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=self.pg)
            torch.cuda.synchronize(probe.device)
            return bool(probe.item() == 1.0)
        except RuntimeError:
            return False

    # ---- stream ordering of the collectives
    def _ordered_stream(self, device):
        """-> the stream a collective of this moment has to be ordered after: a private stream that waits for the
        current stream and for every registered producer (an event record + wait each; the compute streams are not
        made to wait for each other, so their overlap is untouched)."""
        cur = torch.cuda.current_stream(device)
        if self._comm_stream is None:
            from . import streams
            self._comm_stream = streams.get(device, "comm")
        c = self._comm_stream
        c.wait_stream(cur)
        for s in _PRODUCERS:
            if s.device == c.device:
                c.wait_stream(s)
        return c

    # ---- unattended mode: the unmodified train.py never calls begin_backbone()/finish() itself
    def enable_auto(self):
        """Drive the buckets from autograd alone (SURVEY §8(e) 'train.py compatibility'): every parameter gets a
        post-accumulate hook; a bucket is all-reduced as soon as all of its parameters have their gradient; the
        compute stream is made to wait for the collectives by a callback at the end of the backward pass, so the
        ``optimizer.step()`` that follows sees averaged gradients.  Parameters the fused backbone backward fills
        directly (they never pass through AccumulateGrad) keep using ready()/adopt()."""
        if self._auto:
            return
        self._auto = True
        self._bucket_of_param = {}
        self._need = {}
        for b, (a, e) in self.ranges.items():
            members = [p for p, (off, _) in self.slot.items() if a <= off < e]
            self._need[b] = len(members)
            for p in members:
                self._bucket_of_param[p] = b
        self._have = {b: 0 for b in self.ranges}
        for p in self.slot:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def _arm(self):
        if self._auto and not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def _on_grad(self, p):
        self._arm()
        v = self.view_for(p)
        if p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
            p.grad = v
        b = self._bucket_of_param[p]
        self._have[b] += 1
        if self._have[b] == self._need[b]:
            self._allreduce(self.grad_slice(b), b)
            self._sent.add(b)
            if b == "head":
                self._head_sent = True

    def _end_of_backward(self):
        self._armed = False
        self._have = {b: 0 for b in self.ranges}
        self.finish()

    # ---- views
    def view_for(self, p):
        """the parameter's slice of the flat gradient buffer (None for a parameter that is not trained)"""
        if p not in self.slot:
            return None
        off, k = self.slot[p]
        return self.flat_grad[off:off + k].view_as(p)

    def grad_slice(self, bucket):
        a, b = self.ranges[bucket]
        return self.flat_grad[a:b]

    # ---- collectives
    def _allreduce(self, t, name="?"):
        if not self.collective:
            if self.check:
                self._checks.append((name, torch.isnan(t).any()))
            return
        backend = dist.get_backend(self.pg)
        if t.is_cuda:
            c = self._ordered_stream(t.device)
            with torch.cuda.stream(c):
                if self.check:
                    self._checks.append((name, torch.isnan(t).any()))
                if backend == "nccl":            # RCCL: runs on the group's own stream, ordered after c
                    op = dist.ReduceOp.AVG if self._avg_ok else dist.ReduceOp.SUM
                    work = dist.all_reduce(t, op=op, group=self.pg, async_op=True)
                    self._pending.append(("work", work, None if self._avg_ok else t))
                else:
                    # gloo with device tensors (N ranks rehearsed on one GPU): the bucket is snapshotted to pinned host
                    # memory ON THE ORDERED STREAM — it sees exactly the bytes RCCL would read — and reduced on the
                    # host when the step waits for its collectives.  A producer stream that was not joined shows up as
                    # a wrong (or, under SCAT_DP_CHECK, NaN) snapshot, as it would on hardware.
                    host = self._host_stage.get(name)
                    if host is None or host.numel() != t.numel():
                        host = self._host_stage[name] = torch.empty(t.numel(), dtype=t.dtype).pin_memory()
                    host.copy_(t, non_blocking=True)
                    ev = c.record_event()
                    self._pending.append(("staged", (ev, host), t))
            return
        # gloo on host tensors (CPU tests): SUM, scaled by 1/N in wait_pending()
        if self.check:
            self._checks.append((name, torch.isnan(t).any()))
        self._pending.append(("work", dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), t))

    def _gather(self, params):
        """p.grad -> the flat views (one multi-tensor copy); a parameter without a gradient owes its bucket zeros"""
        dst, src = [], []
        for p in params:
            v = self.view_for(p)
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)

    def begin_backbone(self):
        """Called at the top of the backbone backward: head gradients are final by then (every head node is
        nearer the loss than the backbone).  Gather them into the flat buffer and reduce them first."""
        self._arm()
        if self._head_sent:
            return          # the hooks (or an earlier backbone node of this backward) already reduced the head bucket
        backbone = getattr(self.model, "main_encoder", None)
        for p in getattr(backbone, "_flat_params", ()):
            v = self.view_for(p)
            if v is not None and p.grad is not None and p.grad.data_ptr() == v.data_ptr():
                raise RuntimeError(
                    "scat_amd.dp: the fused backbone backward writes its weight gradients straight into the flat "
                    "bucket (it does not accumulate): call zero_grad() before every backward — gradient accumulation "
                    "over several backward passes is not supported with flat buckets")
        self._gather(self.head_params)
        self._allreduce(self.grad_slice("head"), "head")
        self._head_sent = True
        self._sent.add("head")

    def ready(self, buckets):
        for b in buckets:
            self._allreduce(self.grad_slice(b), b)
            self._sent.add(b)
        if self.tail_hook is not None and "layer1" in buckets:
            self.tail_hook()

    def wait_pending(self):
        """make the CURRENT stream wait for every all-reduce issued so far (and apply the 1/N of the SUM fallback)"""
        for kind, work, scale_t in self._pending:
            if kind == "staged":
                ev, host = work
                ev.synchronize()                      # the snapshot is in host memory
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.pg)
                host.mul_(1.0 / self.world)
                scale_t.copy_(host, non_blocking=True)   # back into the bucket, on the waiting stream
                continue
            work.wait()
            if scale_t is not None:
                scale_t.mul_(1.0 / self.world)
        self._pending = []

    def adopt(self, params):
        """Point ``p.grad`` at the flat views the fused backward has just filled."""
        for p in params:
            v = self.view_for(p)
            if v is None:
                continue
            p.grad = v

    def finish(self):
        """Make the compute stream wait for every outstanding all-reduce (host does not block on GPU).  Buckets
        nobody declared ready — a backbone without the fused-backward protocol (HRNet, a ResNet assembled from
        scat_amd.nn modules), a frozen backbone — are gathered from ``p.grad`` and reduced here."""
        for b in self.ranges:
            if b in self._sent or not self.bucket_params[b]:
                continue
            self._gather(self.bucket_params[b])
            self._allreduce(self.grad_slice(b), b)
        self.wait_pending()
        self._head_sent = False
        self._sent = set()
        if self.check and self._checks:
            bad = [n for n, f in self._checks if bool(f.item())]
            self._checks = []
            if bad:
                raise RuntimeError(f"scat_amd.dp: buckets {bad} were handed to their all-reduce before every gradient "
                                   "in them had been written on the stream the collective is ordered after")

    def zero_grad(self):
        """``p.grad = None`` for every parameter (torch's set_to_none).  The flat gradient buffer itself is not
        cleared: every trained parameter's slice is rewritten by the next backward (fused backward: stores;
        everything else: gathered from p.grad, zeros where there is none) — required before EVERY backward."""
        for p in self.slot:
            p.grad = None
        if self.check:
            self.flat_grad.fill_(float("nan"))
            if self._pad_idx is None:      # the alignment padding between parameters is nobody's to write: keep it 0
                keep = torch.ones(self.flat_grad.numel(), dtype=torch.bool)
                for off, k in self.slot.values():
                    keep[off:off + k] = False
                self._pad_idx = keep.nonzero().flatten().to(self.flat_grad.device)
            self.flat_grad.index_fill_(0, self._pad_idx, 0.0)


def init_distributed():
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        # before model construction: the ctor calls .cuda() (hand_net.py:321).  (modulo: rehearsing N ranks on a
        # 1-GPU box with the gloo backend maps every rank onto the one device)
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    if (world > 1 or _force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("SCAT_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        if torch.cuda.is_available():
            from . import streams     # hardware queues: the train step's streams are bound before RCCL's (streams.py)
            streams.plan_for_collectives(torch.device("cuda", local))
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        if torch.cuda.is_available() and backend == "nccl":
            # the queue plan is verified, not trusted: did the communicator take the stream left for it, and do the
            # step's heavy streams still sit on queues of their own (re-picked / warned about otherwise)
            try:
                streams.verify_collective_plan(torch.device("cuda", local))
            except Exception as e:   # noqa: BLE001 - the check is advisory: it must never cost a run its process group
                import warnings

                warnings.warn(f"scat_amd.dp: the hardware-queue plan could not be verified ({e!r}); continuing",
                              RuntimeWarning)
    return rank, local, world


def auto_attach(model):
    """Called from the model's forward: under ``torch.distributed.run`` (WORLD_SIZE > 1) the first training-mode
    forward creates the flat buckets in unattended mode, so the reference's train.py (plain ``optim.Adam`` over
    ``net.parameters()``, ``loss.backward(); optimizer.step()``) trains data-parallel without modification.
    A model already owned by ``scat_amd.trainer.TrainStep`` (or attached before) is left alone."""
    import os

    if getattr(model, "_dp_buckets", None) is not None or not model.training:
        return
    if getattr(getattr(model, "main_encoder", None), "_grad_sink", None) is not None:
        return
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return
    if (torch.cuda.is_available() and torch.cuda.is_initialized() and not dist.is_initialized()
            and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0"):
        import warnings

        warnings.warn("scat_amd.dp: HIP was initialised with HSA_ENABLE_IPC_MODE_LEGACY != 0; on hosts whose driver "
                      "only supports dmabuf IPC RCCL then fails with 'hipIpcGetMemHandle: invalid argument' — export "
                      "HSA_ENABLE_IPC_MODE_LEGACY=0 in the launcher", RuntimeWarning)
    init_distributed()
    model._dp_buckets = GradBuckets(model)
    model._dp_buckets.enable_auto()
