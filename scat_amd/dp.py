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

from collections import OrderedDict
from typing import Dict, List

import torch
import torch.distributed as dist

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
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self._pending = []
        self._head_sent = False
        self._avg_ok = True
        backbone = getattr(model, "main_encoder", None)
        if backbone is not None:
            backbone._grad_sink = self
        self.model = model
        self._auto = False
        self._armed = False
        self.tail_hook = None      # called once layer1's bucket has been issued (only the stem is left): see
                                   # trainer.FusedAdam.early

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
            self._allreduce(self.grad_slice(b))
            if b == "head":
                self._head_sent = True

    def _end_of_backward(self):
        self._armed = False
        self._have = {b: 0 for b in self.ranges}
        self.finish()

    # ---- views
    def view_for(self, p):
        off, k = self.slot[p]
        return self.flat_grad[off:off + k].view_as(p)

    def grad_slice(self, bucket):
        a, b = self.ranges[bucket]
        return self.flat_grad[a:b]

    # ---- collectives
    def _allreduce(self, t):
        if self.world <= 1:
            return
        backend = dist.get_backend(self.pg)
        if backend == "nccl" and self._avg_ok:   # RCCL on ROCm
            try:
                self._pending.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.pg, async_op=True), None))
                return
            except RuntimeError:
                self._avg_ok = False             # older RCCL without AVG: sum, then scale
        if t.is_cuda and backend != "nccl":
            # gloo with device tensors (rehearsing N ranks on one GPU): its asynchronous device path does not order
            # itself against HIP streams the way RCCL does (measured: buckets reduced before their last kernel had
            # written them) — reduce synchronously; this is a test vehicle, not a performance path
            torch.cuda.current_stream(t.device).synchronize()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
            t.mul_(1.0 / self.world)
            return
        # gloo on host tensors (CPU tests) / RCCL without AVG: SUM, scaled by 1/N in finish()
        self._pending.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), t))

    def begin_backbone(self):
        """Called at the top of the backbone backward: head gradients are final by then (every head node is
        nearer the loss than the backbone).  Gather them into the flat buffer and reduce them first."""
        self._arm()
        if self._auto and self._head_sent:
            return          # the hooks already reduced the head bucket
        dst, src = [], []
        for p in self.head_params:
            v = self.view_for(p)
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)   # one multi-tensor launch instead of one copy per parameter
        self._allreduce(self.grad_slice("head"))
        self._head_sent = True

    def ready(self, buckets):
        for b in buckets:
            self._allreduce(self.grad_slice(b))
        if self.tail_hook is not None and "layer1" in buckets:
            self.tail_hook()

    def wait_pending(self):
        """make the CURRENT stream wait for every all-reduce issued so far (and apply the 1/N of the SUM fallback)"""
        for work, scale_t in self._pending:
            work.wait()
            if scale_t is not None:
                scale_t.mul_(1.0 / self.world)
        self._pending = []

    def adopt(self, params):
        """Point ``p.grad`` at the flat views the fused backward has just filled."""
        for p in params:
            v = self.view_for(p)
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.add_(p.grad)   # honour gradient accumulation semantics
            p.grad = v

    def backbone_done(self):
        pass

    def finish(self):
        """Make the compute stream wait for every outstanding all-reduce (host does not block on GPU)."""
        if not self._head_sent:      # no backbone backward ran (e.g. frozen backbone): still gather the head
            self.begin_backbone()
        self.wait_pending()
        self._head_sent = False

    def zero_grad(self):
        for p in self.slot:
            p.grad = None


def init_distributed():
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        # before model construction: the ctor calls .cuda() (hand_net.py:321).  (modulo: rehearsing N ranks on a
        # 1-GPU box with the gloo backend maps every rank onto the one device)
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("SCAT_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
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
    init_distributed()
    model._dp_buckets = GradBuckets(model)
    model._dp_buckets.enable_auto()
