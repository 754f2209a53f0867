"""The reference's train step (train.py:136-209), restated for the benchmark and parity harness.

train.py itself needs torchvision / warmup_scheduler / oss2 / MANO assets that are not shipped, so
its inner iteration is restated here line by line in meaning (citations inline):
zero_grad -> net(inputs) -> orthographic projection + MSE/L1 loss (+ the constant pose-length
term) -> backward -> Adam.  Differences are only in *how* it runs on MI355X: the loss forward and
its gradient are one kernel, Adam is one launch over a flat parameter buffer, and in
data-parallel runs gradient buckets are all-reduced over RCCL while backward is still running.
"""
from __future__ import annotations

import os

import torch

from . import _switches as _sw

from . import ops
from .dp import GradBuckets


class _LossFn(torch.autograd.Function):
    """w3d*MSE(3-D) + w2d*L1(2-D after batch_orth_proj_idrot*112+112) — train.py:165-203."""

    @staticmethod
    def forward(ctx, out, labels, w3d, w2d):
        losses, dout = ops.loss_fwd_bwd(out.contiguous(), labels.contiguous(), w3d, w2d)
        ctx.save_for_backward(dout)
        ctx.mark_non_differentiable(losses)
        return losses[0].clone(), losses

    @staticmethod
    def backward(ctx, g, _):
        (dout,) = ctx.saved_tensors
        return dout * g, None, None, None


def scat_loss(outputs, labels, w3d=100000.0, w2d=10.0):
    """-> (loss scalar with grad, tensor [loss, l_3d, l_2d])."""
    return _LossFn.apply(outputs, labels, w3d, w2d)


def pose_length_term(pl_term):
    """train.py:178-183.  A constant w.r.t. the parameters (pl_term has no graph): reported, adds no gradient.
    Two small launches (csrc/misc.hip) instead of a dozen torch reductions."""
    if not pl_term.is_cuda:
        pl_len = pl_term.square().sum(dim=[2, 3]).mean(dim=[1]).sqrt()
        pl_mean = 0.01 * pl_len.mean()
        return (pl_len - pl_mean).square().mean()
    return ops.pose_length_term(pl_term)


EARLY_ADAM = _sw.ab("SCAT_EARLY_ADAM", True)


def _aux_stream(device):
    from . import streams
    return streams.get(device, "aux")


class FusedAdam:
    """torch.optim.Adam(params, lr) defaults (train.py:60) as one kernel over the flat buffers.

    With a fused backbone the update is issued in two parts: everything but the stem bucket from INSIDE the backward,
    as soon as layer1's gradients are final (``early``, hooked into GradBuckets.ready) — on its own stream, followed
    by the next step's weight re-layout — so that both run under the stem's backward (max-pool, BatchNorm, the 7x7
    weight gradient: 0.8 ms in which nothing else is left to do) instead of alone at the step boundary; ``step``
    then only updates the stem bucket."""

    def __init__(self, buckets: GradBuckets, lr=5e-4, betas=(0.9, 0.999), eps=1e-8):
        self.b = buckets
        self.lr, self.betas, self.eps = lr, betas, eps
        self.m = torch.zeros_like(buckets.flat_param)
        self.v = torch.zeros_like(buckets.flat_param)
        self.t = 0
        self._early_done = False
        self._opt_stream = None
        backbone = getattr(buckets.model, "main_encoder", None)
        self._wprep = getattr(backbone, "_wprep", None)
        if EARLY_ADAM and buckets.flat_param.is_cuda and "stem" in buckets.ranges and self._wprep is not None:
            buckets.tail_hook = self.early

    def zero_grad(self):
        self.b.zero_grad()

    def _adam(self, a, e):
        ops.adam(self.b.flat_param[a:e], self.b.flat_grad[a:e], self.m[a:e], self.v[a:e], self.lr, self.t,
                 self.betas[0], self.betas[1], self.eps)

    def early(self):
        """(called from the backbone backward) update every bucket but the stem's, then re-lay the weights"""
        if self._early_done:
            return
        main = torch.cuda.current_stream()
        if self._opt_stream is None:
            from . import streams      # (with collectives: the stream they are ordered after — it waits for them anyway)
            self._opt_stream = streams.get(self.b.flat_param.device, "opt")
        opt = self._opt_stream
        opt.wait_stream(main)
        with torch.cuda.stream(opt):
            self.b.wait_pending()          # data-parallel: the all-reduces of these buckets
            self.t += 1
            self._adam(0, self.b.ranges["stem"][0])      # the stem is the last range of the flat buffers
            self._wprep.run_early()
        self._early_done = True

    def step(self):
        self.b.finish()            # compute stream waits for the RCCL stream here
        if self._early_done:
            torch.cuda.current_stream().wait_stream(self._opt_stream)
            self._adam(*self.b.ranges["stem"])
            self._early_done = False
            return
        self.t += 1
        self._adam(0, self.b.flat_param.numel())


class TrainStep:
    """One object per process/GPU: model + flat buckets + fused Adam; ``__call__(inputs, labels)``
    runs exactly one train.py inner iteration and returns the loss tensors (no host sync)."""

    def __init__(self, net, lr=5e-4, w3d=100000.0, w2d=10.0, process_group=None):
        self.net = net
        self.buckets = GradBuckets(net, process_group)
        self.opt = FusedAdam(self.buckets, lr)
        self.w3d, self.w2d = w3d, w2d

    def __call__(self, inputs, labels):
        self.opt.zero_grad()                                   # train.py:154
        out = self.net(inputs)                                 # train.py:158-161
        pred = out[0]
        loss, parts = scat_loss(pred, labels, self.w3d, self.w2d)
        # train.py:200-201 adds 10 * l_pl to the loss before backward(); l_pl has no graph (pl_term is a detached
        # gradient), so the gradients are those of `loss` alone.  Its dozen tiny reductions are therefore issued
        # AFTER the backward has been queued, on an auxiliary stream, instead of sitting between forward and backward.
        loss.backward()                                        # train.py:206
        l_pl = None
        total = loss.detach()
        # which output is the pose-length term: EncoderTransformer returns (pred, feat_visual[, pl_term])
        # (hand_net.py:395-398), EncoderTransformerCoarse (pred, feat_visual, attn[, pl_term]) (hand_net.py:305-311)
        # — a [B,8,21,21] attention map in third place is not a pose-length term
        has_pl = bool(self.net.pl) if hasattr(self.net, "pl") else len(out) == 3
        pl_term = out[-1] if has_pl and len(out) >= 3 and out[-1].shape == out[1].shape else None
        if pl_term is not None:
            if pred.is_cuda:
                main = torch.cuda.current_stream()
                aux = _aux_stream(pred.device)
                aux.wait_stream(main)
                with torch.cuda.stream(aux):
                    l_pl = pose_length_term(pl_term)
                    total = torch.add(total, l_pl, alpha=10.0)
                pl_term.record_stream(aux)
                main.wait_stream(aux)          # (queued behind the whole backward: costs the critical path nothing)
                l_pl.record_stream(main)
                total.record_stream(main)
            else:
                l_pl = pose_length_term(pl_term)
                total = total + 10 * l_pl
        self.opt.step()                                        # train.py:209
        return total, parts, l_pl, pred.detach()
