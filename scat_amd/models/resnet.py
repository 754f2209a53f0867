"""ResNet backbone with the reference's interface (models/resnet.py:103-222 of tomguluson92/SCAT):
same class/attribute names, constructor arguments, ``state_dict`` keys and the 5-tuple
``(feat[B,1024], x1, x2, x3, x4)`` return — executed as ONE autograd node whose forward and
backward are explicit sequences of libscat_hip kernels.

MI355X design (not a translation of the reference's module-by-module call graph):
* BatchNorm batch statistics are one HBM pass; the normalise+ReLU of bn1/bn2 is never
  materialised — it is folded into the operand load of the consuming conv (forward) and of the
  weight-gradient contraction (backward), and the ReLU mask is recomputed from the raw conv
  output in the BN backward.  Only block outputs (needed as residuals and returned as x1..x4)
  are written.
* The stem's BN+ReLU is folded into the max-pool load.
* Weight gradients are produced by a deterministic split-K contraction (fixed-order slab
  reduction, no float atomics), optionally straight into a flat gradient bucket that the
  data-parallel layer all-reduces on a side stream while earlier layers still run.
"""
from __future__ import annotations

import os

import torch

from .. import _switches as _sw
import torch.nn as nn

from .. import nn as snn
from .. import ops

__all__ = ["ResNet", "Bottleneck", "resnet50", "resnet101", "resnet152"]


class Bottleneck(nn.Module):
    """Parameter layout of models/resnet.py:62-76 (stride on the 3x3)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = snn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = snn.BatchNorm2d(planes)
        self.conv2 = snn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = snn.BatchNorm2d(planes)
        self.conv3 = snn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = snn.BatchNorm2d(planes * 4)
        self.relu = snn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        """models/resnet.py:78-98.  Inside ResNet the fused backbone node runs the blocks itself; this is the same
        block executor for a block used on its own (``net.layer1[0](x)``, ``net.layer1(x)``)."""
        return _BlockFn.apply(x, self, *self.parameters())


def _bn_buffers(bn):
    return bn.running_mean, bn.running_var


_NBT = []   # num_batches_tracked buffers touched by the running forward: bumped by ONE multi-tensor add at its end
_DEFER = [False]   # set by a network whose forward bumps them itself at its end (HRNet): stand-alone blocks leave _NBT alone


class _BNState:
    """Per-BN tensors produced in forward and consumed in backward."""
    __slots__ = ("mean", "invstd", "scale", "shift", "xs")

    def __init__(self, x, bn, training):
        if training:
            self.mean, self.invstd, self.scale, self.shift = ops.bn_train_stats(
                x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
            bn._stat_ref = self.mean       # next step's reference for the epilogue sums (ops.conv2d_fwd stats_shift)
            _NBT.append(bn.num_batches_tracked)
            bn._fold_cache = None          # running statistics move (by a raw kernel: no version bump)
        else:
            # inference: scale/shift from the running statistics, cached until a tensor they come from changes
            # (53 tiny launches per forward otherwise — a tenth of the batch-1 latency)
            ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
                   bn.weight.data_ptr(), bn.running_mean.data_ptr())
            c = getattr(bn, "_fold_cache", None)
            if c is None or c[0] != ver:
                c = (ver, *ops.bn_eval_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps))
                bn._fold_cache = c
            self.scale, self.shift = c[1], c[2]
            self.mean = self.invstd = None


BNB1 = _sw.ab("SCAT_BNB1", False)   # same for bn1 -> conv1: measured slower (its passes hide under the side stream), off
BNB1_MIN_H = _sw.ab_int("SCAT_BNB1_MIN_H", 0)
BNB_MIN_H = _sw.ab_int("SCAT_BNB_MIN_H", 28)   # fold bn3 only where it pays: 56x56 and 28x28 planes (tools/bnb_bench.py: at 14x14 the dual-source kernels cost more than the pass they save)
BNB = _sw.ab("SCAT_BNB", True)   # fold bn3's backward apply into conv3's gradient kernels
SUBSAMPLE = _sw.ab("SCAT_SUBSAMPLE", True)   # pack the input of the 1x1/stride-2 shortcuts (stride-1 kernels)
SIDE_SHORTCUT = _sw.ab("SCAT_SIDE_SHORTCUT", True)   # forward: the shortcut convolution beside conv1..conv3 (own stream)
SIDE_DS_BN = _sw.ab("SCAT_SIDE_DS_BN", True)   # the shortcut's BatchNorm backward beside the main data-gradient chain
STEM_FUSED_BWD = _sw.ab("SCAT_STEM_FUSED_BWD", True)   # max-pool backward inside bn1's backward (stem)
SIDE_WGRAD = _sw.ab("SCAT_SIDE_WGRAD", True)   # bench.py clears it for its serialized, per-kernel-timed step
EPI_BNB = _sw.ab("SCAT_EPI_BNB", True)   # bn3's backward reduction in the epilogue of the kernel that completes its gradient


def _side_stream(device, who="backbone"):
    """The side stream of the weight-gradient contractions (SCAT_SIDE_WGRAD=0 disables): the backbone's, or the token
    path's (who="tokens").  Created through scat_amd.streams so that it gets a hardware queue of its own."""
    if not SIDE_WGRAD:
        return None
    from .. import streams
    return streams.get(device, "wgrad" if who == "backbone" else "tokens_wgrad")


STAT_REF = _sw.ab("SCAT_STAT_REF", True)   # epilogue BatchNorm sums about the previous step's batch mean


def _ref(bn):
    """the reference the convolution epilogue takes its BatchNorm sums about: the previous training step's batch mean of
    this BatchNorm (None on the first step: plain sums)"""
    return getattr(bn, "_stat_ref", None) if STAT_REF else None


def _block_forward(blk, xin, training, wp=None):
    """Bottleneck.forward (models/resnet.py:78-98) as a kernel sequence -> tape record
    (blk, xin, c1, s1, c2, s2, c3, s3, cd, sd, out, omask).  bn1/bn2's normalise+ReLU live in the operand load of
    the next convolution; bn3 (+ the shortcut's BatchNorm) + residual + ReLU is the one pass that writes the output."""
    def shortcut():
        # stride-2 shortcut: pack the pixels it reads once, then it (and its weight gradient) is a
        # stride-1 pointwise convolution
        xs = ops.subsample2(xin) if blk.stride == 2 and SUBSAMPLE else None
        if xs is not None:
            cd = ops.conv2d_fwd(xs, blk.downsample[0].weight, 1, 0, wp=wp, stats=training,
                                stats_shift=_ref(blk.downsample[1]))
        else:
            cd = ops.conv2d_fwd(xin, blk.downsample[0].weight, blk.stride, 0, wp=wp, stats=training,
                                stats_shift=_ref(blk.downsample[1]))
        sd = _BNState(cd, blk.downsample[1], training)
        sd.xs = xs if training else None
        return cd, sd

    cd = sd = None
    side = _side_stream(xin.device) if (blk.downsample is not None and SIDE_SHORTCUT and xin.is_cuda) else None
    if side is not None:
        # The forward is one dependent chain (every convolution waits for the previous one's batch statistics): the
        # shortcut convolution of a stage's first block depends on the block input only, so it runs on the (otherwise
        # idle) weight-gradient stream beside conv1..conv3 and fills their tails.  Its tensors are allocated while that
        # stream is current: record_stream tells the caching allocator that the main stream uses them too.
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            cd, sd = shortcut()
        xin.record_stream(side)
    c1 = ops.conv2d_fwd(xin, blk.conv1.weight, 1, 0, wp=wp, stats=training, stats_shift=_ref(blk.bn1))
    s1 = _BNState(c1, blk.bn1, training)
    c2 = ops.conv2d_fwd(c1, blk.conv2.weight, blk.stride, 1, s1.scale, s1.shift, True, wp=wp, stats=training,
                        stats_shift=_ref(blk.bn2))
    s2 = _BNState(c2, blk.bn2, training)
    c3 = ops.conv2d_fwd(c2, blk.conv3.weight, 1, 0, s2.scale, s2.shift, True, wp=wp, stats=training,
                        stats_shift=_ref(blk.bn3))
    s3 = _BNState(c3, blk.bn3, training)
    if blk.downsample is not None:
        if side is not None:
            main.wait_stream(side)
            for t in (cd, sd.scale, sd.shift, sd.mean, sd.invstd, sd.xs):
                if t is not None:
                    t.record_stream(main)
        else:
            cd, sd = shortcut()
        res, rsc, rsh = cd, sd.scale, sd.shift      # the shortcut's BatchNorm is applied while adding
    else:
        res, rsc, rsh = xin, None, None
    if training:    # the backward wants only the sign of the block output: keep 1 bit per element for it
        out, omask = ops.bn_apply(c3, s3.scale, s3.shift, res, True, want_mask=True, res_scale=rsc, res_shift=rsh)
    else:
        out, omask = ops.bn_apply(c3, s3.scale, s3.shift, res, True, res_scale=rsc, res_shift=rsh), None
    return (blk, xin, c1, s1, c2, s2, c3, s3, cd, sd, out, omask)


class _Bwd:
    """What a block's backward needs from its surroundings: where weight gradients go (flat bucket views or fresh
    tensors), the side stream they run on, the prepared data-gradient weights."""

    def __init__(self, sink, device, wp):
        self.sink, self.wp = sink, wp
        self.grads = {}
        # Weight gradients are off the critical path (nothing in this backward reads them): they run on a side
        # stream, so their ramp-up/tail and the HBM-bound BatchNorm passes of the next layer overlap.
        self.main = torch.cuda.current_stream()
        self.side = _side_stream(device)
        # bn3's backward is split: its reduce masks the incoming gradient in place (that is also the residual
        # gradient) and leaves  dc3 = ca*g + cb*c3 + cc  to conv3's two gradient kernels, which form it while loading
        # — dc3 is never written or re-read (SCAT_BNB=0: materialise it, the general path)
        self.use_bnb = BNB and ops.get_math_mode() == 1
        self.pending = [None]      # a gradient contribution not yet added to dcur (see add_ext)
        # the fixed-order sums of the weight gradients' split-K slabs are recorded by the library and performed by ONE
        # grouped launch per join() instead of one small launch behind every contraction (ops.wgrad_defer / wgrad_flush)
        ops.wgrad_defer_reset()

    def gbuf(self, p):
        return self.sink.view_for(p) if self.sink is not None else None

    def put(self, p, g):
        self.grads[p] = g

    def wgrad(self, dy, x, w, stride, pad, sc=None, sh=None, relu=False):
        out, side = self.gbuf(w), self.side
        if out is None:
            out = torch.empty_like(w)
        ops.wgrad_defer(True)
        try:
            if side is None:
                return ops.conv2d_wgrad(dy, x, tuple(w.shape), stride, pad, sc, sh, relu, out=out)
            side.wait_stream(self.main)
            with torch.cuda.stream(side):
                ops.conv2d_wgrad(dy, x, tuple(w.shape), stride, pad, sc, sh, relu, out=out, ws_slot="side")
            dy.record_stream(side)
            return out
        finally:
            ops.wgrad_defer(False)

    def join(self):
        """every weight gradient issued so far is complete on the main stream afterwards (recorded reduces included)"""
        if self.side is not None:
            with torch.cuda.stream(self.side):
                ops.wgrad_flush()
            self.main.wait_stream(self.side)
        else:
            ops.wgrad_flush()

    def wgrad_bnb(self, gm, z, coef, xop, w, sc=None, sh=None, relu=False):
        """conv weight gradient from a folded BatchNorm backward; returns (dw, event after the read of gm)"""
        out, side = self.gbuf(w), self.side
        if out is None:
            out = torch.empty_like(w)
        ops.wgrad_defer(True)
        try:
            if side is None:
                return ops.conv1x1_wgrad_bnb(gm, z, coef, xop, tuple(w.shape), sc, sh, relu, out=out), None
            side.wait_stream(self.main)
            with torch.cuda.stream(side):
                ops.conv1x1_wgrad_bnb(gm, z, coef, xop, tuple(w.shape), sc, sh, relu, out=out, ws_slot="side")
                ev = side.record_event()       # (the contraction has read gm; its recorded reduce reads only the slabs)
            gm.record_stream(side)
            coef.record_stream(side)
            return out, ev
        finally:
            ops.wgrad_defer(False)


def _fold3(bc, rec):
    """is bn3's backward of this block split (reduce here, apply inside conv3's gradient kernels)?"""
    blk, c3, omask = rec[0], rec[6], rec[11]
    return (bc.use_bnb and omask is not None and blk.conv3.weight.shape[0] % 16 == 0 and c3.shape[2] >= BNB_MIN_H)


def _block_backward(bc, rec, dcur, nxt=None, pre3=None):
    """Backward of one Bottleneck from its tape record: dcur = gradient of the block output (modified in place: the
    masked gradient is also the residual branch's gradient) -> (gradient of the block input, pre3 for the next call).

    nxt: the record of the block whose OUTPUT this block's input is (the next call's ``rec``), or None.  When this block's
    shortcut is the identity, its last kernel — conv1's data gradient, accumulated onto the masked gradient — completes the
    gradient of that block's output: armed (EPI_BNB), its epilogue applies that block's output mask and leaves the sums
    of its bn3 backward, and the next call receives them as ``pre3`` = (partials, groups) instead of running the
    reduction pass over the three tensors again."""
    blk, xin, c1, s1, c2, s2, c3, s3, cd, sd, out, omask = rec
    gbuf, put, wgrad, wgrad_bnb, wp, pending = bc.gbuf, bc.put, bc.wgrad, bc.wgrad_bnb, bc.wp, bc.pending
    # out = relu(bn3(c3) + res): g = dcur * (out>0) is also the residual branch's gradient
    fold3 = _fold3(bc, rec)
    if pending[0] is not None and not fold3:
        dcur = ops.axpy(dcur, pending[0], 1.0, out=dcur)
        pending[0] = None
    g = dcur
    ev3 = None
    w3 = blk.conv3.weight
    if fold3:
        if pre3 is not None:       # dcur IS the masked gradient already, its sums came with it
            coef3, dg, db = ops.bn_bwd_pre_partials(pre3[0], pre3[1], tuple(c3.shape), s3.mean, s3.invstd, blk.bn3.weight,
                                                    gbuf(blk.bn3.weight), gbuf(blk.bn3.bias))
        else:
            coef3, dg, db = ops.bn_bwd_pre(dcur, c3, True, s3.scale, s3.shift, s3.mean, s3.invstd, blk.bn3.weight,
                                           gbuf(blk.bn3.weight), gbuf(blk.bn3.bias), y_mask=omask,
                                           dy_add=pending[0])
        pending[0] = None
        put(blk.bn3.weight, dg), put(blk.bn3.bias, db)
        dw3, ev3 = wgrad_bnb(g, c3, coef3, c2, w3, s2.scale, s2.shift, True)   # g is overwritten further down
        put(w3, dw3)
        da2 = ops.conv1x1_dgrad_bnb(g, c3, coef3, w3, tuple(c2.shape), wp=wp)
    else:
        dc3, dg, db = ops.bn_bwd(dcur, c3, out, True, s3.scale, s3.shift, s3.mean, s3.invstd, blk.bn3.weight,
                                 gbuf(blk.bn3.weight), gbuf(blk.bn3.bias), dres=dcur, y_mask=omask)
        put(blk.bn3.weight, dg), put(blk.bn3.bias, db)
        put(w3, wgrad(dc3, c2, w3, 1, 0, s2.scale, s2.shift, True))
        da2 = ops.conv2d_dgrad_w(dc3, w3, tuple(c2.shape), 1, 0, wp=wp)
        del dc3
    ev_ds = dcd = None
    if cd is not None and SIDE_DS_BN and bc.side is not None:
        # The shortcut's BatchNorm backward needs only g: it runs on the idle fourth queue beside the conv3 -> conv2 -> conv1
        # data-gradient chain (HBM-bound next to MFMA-bound kernels) instead of after it on the main stream.  Out of place:
        # the chain is still reading g.
        from .. import streams
        aux = streams.get(g.device, "aux")
        dsbn = blk.downsample[1]
        dcd = torch.empty_like(cd)
        dgs, dbs = gbuf(dsbn.weight), gbuf(dsbn.bias)
        dgs = dgs if dgs is not None else torch.empty_like(dsbn.weight)
        dbs = dbs if dbs is not None else torch.empty_like(dsbn.bias)
        aux.wait_stream(bc.main)
        with torch.cuda.stream(aux):
            ops.bn_bwd(g, cd, None, False, sd.scale, sd.shift, sd.mean, sd.invstd, dsbn.weight, dgs, dbs, dx=dcd)
            ev_ds = aux.record_event()
        for tns in (g, cd, dcd, dgs, dbs):
            tns.record_stream(aux)
        put(dsbn.weight, dgs), put(dsbn.bias, dbs)
    dc2, dg, db = ops.bn_bwd(da2, c2, None, True, s2.scale, s2.shift, s2.mean, s2.invstd, blk.bn2.weight,
                             gbuf(blk.bn2.weight), gbuf(blk.bn2.bias), dx=da2)
    put(blk.bn2.weight, dg), put(blk.bn2.bias, db)
    put(blk.conv2.weight, wgrad(dc2, c1, blk.conv2.weight, blk.stride, 1, s1.scale, s1.shift, True))
    da1 = ops.conv2d_dgrad_w(dc2, blk.conv2.weight, tuple(c1.shape), blk.stride, 1, wp=wp)
    del dc2, da2
    w1 = blk.conv1.weight
    fold1 = (bc.use_bnb and BNB1 and (c1.shape[2] * c1.shape[3]) % 4 == 0 and w1.shape[0] % 16 == 0
             and c1.shape[2] >= BNB1_MIN_H)
    if fold1:       # same split for bn1 -> conv1 (mask recomputed from c1: relu(bn1(c1)) was never stored)
        coef1, dg, db = ops.bn_bwd_pre(da1, c1, True, s1.scale, s1.shift, s1.mean, s1.invstd, blk.bn1.weight,
                                       gbuf(blk.bn1.weight), gbuf(blk.bn1.bias))
        put(blk.bn1.weight, dg), put(blk.bn1.bias, db)
        dw1, _ = wgrad_bnb(da1, c1, coef1, xin, w1)
        put(w1, dw1)
    else:
        dc1, dg, db = ops.bn_bwd(da1, c1, None, True, s1.scale, s1.shift, s1.mean, s1.invstd, blk.bn1.weight,
                                 gbuf(blk.bn1.weight), gbuf(blk.bn1.bias), dx=da1)
        put(blk.bn1.weight, dg), put(blk.bn1.bias, db)
        put(w1, wgrad(dc1, xin, w1, 1, 0))
    if ev3 is not None:
        bc.main.wait_event(ev3)      # the side stream's conv3 weight gradient has finished reading g
    if cd is not None:
        dsw, dsbn = blk.downsample[0].weight, blk.downsample[1]
        if ev_ds is not None:
            bc.main.wait_event(ev_ds)
        else:
            dcd, dg, db = ops.bn_bwd(g, cd, None, False, sd.scale, sd.shift, sd.mean, sd.invstd, dsbn.weight,
                                     gbuf(dsbn.weight), gbuf(dsbn.bias), dx=g)
            put(dsbn.weight, dg), put(dsbn.bias, db)
        if sd.xs is not None:
            put(dsw, wgrad(dcd, sd.xs, dsw, 1, 0))
        else:
            put(dsw, wgrad(dcd, xin, dsw, blk.stride, 0))
        dxin = None     # conv1's data gradient is written first, the shortcut's lands on top of it: at
                        # stride 2 that touches only the even pixels (no zero fill, no read-modify-write
                        # of the whole plane)
    else:
        dxin = g
    pre_next = None
    if fold1:
        dcur = ops.conv1x1_dgrad_bnb(da1, c1, coef1, w1, tuple(xin.shape), out=dxin, accumulate=dxin is not None,
                                     wp=wp)
    else:
        arm = (EPI_BNB and nxt is not None and dxin is not None and pending[0] is None and _fold3(bc, nxt)
               and nxt[10].data_ptr() == xin.data_ptr())
        if arm:
            part = ops.epilogue_bnb_arm(nxt[6], nxt[11], nxt[7].mean)
        try:
            dcur = ops.conv2d_dgrad_w(dc1, w1, tuple(xin.shape), 1, 0, out=dxin, accumulate=dxin is not None, wp=wp)
        finally:
            groups = ops.epilogue_bnb_groups() if arm else 0
        if groups:
            pre_next = (part, groups)
    if cd is not None:
        dcur = ops.conv2d_dgrad_w(dcd, dsw, tuple(xin.shape), blk.stride, 0, out=dcur, accumulate=True, wp=wp)
    return dcur, pre_next


class _BlockFn(torch.autograd.Function):
    """One Bottleneck (models/resnet.py:78-98) as an autograd node of its own: ``blk(x)`` on a block taken out of
    the network.  The same two functions the fused backbone runs per block, so its parity is this block's parity."""

    @staticmethod
    def forward(ctx, x, blk, *params):
        x = x if x.is_contiguous() else x.contiguous()
        if not _DEFER[0]:
            _NBT.clear()
        # (a network that prepares all its convolution weights with one launch hands its ops.WeightPrep to its blocks: HRNet)
        rec = _block_forward(blk, x, blk.training, getattr(blk, "_wprep", None))
        if _NBT and not _DEFER[0]:
            torch._foreach_add_(_NBT, 1)
            _NBT.clear()
        out = rec[10]
        if blk.training and any(ctx.needs_input_grad):
            # the backward needs the output only for its sign: the 1-bit mask when bn_apply wrote one; on maps it
            # cannot pack (H*W % 4 != 0, e.g. layer4's 7x7) there is no mask and the output itself must stay
            ctx.rec = rec if rec[11] is None else rec[:10] + (None, rec[11])
            ctx.params = params
        else:
            ctx.rec = None
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.rec is None:
            raise RuntimeError("scat_amd: Bottleneck backward needs a training-mode forward (BN batch statistics)")
        bc = _Bwd(None, dout.device, getattr(ctx.rec[0], "_wprep", None))
        dcur = dout.contiguous().clone()               # masked in place below; the caller's tensor stays intact
        dx, _ = _block_backward(bc, ctx.rec, dcur)
        bc.join()
        params, ctx.rec = ctx.params, None
        return (dx if ctx.needs_input_grad[0] else None, None, *[bc.grads.get(p) for p in params])


class _BackboneFn(torch.autograd.Function):
    """ResNet.forward (models/resnet.py:142-162) + its whole backward as one node."""

    @staticmethod
    def forward(ctx, x, net, part, *params):
        """part 0: the whole backbone, x = image -> (feat, x1, x2, x3, x4).
        part 1: stem + layer1 + layer2, x = image -> (x1, x2);  part 2: layer3 + layer4 + head, x = x2 ->
        (feat, x3, x4).  The two halves let a caller run what depends only on x2 (the token path of hand_net)
        on another stream next to layer3/layer4, forward and backward."""
        training = net.training
        lids = {0: (0, 1, 2, 3), 1: (0, 1), 2: (2, 3)}[part]
        ctx.part = part
        # outputs the caller never uses (x1, x3, x4 on the reg_transformer path) must come back as None, not as
        # zero-filled tensors that the backward would then add stage by stage (424 MB of fills + 1.3 GB of axpy)
        ctx.set_materialize_grads(False)
        x = x if x.is_contiguous() else x.contiguous()
        tape = []
        _NBT.clear()
        wp = net._wprep
        if part != 2:
            wp.run(training)       # one launch re-lays every convolution weight for this step (ops.WeightPrep)
            # stem: conv7x7/2 -> [BN -> ReLU -> maxpool fused]
            c0 = ops.conv2d_fwd(x, net.conv1.weight, 2, 3, stats=training, stats_shift=_ref(net.bn1))
            s0 = _BNState(c0, net.bn1, training)
            cur, idx0 = ops.maxpool_fwd(c0, s0.scale, s0.shift, True)
        else:
            c0 = s0 = idx0 = None
            cur = x
        feats = []
        all_layers = (net.layer1, net.layer2, net.layer3, net.layer4)
        for layer in [all_layers[i] for i in lids]:
            for blk in layer:
                rec = _block_forward(blk, cur, training, wp)
                cur = rec[10]
                tape.append(rec)
            feats.append(cur)
        if _NBT:
            torch._foreach_add_(_NBT, 1)
            _NBT.clear()
        if part != 1:
            pooled = ops.avgpool_fwd(cur, relu=True)                      # AvgPool2d(7) -> view -> relu
            fc = ops.linear_fwd(pooled, net.fc1.weight, net.fc1.bias)
            feat = ops.relu_fwd(fc)
        else:
            pooled = feat = None
        if training and any(ctx.needs_input_grad):
            # Tensors that are also OUTPUTS of this node (x1..x4) go through save_for_backward (no
            # ctx<->output reference cycle); the tape keeps their index instead.
            def unhook(t):
                for k, f in enumerate(feats):
                    if t is f:
                        return k
                return t
            ctx.tape = [tuple(unhook(v) if isinstance(v, torch.Tensor) else v for v in rec) for rec in tape]
            ctx.net = net
            ctx.stem = (x, c0, s0, idx0)
            ctx.tail = (pooled,)
            if feat is not None:
                ctx.save_for_backward(feat, *feats)
            else:
                ctx.save_for_backward(*feats)
        else:
            ctx.net = None
        if part == 1:
            # x2 twice (same storage, two autograd outputs): one for the token path, one for second_half — the two
            # gradients then arrive separately and are combined in place on the one this module produced itself
            return (feats[0], feats[1], feats[1].view_as(feats[1]))
        return (feat, *feats) if feat is not None else tuple(feats)

    @staticmethod
    def backward(ctx, *douts):
        net = ctx.net
        if net is None:
            raise RuntimeError("scat_amd: backbone backward needs a training-mode forward (BN batch statistics)")
        part = ctx.part
        lids = {0: (0, 1, 2, 3), 1: (0, 1), 2: (2, 3)}[part]
        if part == 1:
            dfeat, feat, outs = None, None, list(ctx.saved_tensors)
            stage_grads = {0: douts[0], 1: (douts[1], douts[2])}
        else:
            dfeat = douts[0]
            feat, *outs = ctx.saved_tensors
            stage_grads = dict(zip(lids, douts[1:]))
        (pooled,) = ctx.tail
        wp = net._wprep            # data-gradient weights were re-laid with the forward ones at the start of the step
        tape = [tuple(outs[v] if isinstance(v, int) else v for v in rec) for rec in ctx.tape]
        sink = getattr(net, "_grad_sink", None)   # flat gradient buckets (scat_amd.dp.GradBuckets), or None
        if sink is not None and part != 2:
            # (split backbone: the token path's backward overlaps layer4/layer3 — the head gradients are final when
            # the first half's backward starts, which waits for the token path's input gradient)
            sink.begin_backbone()
        bc = _Bwd(sink, outs[0].device, wp)
        grads, gbuf, put, wgrad, join, pending = bc.grads, bc.gbuf, bc.put, bc.wgrad, bc.join, bc.pending

        # ---- tail: relu(fc1(relu(avgpool(x4))))
        x4 = tape[-1][-2]          # (block output; the last entry is its sign mask)
        if part == 1:
            dcur = None
        elif dfeat is not None:
            dfc = ops.relu_bwd(dfeat if dfeat.is_contiguous() else dfeat.contiguous(), feat)
            put(net.fc1.weight, ops.linear_wgrad(dfc, pooled, out=gbuf(net.fc1.weight)))
            put(net.fc1.bias, ops.colsum(dfc, out=gbuf(net.fc1.bias)))
            dpool = ops.linear_dgrad(dfc, net.fc1.weight)
            dcur = ops.avgpool_bwd(dpool, pooled, tuple(x4.shape), relu=True)
        else:
            dcur = torch.zeros_like(x4)
            if sink is not None:     # the flat bucket is all-reduced as a whole: an unused head still owes it zeros
                gbuf(net.fc1.weight).zero_()
                gbuf(net.fc1.bias).zero_()
        if sink is not None and part != 1:
            sink.ready(("fc1",))
        # ---- residual stages, last block first
        layers = (net.layer1, net.layer2, net.layer3, net.layer4)
        def add_ext(dcur, ext, like):
            """gradient entering a stage = what the stages above passed down + the caller's gradient of that stage's
            output (an incoming gradient is never modified in place: the masking below works on our own tensor)"""
            if isinstance(ext, tuple):      # (first half) x2's two gradients: token path, second half
                own = getattr(net, "_own_dx2", None)
                net._own_dx2 = None
                ext = [e for e in ext if e is not None]
                for e in ext:                  # start from the tensor second_half's backward allocated itself
                    if dcur is None and own is not None and e.data_ptr() == own and e.is_contiguous():
                        dcur = e
                        ext = [f for f in ext if f is not e]
                        break
                if (dcur is not None and len(ext) == 1 and ext[0].is_contiguous() and ext[0].data_ptr() % 16 == 0
                        and _sw.ab("SCAT_DX2_FOLD", True)):
                    pending[0] = ext[0]        # summed by the first BatchNorm-backward reduction on its way in
                    return dcur
                for e in ext:
                    dcur = add_ext(dcur, e, like)
                return dcur if dcur is not None else torch.zeros_like(like)
            if ext is None:
                return dcur if dcur is not None else torch.zeros_like(like)
            ext = ext if ext.is_contiguous() else ext.contiguous()
            if dcur is None:
                return ext.clone()
            return ops.axpy(dcur, ext, 1.0, out=dcur)

        li = lids[-1]
        remaining = len(layers[li])
        dcur = add_ext(dcur, stage_grads[li], x4)
        recs = list(reversed(tape))
        pre3 = None
        for k, rec in enumerate(recs):
            nxt = recs[k + 1] if remaining > 1 and k + 1 < len(recs) else None     # (the next block of the SAME stage)
            dcur, pre3 = _block_backward(bc, rec, dcur, nxt, pre3)
            remaining -= 1
            if remaining == 0:
                if sink is not None:
                    join()       # the all-reduce stream orders itself after the CURRENT stream only
                    sink.ready(("layer%d" % (li + 1),))
                li -= 1
                if li >= lids[0]:
                    remaining = len(layers[li])
                    dcur = add_ext(dcur, stage_grads[li], dcur)
        if part == 2:
            # second half: dcur is the gradient of its input x2; its buckets are done, the first half adopts them all
            join()
            ctx.tape = ctx.stem = ctx.tail = None
            net._own_dx2 = dcur.data_ptr()      # (first_half's backward may finish this gradient in place)
            if sink is not None:
                return (dcur, None, None, *[None for _ in net._flat_params])
            return (dcur, None, None, *[grads.get(p) for p in net._flat_params])
        # ---- stem: maxpool <- relu <- bn1 <- conv1
        x, c0, s0, idx0 = ctx.stem
        if STEM_FUSED_BWD and c0.shape[2] % 2 == 0 and c0.shape[3] % 4 == 0 and dcur.is_contiguous():
            # the max-pool's backward folded into bn1's: its 308 MB scattered gradient is never written or re-read
            dc0, dg, db = ops.bn_bwd_maxpool(dcur, idx0, c0, True, s0.scale, s0.shift, s0.mean, s0.invstd, net.bn1.weight,
                                             gbuf(net.bn1.weight), gbuf(net.bn1.bias))
        else:
            da0 = ops.maxpool_bwd(dcur, idx0, tuple(c0.shape))
            dc0, dg, db = ops.bn_bwd(da0, c0, None, True, s0.scale, s0.shift, s0.mean, s0.invstd, net.bn1.weight,
                                     gbuf(net.bn1.weight), gbuf(net.bn1.bias), dx=da0)
        put(net.bn1.weight, dg), put(net.bn1.bias, db)
        put(net.conv1.weight, wgrad(dc0, x, net.conv1.weight, 2, 3))
        join()
        ctx.tape = ctx.stem = ctx.tail = None
        # the input image needs no gradient on this path (train.py feeds data, not a leaf)
        if sink is not None:
            # gradients already sit in the flat buckets (and may be mid all-reduce on the RCCL stream):
            # hand them to the parameters directly instead of through AccumulateGrad copies
            sink.ready(("stem",))
            sink.adopt(net._flat_params)
            return (None, None, None, *[None for _ in net._flat_params])
        return (None, None, None, *[grads.get(p) for p in net._flat_params])


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        # num_classes is accepted and ignored, exactly like the reference (resnet.py:103,116)
        self.inplanes = 64
        super().__init__()
        self.conv1 = snn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = snn.BatchNorm2d(64)
        self.relu = snn.ReLU(inplace=True)
        self.maxpool = snn.MaxPool2d(kernel_size=3, stride=2, padding=1)   # (inside forward() pooling is part of the
                                                                            # fused node, with bn1+ReLU in its load)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = snn.AvgPool2d(7)
        self.fc1 = snn.Linear(512 * block.expansion, 1024)
        self._wprep = ops.WeightPrep()   # prepared convolution weights (one re-layout launch per step)
        for m in self.modules():   # same initialisers as resnet.py:118-123
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._flat_params = None

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                snn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                snn.BatchNorm2d(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        if x.shape[-1] != 224 or x.shape[-2] != 224:
            # AvgPool2d(7)+fc1(2048) fix the geometry to 224x224 in the reference too (resnet.py:115-116)
            raise RuntimeError(f"scat_amd ResNet expects 224x224 input like the reference, got {tuple(x.shape)}")
        self._flat_params = list(self.parameters())
        return _BackboneFn.apply(x, self, 0, *self._flat_params)

    # The same network as two autograd nodes, for callers that overlap x2-only work with layer3/layer4
    # (hand_net.EncoderTransformer): first_half(x) -> (x1, x2, x2 again); second_half(x2) -> (feat, x3, x4).
    def first_half(self, x):
        if x.shape[-1] != 224 or x.shape[-2] != 224:
            raise RuntimeError(f"scat_amd ResNet expects 224x224 input like the reference, got {tuple(x.shape)}")
        self._flat_params = list(self.parameters())
        return _BackboneFn.apply(x, self, 1, *self._flat_params)

    def second_half(self, x2):
        return _BackboneFn.apply(x2, self, 2, *self._flat_params)


def pretrained_checkpoint(name):
    """Where ``pretrained=True`` looks for the ImageNet weights the reference downloads (resnet.py:192-195): the file
    named by SCAT_<NAME>_CKPT (e.g. SCAT_RESNET50_CKPT), else ``$SCAT_PRETRAINED_DIR/<name>.pth``, else the torch hub
    checkpoint cache (where model_zoo.load_url would have put it).  None if there is no such file."""
    import glob

    cand = [os.environ.get("SCAT_%s_CKPT" % name.upper())]
    if os.environ.get("SCAT_PRETRAINED_DIR"):
        cand.append(os.path.join(os.environ["SCAT_PRETRAINED_DIR"], name + ".pth"))
    hub = os.path.join(os.environ.get("TORCH_HOME", os.path.expanduser("~/.cache/torch")), "hub", "checkpoints")
    cand += sorted(glob.glob(os.path.join(hub, name + "-*.pth")))
    for c in cand:
        if c and os.path.isfile(c):
            return c
    return None


def _make(name, layers, pretrained, **kwargs):
    net = ResNet(Bottleneck, layers, **kwargs)
    if pretrained:
        # the reference downloads ImageNet weights here and loads them strict=False (resnet.py:192-195); there is no
        # network on the GPU box: take them from a local file, and say so loudly when there is none — training the
        # backbone from its random initialisation is a different recipe from the reference's
        path = pretrained_checkpoint(name)
        if path is None:
            import warnings

            warnings.warn(f"scat_amd.models.resnet.{name}(pretrained=True): no local ImageNet checkpoint found (set "
                          f"SCAT_{name.upper()}_CKPT or SCAT_PRETRAINED_DIR, or call scat_amd.schedule."
                          "load_pretrained_backbone): the backbone keeps its RANDOM initialisation, unlike the "
                          "reference, which starts from ImageNet weights", RuntimeWarning, stacklevel=3)
        else:
            from ..schedule import load_pretrained_backbone

            load_pretrained_backbone(net, path)
    return net


def resnet50(pretrained=False, **kwargs):
    return _make("resnet50", [3, 4, 6, 3], pretrained, **kwargs)


def resnet101(pretrained=False, **kwargs):
    return _make("resnet101", [3, 4, 23, 3], pretrained, **kwargs)


def resnet152(pretrained=False, **kwargs):
    return _make("resnet152", [3, 8, 36, 3], pretrained, **kwargs)
