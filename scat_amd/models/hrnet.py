"""HRNet backbone with the reference's module tree (models/hrnet.py:10-261 of tomguluson92/SCAT):
``HRNet(c, nof_joints, bn_momentum)``, stem (two stride-2 3x3 convs), ``layer1`` (4 Bottlenecks),
``transition1..3``, ``stage2..4`` of ``StageModule`` (4 BasicBlocks per branch + fuse layers with
1x1+BN+nearest-upsample / strided 3x3 chains), ``final_layer`` 1x1 with bias — identical
``state_dict`` keys.  Every conv / BN / ReLU / upsample / add runs on libscat_hip kernels; the BasicBlocks of the
stage branches (208 of the 293 convs) execute as one fused autograd node each (``_BasicBlockFn``), the rest as
per-layer nodes.

Quirk kept: ``BasicBlock.conv2`` is declared ``inplanes -> planes`` (hrnet.py:56), harmless because
inplanes == planes wherever it is used.
"""
from __future__ import annotations

import torch

from .. import _switches as _sw
import torch.nn as nn

from .. import nn as snn
from .. import ops
from . import resnet as _rn


class _BasicBlockFn(torch.autograd.Function):
    """A stride-1 BasicBlock without projection (every block of the StageModule branches: 208 of HRNet-W32's 293
    convs) as ONE autograd node, the same way the ResNet Bottleneck is executed: bn1+ReLU is never materialised (folded
    into conv2's operand load forward and into the weight gradient backward, the mask recomputed from the raw conv
    output), bn2 + residual add + ReLU is one pass that also leaves a 1-bit sign mask for the backward, the BatchNorm
    backward runs in place, and conv1's data gradient accumulates straight into the residual gradient."""

    @staticmethod
    def forward(ctx, x, blk, w1, g1, b1, w2, g2, b2):
        training = blk.training
        x = x if x.is_contiguous() else x.contiguous()
        wp = getattr(blk, "_wprep", None)
        c1 = ops.conv2d_fwd(x, w1, 1, 1, wp=wp, stats=training and EPI_STATS)
        s1 = _rn._BNState(c1, blk.bn1, training)
        c2 = ops.conv2d_fwd(c1, w2, 1, 1, s1.scale, s1.shift, True, wp=wp, stats=training and EPI_STATS)
        s2 = _rn._BNState(c2, blk.bn2, training)
        if _rn._NBT and not _DEFER_NBT[0]:
            torch._foreach_add_(_rn._NBT, 1)
            _rn._NBT.clear()
        need = training and any(ctx.needs_input_grad)
        if need:
            out, mask = ops.bn_apply(c2, s2.scale, s2.shift, x, True, want_mask=True)
            ctx.blk, ctx.s1, ctx.s2, ctx.has_mask = blk, s1, s2, mask is not None
            ctx.save_for_backward(x, c1, c2, mask if mask is not None else out)
        else:
            out = ops.bn_apply(c2, s2.scale, s2.shift, x, True)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, c1, c2, m = ctx.saved_tensors
        blk, s1, s2 = ctx.blk, ctx.s1, ctx.s2
        dout = dout if dout.is_contiguous() else dout.contiguous()
        g = torch.empty_like(dout)          # masked gradient = the residual branch's gradient (dout itself is autograd's)
        dc2, dg2, db2 = ops.bn_bwd(dout, c2, None if ctx.has_mask else m, True, s2.scale, s2.shift, s2.mean, s2.invstd,
                                   blk.bn2.weight, dres=g, y_mask=m if ctx.has_mask else None)
        dw2 = ops.conv2d_wgrad(dc2, c1, tuple(blk.conv2.weight.shape), 1, 1, s1.scale, s1.shift, True)
        wp = getattr(blk, "_wprep", None)
        da1 = ops.conv2d_dgrad_w(dc2, blk.conv2.weight, tuple(c1.shape), 1, 1, wp=wp)
        del dc2
        dc1, dg1, db1 = ops.bn_bwd(da1, c1, None, True, s1.scale, s1.shift, s1.mean, s1.invstd, blk.bn1.weight, dx=da1)
        dw1 = ops.conv2d_wgrad(dc1, x, tuple(blk.conv1.weight.shape), 1, 1)
        dx = ops.conv2d_dgrad_w(dc1, blk.conv1.weight, tuple(x.shape), 1, 1, out=g, accumulate=True, wp=wp)
        return dx, None, dw1, dg1, db1, dw2, dg2, db2


class _CbrFn(torch.autograd.Function):
    """conv -> BatchNorm -> ReLU (models/hrnet.py: the stem, the transitions, the inner units of the strided fuse chains) as
    ONE autograd node instead of three: the BatchNorm's sums come from the convolution's epilogue where its kernel has one
    (else one pass), normalise + ReLU is one pass, nothing but the raw convolution output is kept for the backward (the
    ReLU sign is recomputed from it inside the BatchNorm backward) — no separate ReLU forward / backward passes."""

    @staticmethod
    def forward(ctx, x, conv, bn, w, gamma, beta):
        training = bn.training
        x = x if x.is_contiguous() else x.contiguous()
        wp = getattr(conv, "_wprep", None)
        st, pd = conv.stride[0], conv.padding[0]
        c = ops.conv2d_fwd(x, w, st, pd, wp=wp, stats=training and EPI_STATS_CBR,
                           stats_shift=_rn._ref(bn) if training else None)
        s = _rn._BNState(c, bn, training)
        if _rn._NBT and not _DEFER_NBT[0]:
            torch._foreach_add_(_rn._NBT, 1)
            _rn._NBT.clear()
        out = ops.bn_apply(c, s.scale, s.shift, None, True)
        ctx.rec = training and any(ctx.needs_input_grad)
        if ctx.rec:
            ctx.conv, ctx.bn, ctx.s = conv, bn, s
            ctx.save_for_backward(x, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.rec:     # (callers take the per-layer path for an eval-mode BatchNorm under grad: _fused_ok)
            raise RuntimeError("scat_amd: conv+BatchNorm+ReLU backward needs a training-mode forward (BN batch statistics)")
        x, c = ctx.saved_tensors
        conv, bn, s = ctx.conv, ctx.bn, ctx.s
        st, pd = conv.stride[0], conv.padding[0]
        dout = dout if dout.is_contiguous() else dout.contiguous()
        dc, dg, db = ops.bn_bwd(dout, c, None, True, s.scale, s.shift, s.mean, s.invstd, bn.weight)
        dw = ops.conv2d_wgrad(dc, x, tuple(conv.weight.shape), st, pd) if ctx.needs_input_grad[3] else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_dgrad_w(dc, conv.weight, tuple(x.shape), st, pd, wp=getattr(conv, "_wprep", None))
        return dx, None, None, dw, dg, db


class _FuseSumFn(torch.autograd.Function):
    """One output of an exchange unit (models/hrnet.py:117-144): relu(sum_j term_j) as one node and ONE forward pass.  A
    term is the branch's own map, or the raw output of the last convolution of fuse_layers[i][j] together with that
    layer's BatchNorm (and its upsample factor 2^k): the BatchNorm's statistics are taken here, its normalise, the
    upsample, the additions (in the reference's order) and the ReLU are scat_fuse_sum.  Backward: the masked gradient is
    the gradient of every same-resolution term, its 2^k x 2^k block sums that of an upsampled one; each BatchNorm's
    backward follows."""

    @staticmethod
    def forward(ctx, spec, *flat):
        # spec: tuple of (bn module | None, k) per term, in order; flat: the terms' tensors, then (gamma, beta) of every
        # term that has a BatchNorm
        n = len(spec)
        tens = [t if t.is_contiguous() else t.contiguous() for t in flat[:n]]
        training = any(bn is not None and bn.training for bn, _ in spec)
        states, terms = [], []
        for (bn, k), t in zip(spec, tens):
            if bn is None:
                states.append(None)
                terms.append((t, None, None, k))
            else:
                st = _rn._BNState(t, bn, bn.training)
                states.append(st)
                terms.append((t, st.scale, st.shift, k))
        if _rn._NBT and not _DEFER_NBT[0]:
            torch._foreach_add_(_rn._NBT, 1)
            _rn._NBT.clear()
        out = ops.fuse_sum(terms, relu=True)
        ctx.rec = training and any(ctx.needs_input_grad)
        if ctx.rec:
            ctx.spec, ctx.states = spec, states
            ctx.save_for_backward(out, *tens)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.rec:
            raise RuntimeError("scat_amd: exchange-unit backward needs a training-mode forward (BN batch statistics)")
        out, *tens = ctx.saved_tensors
        spec, states = ctx.spec, ctx.states
        n = len(spec)
        gm = ops.relu_bwd(dout if dout.is_contiguous() else dout.contiguous(), out)
        grads, pgrads = [], []
        down = {0: gm}
        for (bn, k), st, t in zip(spec, states, tens):
            if k not in down:
                down[k] = ops.upsample_nearest_bwd(gm, 1 << k)
            g = down[k]
            if bn is None:
                grads.append(g)
            else:
                dc, dg, db = ops.bn_bwd(g, t, None, False, st.scale, st.shift, st.mean, st.invstd, bn.weight)
                grads.append(dc)
                pgrads += [dg, db]
        return (None, *grads, *pgrads)


def _fused_ok(*bns):
    """The fused nodes keep what their backward needs only in training mode (batch statistics).  A backward through an
    eval-mode BatchNorm (frozen-BN fine-tuning, saliency maps) is legal for nn.BatchNorm2d, so those calls take the
    per-layer autograd path instead."""
    return not torch.is_grad_enabled() or all(bn.training for bn in bns)


class _CBR(nn.Sequential):
    """nn.Sequential(Conv2d, BatchNorm2d, ReLU) with the reference's state_dict keys, executed as one node"""

    def forward(self, x):
        if FUSED_CBR and x.is_cuda and self[0].bias is None and _fused_ok(self[1]):
            return _CbrFn.apply(x, self[0], self[1], self[0].weight, self[1].weight, self[1].bias)
        return super().forward(x)


import os

_DEFER_NBT = [False]     # set by HRNet.forward: the blocks leave their num_batches_tracked bumps to its end

# BatchNorm sums in the convolution epilogue: measured a wash on these short contractions (K = 288 .. 2304 with the
# statistics passes already hidden on the branch streams: 70.0 vs 69.0 ms / step), so off here; ResNet-50: +1.8 %
EPI_STATS = _sw.ab("SCAT_HRNET_EPI", False)
FUSED_BASIC = _sw.ab("SCAT_HRNET_FUSED", True)   # 0: the per-layer autograd path, for A/B runs
FUSED_EXCHANGE = _sw.ab("SCAT_HRNET_FUSED_X", True)   # an exchange output's BatchNorms, upsamples, adds and ReLU: one node
FUSED_CBR = _sw.ab("SCAT_HRNET_FUSED_CBR", True)   # conv + BatchNorm + ReLU units as one node each
EPI_STATS_CBR = _sw.ab("SCAT_HRNET_CBR_EPI", True)   # ... with the BatchNorm sums from the convolution epilogue
FUSED_BOTTLENECK = _sw.ab("SCAT_HRNET_FUSED_L1", True)   # layer1's Bottlenecks on ResNet's block executor
PARALLEL_BRANCHES = _sw.ab("SCAT_HRNET_PAR", True)   # one stream per resolution branch of a stage
STAR_EXCHANGE = _sw.ab("SCAT_HRNET_STAR", True)   # the exchange's barrier through the caller's stream + relay nodes


class _RelayFn(torch.autograd.Function):
    """The same tensor as a node of the CALLER's stream (no kernel in either direction): autograd then orders a branch
    stream against the caller's stream only, never against another branch's (StageModule.forward)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g


def _branch_stream(device, i):
    from .. import streams
    return streams.get(device, "branch%d" % i)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, bn_momentum=0.1):
        super().__init__()
        self.conv1 = snn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = snn.BatchNorm2d(planes, momentum=bn_momentum)
        self.conv2 = snn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = snn.BatchNorm2d(planes, momentum=bn_momentum)
        self.conv3 = snn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = snn.BatchNorm2d(planes * self.expansion, momentum=bn_momentum)
        self.relu = snn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if FUSED_BOTTLENECK and x.is_cuda:
            # layer1's four Bottlenecks are ResNet-50's layer1 (models/resnet.py:78-98, 64 / 256 channels at 56 x 56): the
            # same fused block executor — BatchNorm sums in the convolution epilogues, bn1 / bn2 + ReLU folded into the next
            # convolution's operand load, bn3 + shortcut + ReLU one pass with a 1-bit sign mask, folded bn3 backward
            return _rn._BlockFn.apply(x, self, *self.parameters())
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        residual = x if self.downsample is None else self.downsample(x)
        return self.relu(snn.add(out, residual))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, bn_momentum=0.1):
        super().__init__()
        self.conv1 = snn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = snn.BatchNorm2d(planes, momentum=bn_momentum)
        self.relu = snn.ReLU(inplace=True)
        self.conv2 = snn.Conv2d(inplanes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = snn.BatchNorm2d(planes, momentum=bn_momentum)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if FUSED_BASIC and self.downsample is None and self.stride == 1 and x.is_cuda:
            return _BasicBlockFn.apply(x, self, self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight,
                                       self.bn2.weight, self.bn2.bias)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        residual = x if self.downsample is None else self.downsample(x)
        return self.relu(snn.add(out, residual))


class StageModule(nn.Module):
    def __init__(self, stage, output_branches, c, bn_momentum):
        super().__init__()
        self.stage = stage
        self.output_branches = output_branches
        self.branches = nn.ModuleList()
        for i in range(stage):
            w = c * (2 ** i)
            self.branches.append(nn.Sequential(*[BasicBlock(w, w, bn_momentum=bn_momentum) for _ in range(4)]))
        self.fuse_layers = nn.ModuleList()
        for i in range(output_branches):
            self.fuse_layers.append(nn.ModuleList())
            for j in range(stage):
                if i == j:
                    self.fuse_layers[-1].append(nn.Sequential())
                elif i < j:
                    self.fuse_layers[-1].append(nn.Sequential(
                        snn.Conv2d(c * (2 ** j), c * (2 ** i), kernel_size=(1, 1), stride=(1, 1), bias=False),
                        snn.BatchNorm2d(c * (2 ** i), eps=1e-05, momentum=0.1),
                        snn.Upsample(scale_factor=(2.0 ** (j - i)), mode="nearest")))
                else:
                    chain = []
                    for _ in range(i - j - 1):
                        chain.append(_CBR(
                            snn.Conv2d(c * (2 ** j), c * (2 ** j), kernel_size=(3, 3), stride=(2, 2), padding=(1, 1),
                                       bias=False),
                            snn.BatchNorm2d(c * (2 ** j), eps=1e-05, momentum=0.1),
                            snn.ReLU(inplace=True)))
                    chain.append(nn.Sequential(
                        snn.Conv2d(c * (2 ** j), c * (2 ** i), kernel_size=(3, 3), stride=(2, 2), padding=(1, 1),
                                   bias=False),
                        snn.BatchNorm2d(c * (2 ** i), eps=1e-05, momentum=0.1)))
                    self.fuse_layers[-1].append(nn.Sequential(*chain))
        self.relu = snn.ReLU(inplace=True)

    def _fuse(self, i, x):
        if (FUSED_EXCHANGE and x[0].is_cuda and len(self.branches) <= 4
                and _fused_ok(*[m for m in self.fuse_layers[i].modules() if isinstance(m, nn.BatchNorm2d)])):
            # the convolutions stay nodes of their own; everything after the last one of each term is one node
            spec, tens, params = [], [], []
            for j in range(len(self.branches)):
                layer = self.fuse_layers[i][j]
                if j == i:
                    spec.append((None, 0))
                    tens.append(x[j])
                elif i < j:                                  # 1x1 convolution, BatchNorm, nearest upsample by 2^(j-i)
                    spec.append((layer[1], j - i))
                    tens.append(layer[0](x[j]))
                    params += [layer[1].weight, layer[1].bias]
                else:                                        # strided 3x3 chain: the last unit is convolution + BatchNorm
                    h = x[j]
                    for unit in layer[:-1]:
                        h = unit(h)
                    spec.append((layer[-1][1], 0))
                    tens.append(layer[-1][0](h))
                    params += [layer[-1][1].weight, layer[-1][1].bias]
            return _FuseSumFn.apply(tuple(spec), *tens, *params)
        acc = self.fuse_layers[i][0](x[0])
        for j in range(1, len(self.branches)):
            acc = snn.add(acc, self.fuse_layers[i][j](x[j]))
        return self.relu(acc)

    def forward(self, x):
        assert len(self.branches) == len(x)
        n, nout = len(self.branches), len(self.fuse_layers)
        if not (PARALLEL_BRANCHES and n > 1 and x[0].is_cuda):
            x = [branch(b) for branch, b in zip(self.branches, x)]
            return [self._fuse(i, x) for i in range(nout)]
        # The branches of a stage are independent until the exchange, each does the same number of FLOPs
        # (channels double, pixels quarter) and none of them fills the GPU on its own — the low-resolution ones are
        # a few dozen workgroups per kernel.  Branch i runs on stream i (stream 0 = the caller's), so does exchange
        # output i; autograd replays each node's backward on its forward stream, so the backward overlaps the same way.
        main = torch.cuda.current_stream()
        sts = [main] + [_branch_stream(x[0].device, i) for i in range(1, max(n, nout))]
        outs = [None] * n
        for i in range(n - 1, -1, -1):
            if i:
                sts[i].wait_stream(main)
            with torch.cuda.stream(sts[i]):
                outs[i] = self.branches[i](x[i])
            if i:
                x[i].record_stream(sts[i])
        # every exchange output reads every branch: a barrier.  Routed through the caller's stream (it waits for the
        # branches, the branches then wait for it) rather than branch-to-branch: the same ordering, and a stream
        # capture's fork / join graph stays a star around the capturing stream (hipStreamEndCapture does not survive
        # side streams that wait on each other).  Autograd replays each node on its forward's stream and orders producer
        # and consumer streams itself, so a branch output that another branch's stream reads passes through a relay node
        # on the caller's stream: in the backward, too, the branch streams only ever meet the caller's.
        relay = list(outs)
        if STAR_EXCHANGE:
            for j in range(1, n):
                main.wait_stream(sts[j])
            for j in range(1, n):
                relay[j] = _RelayFn.apply(outs[j])
        fused = [None] * nout
        for i in range(nout - 1, -1, -1):
            if STAR_EXCHANGE:
                if i:
                    sts[i].wait_stream(main)
            else:                      # (A/B runs: every exchange stream waits for every other branch's stream directly)
                for j in range(n):
                    if j != i:
                        sts[i].wait_stream(sts[j])
            srcs = [outs[j] if j == i else relay[j] for j in range(n)]
            for j in range(n):
                if j != i:
                    outs[j].record_stream(sts[i])
            with torch.cuda.stream(sts[i]):
                fused[i] = self._fuse(i, srcs)
        for i in range(1, max(n, nout)):
            main.wait_stream(sts[i])
        for i in range(1, nout):
            fused[i].record_stream(main)
        return fused


class HRNet(nn.Module):
    def __init__(self, c=48, nof_joints=17, bn_momentum=0.1):
        super().__init__()
        m = bn_momentum
        self.conv1 = snn.Conv2d(3, 64, kernel_size=(3, 3), stride=(2, 2), padding=(1, 1), bias=False)
        self.bn1 = snn.BatchNorm2d(64, eps=1e-05, momentum=m)
        self.conv2 = snn.Conv2d(64, 64, kernel_size=(3, 3), stride=(2, 2), padding=(1, 1), bias=False)
        self.bn2 = snn.BatchNorm2d(64, eps=1e-05, momentum=m)
        self.relu = snn.ReLU(inplace=True)
        downsample = nn.Sequential(snn.Conv2d(64, 256, kernel_size=(1, 1), stride=(1, 1), bias=False),
                                   snn.BatchNorm2d(256, eps=1e-05, momentum=m))
        self.layer1 = nn.Sequential(Bottleneck(64, 64, downsample=downsample), Bottleneck(256, 64),
                                    Bottleneck(256, 64), Bottleneck(256, 64))

        def cbr(cin, cout, stride):
            return _CBR(snn.Conv2d(cin, cout, kernel_size=(3, 3), stride=(stride, stride), padding=(1, 1), bias=False),
                        snn.BatchNorm2d(cout, eps=1e-05, momentum=m), snn.ReLU(inplace=True))

        self.transition1 = nn.ModuleList([cbr(256, c, 1), nn.Sequential(cbr(256, c * 2, 2))])
        self.stage2 = nn.Sequential(StageModule(2, 2, c, m))
        self.transition2 = nn.ModuleList([nn.Sequential(), nn.Sequential(), nn.Sequential(cbr(c * 2, c * 4, 2))])
        self.stage3 = nn.Sequential(*[StageModule(3, 3, c, m) for _ in range(4)])
        self.transition3 = nn.ModuleList([nn.Sequential(), nn.Sequential(), nn.Sequential(),
                                          nn.Sequential(cbr(c * 4, c * 8, 2))])
        self.stage4 = nn.Sequential(StageModule(4, 4, c, m), StageModule(4, 4, c, m), StageModule(4, 1, c, m))
        self.final_layer = snn.Conv2d(c, nof_joints, kernel_size=(1, 1), stride=(1, 1))
        # prepared weights: every convolution of the network shares one ops.WeightPrep, re-laid by ONE launch at the
        # start of forward instead of 2 x 293 small ones spread over the step
        object.__setattr__(self, "_wprep", ops.WeightPrep())
        for m in self.modules():
            if isinstance(m, (snn.Conv2d, BasicBlock, Bottleneck)):
                object.__setattr__(m, "_wprep", self._wprep)

    def forward(self, x):
        # the 208 fused blocks bump their num_batches_tracked counters with ONE multi-tensor add at the end of the
        # forward instead of one tiny launch per block
        _DEFER_NBT[0] = _rn._DEFER[0] = True
        try:
            return self._forward(x)
        finally:
            _DEFER_NBT[0] = _rn._DEFER[0] = False
            if _rn._NBT:
                torch._foreach_add_(_rn._NBT, 1)
                _rn._NBT.clear()

    def _forward(self, x):
        self._wprep.run(self.training)
        if FUSED_CBR and x.is_cuda and _fused_ok(self.bn1, self.bn2):
            x = _CbrFn.apply(x, self.conv1, self.bn1, self.conv1.weight, self.bn1.weight, self.bn1.bias)
            x = _CbrFn.apply(x, self.conv2, self.bn2, self.conv2.weight, self.bn2.weight, self.bn2.bias)
        else:
            x = self.relu(self.bn1(self.conv1(x)))
            x = self.relu(self.bn2(self.conv2(x)))
        x = self.layer1(x)
        x = [trans(x) for trans in self.transition1]
        x = self.stage2(x)
        x = [self.transition2[0](x[0]), self.transition2[1](x[1]), self.transition2[2](x[-1])]
        x = self.stage3(x)
        x = [self.transition3[0](x[0]), self.transition3[1](x[1]), self.transition3[2](x[2]),
             self.transition3[3](x[-1])]
        x = self.stage4(x)
        return self.final_layer(x[0])
