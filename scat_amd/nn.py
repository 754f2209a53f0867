"""Parameter-holder modules and single-op autograd Functions on the HIP library.

The module classes subclass torch.nn containers ONLY to inherit parameter registration,
``state_dict`` key layout and default initialisation (so checkpoints interchange with the
reference). Their ``forward`` always runs libscat_hip kernels; there is no torch
compute fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Conv2dFn(torch.autograd.Function):
    """nn.Conv2d forward/backward (models/resnet.py:65-72; hand_net.py:329) on the MFMA engine."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, wp=None):
        x, w = _c(x), _c(w)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, bias is not None)
        ctx.wp = wp               # the owning network's ops.WeightPrep (prepared weights), or None
        return ops.conv2d_fwd(x, w, stride, pad, bias=bias, wp=wp)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        dy = _c(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_dgrad_w(dy, w, tuple(x.shape), stride, pad, wp=ctx.wp)
        if ctx.needs_input_grad[1]:
            dw = ops.conv2d_wgrad(dy, x, tuple(w.shape), stride, pad)
        if has_bias and ctx.needs_input_grad[2]:
            B, C, H, W = dy.shape
            db = ops.colsum(_c(dy.permute(0, 2, 3, 1).reshape(-1, C)))
        return dx, dw, db, None, None, None


class _LinearFn(torch.autograd.Function):
    """nn.Linear (models/resnet.py:116; hand_net.py:353; vision_transformer.py:33-35)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x2 = _c(x).reshape(-1, x.shape[-1])
        w = _c(w)
        ctx.save_for_backward(x2, w)
        ctx.shape = x.shape
        ctx.has_bias = bias is not None
        return ops.linear_fwd(x2, w, bias).reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = _c(dy).reshape(-1, dy.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_dgrad(dy2, w).reshape(ctx.shape)
        if ctx.needs_input_grad[1]:
            dw = ops.linear_wgrad(dy2, x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dy2)
        return dx, dw, db


class _ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.relu_fwd(_c(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.relu_bwd(_c(dy), y)


class _GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(_c(dy), x)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        x2 = _c(x).reshape(-1, x.shape[-1])
        y, mean, rstd = ops.layernorm_fwd(x2, g, b, eps)
        ctx.save_for_backward(x2, g, mean, rstd)
        ctx.shape = x.shape
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, g, mean, rstd = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(_c(dy).reshape(x2.shape), x2, g, mean, rstd)
        return dx.reshape(ctx.shape), dg, db, None


class _BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm2d (+ optional fused ReLU), train and eval (models/resnet.py:68-73)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, training, momentum, eps, relu):
        x = _c(x)
        if training:
            mean, invstd, scale, shift = ops.bn_train_stats(x, gamma, beta, rm, rv, momentum, eps)
        else:
            scale, shift = ops.bn_eval_fold(gamma, beta, rm, rv, eps)
            mean = invstd = None
        y = ops.bn_apply(x, scale, shift, None, relu)
        ctx.training, ctx.relu = training, relu
        if training:
            ctx.save_for_backward(x, gamma, mean, invstd, scale, shift)
        else:
            ctx.save_for_backward(x, gamma, scale, shift)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        if ctx.training:
            x, gamma, mean, invstd, scale, shift = ctx.saved_tensors
            dx, dg, db = ops.bn_bwd(dy, x, None, ctx.relu, scale, shift, mean, invstd, gamma)
            return dx, dg, db, None, None, None, None, None, None
        raise RuntimeError("scat_amd: BatchNorm backward in eval mode is not on the reference's path "
                           "(eval.py runs forward only)")


def conv2d(x, w, bias=None, stride=1, pad=0, wp=None):
    return _Conv2dFn.apply(x, w, bias, stride, pad, wp)


def linear(x, w, bias=None):
    return _LinearFn.apply(x, w, bias)


class Conv2d(nn.Conv2d):
    def forward(self, x):
        assert self.kernel_size[0] == self.kernel_size[1] and self.stride[0] == self.stride[1]
        assert self.dilation == (1, 1) and self.groups == 1
        # _wprep: set by a network that prepares all its convolution weights with one launch (HRNet)
        return conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], getattr(self, "_wprep", None))


class Linear(nn.Linear):
    def forward(self, x):
        return linear(x, self.weight, self.bias)


class ReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return _ReluFn.apply(x)


class GELU(nn.Module):
    def forward(self, x):
        return _GeluFn.apply(x)


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return _LayerNormFn.apply(x, self.weight, self.bias, self.eps)


class BatchNorm2d(nn.BatchNorm2d):
    fuse_relu = False

    # ``_stat_ref`` (models/resnet.py _BNState): the previous training step's batch mean, the reference the convolution
    # epilogue takes its BatchNorm sums about.  It is step-to-step scratch, not state: dropped whenever the step sequence
    # is broken (a checkpoint is loaded, the module leaves training mode), so a resumed run starts like a fresh one.
    def _load_from_state_dict(self, *args, **kwargs):
        self.__dict__.pop("_stat_ref", None)
        return super()._load_from_state_dict(*args, **kwargs)

    def train(self, mode=True):
        if not mode:
            self.__dict__.pop("_stat_ref", None)
        return super().train(mode)

    def forward(self, x):
        if self.training and self.track_running_stats:
            self.num_batches_tracked += 1
        return _BatchNormFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                  self.momentum, self.eps, self.fuse_relu)


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout(_c(x), p, seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(_c(dy), ctx.p, ctx.seed), None, None


class Dropout(nn.Module):
    """nn.Dropout.  p = 0 on the reg_transformer path (hand_net.py:331); the performer block uses p = 0.1 in train
    mode (vision_performer.py:18,28).  The mask comes from a counter hash seeded from python ``random`` — the same
    distribution as torch's, not the same bits (torch's Philox stream is not reproducible off-device either)."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.p > 0 and self.training:
            import random

            return _DropoutFn.apply(x, self.p, random.getrandbits(63))
        return x


class _UpsampleFn(torch.autograd.Function):
    """nn.Upsample(scale_factor=2**k, mode='nearest') (models/hrnet.py:107)."""

    @staticmethod
    def forward(ctx, x, factor):
        ctx.factor = factor
        return ops.upsample_nearest_fwd(_c(x), factor)

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample_nearest_bwd(_c(dy), ctx.factor), None


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.axpy(_c(a), _c(b), 1.0)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class _TokenMeanFn(torch.autograd.Function):
    """x.mean(dim=1) over tokens (models/hand_net.py:203; models/vision_performer.py:108)."""

    @staticmethod
    def forward(ctx, x):
        ctx.T = x.shape[1]
        return ops.token_mean_fwd(_c(x))

    @staticmethod
    def backward(ctx, dy):
        return ops.token_mean_bwd(_c(dy), ctx.T)


def add(a, b):
    return _AddFn.apply(a, b)


def token_mean(x):
    return _TokenMeanFn.apply(x)


class Upsample(nn.Module):
    def __init__(self, scale_factor, mode="nearest"):
        super().__init__()
        assert mode == "nearest" and float(scale_factor) == int(scale_factor)
        self.scale_factor, self.mode = float(scale_factor), mode

    def forward(self, x):
        return _UpsampleFn.apply(x, int(self.scale_factor))


class _MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (models/resnet.py:112): int8 arg-max kept for the backward."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        y, idx = ops.maxpool_fwd(x)
        ctx.save_for_backward(idx)
        ctx.xs = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return ops.maxpool_bwd(_c(dy), idx, ctx.xs)


class _GlobalAvgPoolFn(torch.autograd.Function):
    """nn.AvgPool2d(k) on a k x k plane (models/resnet.py:115 on the 7x7 map) -> [B,C,1,1]"""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        y = ops.avgpool_fwd(x, relu=False)
        ctx.save_for_backward(y)
        ctx.xs = tuple(x.shape)
        return y.view(x.shape[0], x.shape[1], 1, 1)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.avgpool_bwd(_c(dy.reshape(y.shape)), y, ctx.xs, relu=False)


class MaxPool2d(nn.MaxPool2d):
    """The one geometry the path has: 3x3, stride 2, padding 1."""

    def forward(self, x):
        if (self.kernel_size, self.stride, self.padding) != (3, 2, 1) or self.dilation != 1 or self.ceil_mode:
            raise RuntimeError("scat_amd.nn.MaxPool2d: only kernel_size=3, stride=2, padding=1 (models/resnet.py:112)")
        return _MaxPoolFn.apply(x)


class AvgPool2d(nn.AvgPool2d):
    """AvgPool2d(k) applied to a k x k map (the reference's AvgPool2d(7) on x4): a per-channel mean."""

    def forward(self, x):
        k = self.kernel_size if isinstance(self.kernel_size, int) else self.kernel_size[0]
        if x.shape[-1] != k or x.shape[-2] != k or self.padding != 0:
            raise RuntimeError(f"scat_amd.nn.AvgPool2d({k}): the input plane must be {k}x{k} (models/resnet.py:115), "
                               f"got {tuple(x.shape)}")
        return _GlobalAvgPoolFn.apply(x)
