"""Attention-returning dim-halving transformer with the reference's module tree
(models/vision_transformer_attn.py:13-113): per layer ``[Attention, PreNormAttn(dim), PreNorm(dim, FF) | FF3]``,
forward ``x1, attn = Attention(x); x = LN(x1) + x; x = FF(...)`` and ``(x, attn_of_last_layer)`` returned.
Used by ``EncoderTransformerCoarse`` (train_coarse.py).  Runs as one fused node like vision_transformer.py; the
post-attention LayerNorm is a flag of the shared executor.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import nn as snn
from . import _mixer
from ._mixer import LayerCfg, TapeHolder, mixer_backward, run_mixer
from .vision_transformer import FeedForward, PreNorm  # same classes in the reference file

MIN_NUM_PATCHES = 16


class PreNormAttn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = snn.LayerNorm(dim)

    def forward(self, x, **kwargs):
        return self.norm(x)


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.to_qkv = snn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(snn.Linear(inner_dim, dim), snn.Dropout(dropout))


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList([])
        self._cfgs = []
        for l in range(depth):
            last = l == depth - 1
            ff = FeedForward(dim, (dim * 3) // 4, out_dim=3) if last else PreNorm(dim, FeedForward(dim, (dim * 3) // 4))
            self.layers.append(nn.ModuleList([Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout),
                                              PreNormAttn(dim), ff]))
            self._cfgs.append(LayerCfg(False, not last, False, dim_head ** -0.5, heads, dim_head, post_ln=True))
            if not last:
                dim = dim // 2
        self._holder = TapeHolder()

    def _params(self):
        ps = []
        for (att, pren, ff), cfg in zip(self.layers, self._cfgs):
            ps += [att.to_qkv.weight, att.to_out[0].weight, att.to_out[0].bias, pren.norm.weight, pren.norm.bias]
            if cfg.ff_ln:
                ps += [ff.norm.weight, ff.norm.bias]
                net = ff.fn.net
            else:
                net = ff.net
            ps += [net[0].weight, net[0].bias, net[2].weight, net[2].bias]
        return ps

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("scat_amd: attention mask is never used on the reference path")
        y = run_mixer(x, self._holder, self._cfgs, self._params())
        return y, _mixer.last_attn[0]

    def input_grad(self, dy):
        if self._holder.tape is None:
            raise RuntimeError("scat_amd: input_grad needs a forward that kept its tape")
        with torch.no_grad():
            return mixer_backward(self._holder.tape, dy, want_param_grads=False)[0]
