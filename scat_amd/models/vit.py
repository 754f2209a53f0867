"""Dim-preserving transformer with the reference's module tree (models/vit.py:8-101):
``layers[l] = [Residual(Attention), Residual(FeedForward)]``, no LayerNorm in the stack, softmax
scale ``dim**-0.5``; ``FeedForward.net = [Linear, GELU, Dropout, Linear, Dropout]`` so the second
Linear keeps index 3 in ``state_dict``.  Fused execution as in vision_transformer.py.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import nn as snn
from ._mixer import LayerCfg, TapeHolder, mixer_backward, run_mixer
from .vision_transformer import Attention as _VTAttention
from .vision_transformer import PreNorm, Residual  # noqa: F401  (same helper classes as the reference file)

MIN_NUM_PATCHES = 16


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(snn.Linear(dim, hidden_dim), snn.GELU(), snn.Dropout(dropout),
                                 snn.Linear(hidden_dim, dim), snn.Dropout(dropout))

    def forward(self, x):
        return self.net(x)


class Attention(_VTAttention):
    scale_from_dim = True


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout):
        super().__init__()
        if dropout:
            raise NotImplementedError("scat_amd: dropout > 0 is not implemented on the HIP path")
        self.layers = nn.ModuleList([])
        self._cfgs = []
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                Residual(Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                Residual(FeedForward(dim, mlp_dim, dropout=dropout))]))
            self._cfgs.append(LayerCfg(False, False, True, dim ** -0.5, heads, dim_head))
        self._holder = TapeHolder()

    def _params(self):
        ps = []
        for attn, ff in self.layers:
            ps += [attn.fn.to_qkv.weight, attn.fn.to_out[0].weight, attn.fn.to_out[0].bias,
                   ff.fn.net[0].weight, ff.fn.net[0].bias, ff.fn.net[3].weight, ff.fn.net[3].bias]
        return ps

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("scat_amd: attention mask is never used on the reference path")
        return run_mixer(x, self._holder, self._cfgs, self._params())

    def input_grad(self, dy):
        if self._holder.tape is None:
            raise RuntimeError("scat_amd: input_grad needs a forward that kept its tape")
        with torch.no_grad():
            return mixer_backward(self._holder.tape, dy, want_param_grads=False)[0]


class YunqianTransformer(nn.Module):
    """models/vit.py:86-101: transformer + LayerNorm/Linear head."""

    def __init__(self, dim, depth, heads, mlp_dim, dim_head=64, out_dim=61, dropout=0.):
        super().__init__()
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Sequential(snn.LayerNorm(dim), snn.Linear(dim, out_dim))

    def forward(self, img, mask=None):
        return self.mlp_head(self.to_latent(self.transformer(img, mask=None)))
