"""Dim-halving transformer with the reference's module tree (models/vision_transformer.py:13-101):
``layers[l][0] = Residual(PreNorm(dim, Attention))``, ``layers[l][1] = PreNorm(dim, FeedForward)``
(last layer: bare ``FeedForward(dim, 3*dim//4, out_dim=3)``) — so ``state_dict`` keys match
(``layers.L.0.fn.norm.weight``, ``layers.L.0.fn.fn.to_qkv.weight``, ``layers.L.1.fn.net.0.weight`` …).
``Transformer.forward`` runs the whole stack as one fused autograd node (scat_amd/models/_mixer.py);
the small modules also work stand-alone on the HIP kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import nn as snn
from .. import ops
from ._mixer import LayerCfg, TapeHolder, mixer_backward, run_mixer

MIN_NUM_PATCHES = 16


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(x, **kwargs) + x


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = snn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(self.norm(x), **kwargs)


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, out_dim=None):
        super().__init__()
        self.net = nn.Sequential(snn.Linear(dim, hidden_dim), snn.GELU(),
                                 snn.Linear(hidden_dim, dim // 2 if out_dim is None else 3))

    def forward(self, x):
        return self.net(x)


class _AttnCoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, heads, dim_head, scale):
        qkv = qkv.contiguous()
        out, attn = ops.attention_fwd(qkv, heads, dim_head, scale)
        ctx.save_for_backward(qkv, attn)
        ctx.cfg = (heads, dim_head, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, attn = ctx.saved_tensors
        return ops.attention_bwd(dout.contiguous(), qkv, attn, *ctx.cfg), None, None, None


class Attention(nn.Module):
    scale_from_dim = False   # vit.Attention scales by dim**-0.5 instead (models/vit.py:41)

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.dim_head = dim_head
        self.scale = (dim if self.scale_from_dim else dim_head) ** -0.5
        self.to_qkv = snn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(snn.Linear(inner_dim, dim), snn.Dropout(dropout))

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("scat_amd: attention mask is never used on the reference path "
                                      "(models/hand_net.py:375 passes None)")
        return self.to_out(_AttnCoreFn.apply(self.to_qkv(x), self.heads, self.dim_head, self.scale))


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList([])
        self._cfgs = []
        for l in range(depth):
            last = l == depth - 1
            attn = Residual(PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout)))
            if last:
                ff = FeedForward(dim, (dim * 3) // 4, out_dim=3)
            else:
                ff = PreNorm(dim, FeedForward(dim, (dim * 3) // 4))
            self.layers.append(nn.ModuleList([attn, ff]))
            self._cfgs.append(LayerCfg(True, not last, False, dim_head ** -0.5, heads, dim_head))
            if not last:
                dim = dim // 2
        self._holder = TapeHolder()

    def _params(self):
        ps = []
        for (attn, ff), cfg in zip(self.layers, self._cfgs):
            pn = attn.fn
            ps += [pn.norm.weight, pn.norm.bias, pn.fn.to_qkv.weight, pn.fn.to_out[0].weight, pn.fn.to_out[0].bias]
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
        return run_mixer(x, self._holder, self._cfgs, self._params())

    def input_grad(self, dy):
        """d(sum(dy*out))/d(input) from the last forward's tape, weights untouched."""
        if self._holder.tape is None:
            raise RuntimeError("scat_amd: input_grad needs a forward that kept its tape")
        with torch.no_grad():
            return mixer_backward(self._holder.tape, dy, want_param_grads=False)[0]
