"""FAVOR+ performer block and the ViP patch model with the reference's interface
(models/vision_performer.py:12-116 of tomguluson92/SCAT): ``performer_attn_block(emb_s, head,
kernel_ratio, dp_ratio)`` with attributes ``kqv`` (ONE Linear(emb_s -> 3*emb_s) shared by all heads, :17),
``proj``, ``ln1``, ``ln2``, ``mlp``, frozen random features ``w`` (:32, in the state_dict, requires_grad False);
``ViP(opt, mean_params, image_pix, patch_pix, out_dim, emb_s, heads, depth, kernel_ratio, dropout)``.

The reference loops the heads in Python (:59-60); here all (batch, head) pairs run in one launch of the
linear-attention core (csrc/performer.hip) and the shared kqv projection is one GEMM over B*T*heads rows.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import nn as snn
from .. import ops


class _PerformerCoreFn(torch.autograd.Function):
    """prm_exp + linear attention for every head (vision_performer.py:34-53); split order k, q, v (:47)."""

    @staticmethod
    def forward(ctx, kqv, w):
        kqv, w = kqv.contiguous(), w.contiguous()
        y, saved = ops.performer_fwd(kqv, w, kqv.shape[2])
        ctx.save_for_backward(kqv, w, y, *saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        kqv, w, y, *saved = ctx.saved_tensors
        return ops.performer_bwd(dy.contiguous(), kqv, w, y, tuple(saved)), None


class performer_attn_block(nn.Module):
    def __init__(self, emb_s, head, kernel_ratio=0.5, dp_ratio=0.1):
        super().__init__()
        emb = emb_s * head
        self.kqv = snn.Linear(emb_s, 3 * emb_s)
        self.dp = snn.Dropout(dp_ratio)
        self.proj = snn.Linear(emb, emb)
        self.emb_s = emb_s
        self.ln1 = snn.LayerNorm(emb)
        self.ln2 = snn.LayerNorm(emb)
        self.mlp = nn.Sequential(snn.Linear(emb, 4 * emb), snn.GELU(), snn.Linear(4 * emb, emb), snn.Dropout(dp_ratio))
        self.m = int(emb_s * kernel_ratio)
        self.w = nn.Parameter(torch.randn(self.m, emb_s), requires_grad=False)

    def forward_multi_attn(self, x):
        B, T, emb = x.shape
        kqv = self.kqv(x.reshape(B, T, emb // self.emb_s, self.emb_s))        # [B,T,H,3e], one GEMM
        return self.dp(self.proj(_PerformerCoreFn.apply(kqv, self.w)))

    def forward(self, x):
        x = snn.add(x, self.forward_multi_attn(self.ln1(x)))
        return snn.add(x, self.mlp(self.ln2(x)))


class ViP(nn.Module):
    def __init__(self, opt, mean_params, image_pix=64, patch_pix=4, out_dim=10, emb_s=128, heads=4, depth=3,
                 kernel_ratio=0.5, dropout=0.1):
        super().__init__()
        tokens_cnt = (image_pix // patch_pix) * (image_pix // patch_pix)
        patch_size = 3 * patch_pix * patch_pix
        self.pool = "mean"
        self.patch_pix = patch_pix
        self.uf = nn.Identity()   # placeholder for nn.Unfold (no parameters); unfolding is done in forward
        emb = emb_s * heads
        self.pos_emb = nn.Parameter(torch.zeros(1, tokens_cnt, emb))
        self.dp = snn.Dropout(dropout)
        self.head = snn.Linear(emb + out_dim, out_dim)
        self.patch_emb = snn.Linear(patch_size, emb)
        self.cls_token = nn.Parameter(torch.rand(1, 1, emb))
        self.mains = nn.Sequential(*[performer_attn_block(emb_s=emb_s, head=heads, kernel_ratio=kernel_ratio,
                                                          dp_ratio=dropout) for _ in range(depth)])
        self.apply(self._init_weights)
        self.iteration = opt.iteration
        self.mean_params = mean_params.clone().cuda()

    def _init_weights(self, module):   # vision_performer.py:93-100
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if isinstance(module, nn.Linear) and module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def forward(self, x):
        from .hand_net import _HeadLoopFn

        b, c, H, W = x.shape
        p = self.patch_pix
        # nn.Unfold(p, stride p) + transpose: [B, (H/p)(W/p), c*p*p] with (c, kh, kw) fastest — a pure re-layout
        patches = x.reshape(b, c, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(b, -1, c * p * p)
        tok = snn.add(self.patch_emb(patches), self.pos_emb.expand(b, -1, -1))
        tok = torch.cat([self.cls_token.repeat(b, 1, 1), tok], dim=1)
        tok = self.mains(tok.contiguous())
        feat = snn.token_mean(tok)
        return _HeadLoopFn.apply(feat, self.mean_params.reshape(-1), self.head.weight, self.head.bias, self.iteration)
