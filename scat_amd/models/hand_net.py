"""``EncoderTransformer`` — the reference's ``reg_transformer`` network (models/hand_net.py:315-398 of
tomguluson92/SCAT) with the same constructor ``(opt, mean_params)``, attribute names, ``state_dict``
keys (356 entries for ResNet-50) and return tuple ``(pred_params[B,66], feat_visual[B,21,28,28]
[, pl_term[B,21,28,28]])``, running on libscat_hip:

    ResNet-50 (one fused node) -> conv1x1 512->21 on x2 -> tokens (+PE, mask-token scatter)
    -> dim-halving transformer (one fused node) -> 63 offsets on the mean template
    -> ``iteration`` x Linear(1090->66) residual refinement -> root-relative joints.

Quirks kept on purpose: masking draws from python ``random`` exactly like the
reference (also in eval mode); ``mean_params`` is a plain attribute moved with ``.cuda()`` in the
constructor; ``pl_term`` carries no gradient.
"""
from __future__ import annotations

import os
import random
import warnings

import numpy as np
import torch

from .. import _switches as _sw
import torch.nn as nn

from .. import nn as snn
from .. import ops
from ..dp import auto_attach
from . import hrnet, resnet, vision_transformer, vision_transformer_attn, vit


def get_model(arch):
    if hasattr(resnet, arch):
        return getattr(resnet, arch)(pretrained=True, num_classes=512)
    raise ValueError("Invalid Backbone Architecture")


class PositionalEncoding(nn.Module):
    """Sinusoidal table as a ``pe`` buffer [1,max_len,d_model] (models/hand_net.py:61-77)."""

    def __init__(self, d_model, dropout=0.0, max_len=5000):
        super().__init__()
        self.dropout = snn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        # the reference slices dim 0 of the [1,L,D] buffer with the batch size => always the whole table
        return _TokensFn.apply(x, self.pe[0], None, None)


_ONES = {}


def _ones_like(t):
    """a cached all-ones tensor (the cotangent of ``sum(feat_out)``, hand_net.py:396): one fill per shape, not per step.
    The fill runs on whichever stream is current at first use; a reader on ANOTHER stream (a second model in the process,
    SCAT_OVERLAP_TOKENS=0) waits for the event recorded behind it.  Read-only by contract: it is only ever handed to
    Transformer.input_grad as ``dy`` (never an ``out=`` target)."""
    key = (tuple(t.shape), t.dtype, str(t.device))
    e = _ONES.get(key)
    if e is None:
        o = torch.ones_like(t)
        ev = st = None
        if o.is_cuda:
            st = torch.cuda.current_stream(o.device)
            ev = st.record_event()
        e = _ONES[key] = (o, ev, st)
    o, ev, st = e
    if ev is not None:
        cur = torch.cuda.current_stream(o.device)
        if cur != st:
            cur.wait_event(ev)       # (a completed event costs nothing on the device)
    return o


def _upload_indices(idx, device):
    """int32 index tensor on ``device`` without making the host wait: ``torch.tensor(list, device=cuda)`` copies from
    pageable memory, which synchronises the stream — in the middle of the step (after the backbone forward) that
    drains the launch queue and the many small kernels of the token path then run at host speed.  Staged through
    the pinned caching allocator instead (the block is recycled only after the copy has completed)."""
    h = torch.tensor(idx, dtype=torch.int32)
    if torch.device(device).type != "cuda":
        return h.to(device)
    return h.pin_memory().to(device, non_blocking=True)


OVERLAP_TOKENS = _sw.ab("SCAT_OVERLAP_TOKENS", True)   # token path next to layer3/layer4 (own stream)
TOKENS_FIRST_IN_BACKWARD = _sw.ab("SCAT_TOKENS_FIRST", True)   # (0: the round-2 node order, for A/B runs)
TOKEN_PRIO = _sw.ab("SCAT_TOKEN_PRIO", False)   # token stream with high priority (A/B switch)
# The token-path parameters get their gradients from nodes that ran on the token stream while their AccumulateGrad
# nodes belong to the caller's stream: autograd orders the two (that is the design) and says so once per process.
warnings.filterwarnings("ignore", message="The AccumulateGrad node's stream does not match")


def _token_stream(device):
    from .. import streams
    if TOKEN_PRIO and "tokens" not in streams.bound(device):
        streams.get(device, "wgrad")
        streams.bind(device, ["tokens"], priority=-1)
    return streams.get(device, "tokens")


def _backbone_with_tokens(backbone, main_input, token_path):
    """-> (main_feat, token_path(x2)).  The token path of the ResNet wrappers needs x2 only; layer3, layer4 and fc1
    need nothing of it.  Its kernels (2016 tokens at batch 96) are far too small to fill the GPU, so it runs on its
    own stream next to layer3/layer4 — and autograd runs each node's backward on its forward's stream, so the two
    backwards overlap the same way.  (Unattended data-parallel mode reduces the head bucket from parameter hooks on
    whatever stream they fire: it keeps the single-stream schedule.)"""
    sink = getattr(backbone, "_grad_sink", None)
    if not (OVERLAP_TOKENS and main_input.is_cuda) or (sink is not None and getattr(sink, "_auto", False)):
        main_feat, x1, x2, x3, x4 = backbone(main_input)
        return main_feat, token_path(x2)
    main = torch.cuda.current_stream()
    ts = _token_stream(main_input.device)
    x1, x2, x2b = backbone.first_half(main_input)      # (x2b: the same tensor as a second autograd output)
    ts.wait_stream(main)                               # (the token stream waits for the first half only)
    if TOKENS_FIRST_IN_BACKWARD:
        # Autograd runs ready nodes in the reverse of their creation order, one after the other on the host: created
        # BEFORE second_half, the token path's ~100 small backward kernels were only enqueued once the host had issued the
        # whole layer4 / layer3 backward, reached the GPU at its end and kept first_half's backward waiting (it needs their
        # gradient of x2) — 1-2 ms of an otherwise idle main queue in the kernel trace.  Created AFTER it, they are
        # enqueued first and run under layer4's backward.  The forward does not care: the host is far ahead there.
        main_feat, x3, x4 = backbone.second_half(x2b)
        with torch.cuda.stream(ts):
            outs = token_path(x2)
        x2.record_stream(ts)
    else:
        with torch.cuda.stream(ts):
            outs = token_path(x2)
        x2.record_stream(ts)
        main_feat, x3, x4 = backbone.second_half(x2b)
    main.wait_stream(ts)
    for t in outs:
        if isinstance(t, torch.Tensor):
            t.record_stream(main)
    return main_feat, outs


class _TokensFn(torch.autograd.Function):
    """x + pe, then rows ``masked`` <- mask_token (models/hand_net.py:366-373), one pass."""

    @staticmethod
    def forward(ctx, x, pe, mask_token, midx):
        x = x.contiguous()
        ctx.midx = midx
        ctx.has_mask = mask_token is not None and midx is not None and midx.numel() > 0
        mt = mask_token.reshape(-1) if ctx.has_mask else None
        return ops.tokens_fwd(x, pe, mt, midx if ctx.has_mask else None)

    @staticmethod
    def backward(ctx, dy):
        dx, dm = ops.tokens_bwd(dy.contiguous(), ctx.midx if ctx.has_mask else None, want_dmask=ctx.has_mask)
        return dx, None, (dm.view(1, 1, -1) if ctx.has_mask else None), None


class _RegressorFn(torch.autograd.Function):
    """mean template + offsets, ``iters`` residual Linear(1090->66) steps, root-relative joints
    (models/hand_net.py:379-393) as one kernel per direction."""

    @staticmethod
    def forward(ctx, feat, feat_out, mean, w, b, iters):
        feat, feat_out, w = feat.contiguous(), feat_out.contiguous(), w.contiguous()
        out, preds = ops.regressor_fwd(feat, feat_out, mean, w, b, iters)
        ctx.save_for_backward(feat, preds, w)
        ctx.iters = iters
        return out

    @staticmethod
    def backward(ctx, dout):
        feat, preds, w = ctx.saved_tensors
        dfeat, dfo, dw, db = ops.regressor_bwd(dout.contiguous(), feat, preds, w, ctx.iters)
        return dfeat, dfo, None, dw, db, None


class _HeadLoopFn(torch.autograd.Function):
    """``iteration`` x (pred += Linear([feat, pred])) from a constant start vector — the head of the HRNet
    wrapper (models/hand_net.py:206-211) and of ViP (models/vision_performer.py:112-115)."""

    @staticmethod
    def forward(ctx, feat, mean, w, b, iters):
        feat, w = feat.contiguous(), w.contiguous()
        out, preds = ops.regressor_fwd(feat, None, mean, w, b, iters, root_relative=False)
        ctx.save_for_backward(feat, preds, w)
        ctx.iters = iters
        return out

    @staticmethod
    def backward(ctx, dout):
        feat, preds, w = ctx.saved_tensors
        dfeat, _, dw, db = ops.regressor_bwd(dout.contiguous(), feat, preds, w, ctx.iters, root_relative=False,
                                             want_dfeat_out=False)
        return dfeat, None, dw, db, None


class H3DWEncoder(nn.Module):
    """models/hand_net.py:28-58 — the FrankMocap-style regressor ``eval.py:26`` imports: ResNet-50 feature →
    ``relu, Linear(1024,1024), relu`` → 3 x ``Linear(1085 -> 61)`` residual steps from the mean parameters;
    returns ``(feat, pred_params)``.  Like the reference it only runs at batch 1 (``torch.cat`` of ``feat[B,1024]``
    with the un-repeated ``mean_params[1,61]``, hand_net.py:52-54); larger batches raise the same way."""

    def __init__(self, opt, mean_params):
        super().__init__()
        self.mean_params = mean_params.clone().cuda()
        self.total_params_dim = 61
        self.feat_encoder = nn.Sequential(snn.ReLU(inplace=False), snn.Linear(1024, 1024), snn.ReLU(inplace=False))
        self.regressor = nn.Sequential(snn.Linear(1024 + self.total_params_dim, self.total_params_dim))
        self.main_encoder = get_model("resnet50")

    def forward(self, main_input):
        main_feat, _, _, _, _ = self.main_encoder(main_input)
        feat = self.feat_encoder(main_feat)
        if feat.size(0) != self.mean_params.size(0):
            raise RuntimeError(
                f"Sizes of tensors must match except in dimension 1. Expected size {feat.size(0)} but got size "
                f"{self.mean_params.size(0)} for tensor number 1 in the list. (H3DWEncoder concatenates feat with "
                "the un-repeated mean_params, models/hand_net.py:52-54: batch 1 only)")
        lin = self.regressor[0]
        pred_params = _HeadLoopFn.apply(feat, self.mean_params.reshape(-1), lin.weight, lin.bias, 3)
        return feat, pred_params


class EncoderTransformerInception(nn.Module):
    """models/hand_net.py:87-146 — importable (``train_coarse.py:7`` imports the name) but not constructible: the
    reference class is broken as shipped (``vision_transformer.Transformer`` halves the token width while the
    regressor is sized for a width-preserving one: ``mat1 and mat2 shapes cannot be multiplied``), no
    BASELINE config names it, and its Inception-v3 trunk (models/inception.py) is outside the hot path
    (OUT OF SCOPE)."""

    def __init__(self, opt, mean_params):
        super().__init__()
        raise NotImplementedError(
            "EncoderTransformerInception (models/hand_net.py:87-146) is broken as shipped in the reference "
            "(dim-halving transformer vs a Linear(196+61, 61) regressor) and its Inception-v3 trunk is out of "
            "scope of the MI355X hot path; use EncoderTransformer / EncoderTransformerCoarse / "
            "EncoderTransformerHRNet.")


class EncoderTransformerHRNet(nn.Module):
    """models/hand_net.py:150-213.  As shipped the reference pairs this wrapper with the dim-HALVING
    ``vision_transformer.Transformer`` and a regressor sized for a dim-preserving one, which cannot run
    (``2x64 and 257x61``); the working wiring — and the one BASELINE config 4 names — is the
    dim-preserving ``models.vit.Transformer`` (scale dim**-0.5, no LayerNorm), used here.  ``opt.hrnet_width``
    (default 24 like the reference; 32 = HRNet-W32) selects the branch width."""

    def __init__(self, opt, mean_params):
        super().__init__()
        self.mean_params = mean_params.clone().cuda()
        self.total_params_dim = 61
        self.main_encoder = hrnet.HRNet(c=getattr(opt, "hrnet_width", 24), nof_joints=128, bn_momentum=0.1)
        self.conv1x1_channel_reduction = snn.Conv2d(512, 128, 3, 2, 1, bias=False)
        self.transformer = vit.Transformer(196, opt.vit_depth, opt.vit_heads, 64, 392, 0.0)
        self.iteration = opt.iteration
        self.regressor = nn.Sequential(snn.Linear(196 + self.total_params_dim, self.total_params_dim))
        self.pos_embed = opt.pos_embed
        self.positionalEncoding = PositionalEncoding(196, max_len=128)
        self.mask_token = nn.Parameter(torch.randn(1, 1, 196))
        self.mask_rate = opt.mask_rate
        self._midx_cache = {}

    def forward(self, main_input):
        auto_attach(self)
        main_feat = self.main_encoder(main_input)                               # [B,128,56,56]
        B = main_feat.size(0)
        feat = self.conv1x1_channel_reduction(main_feat.view(B, 512, 28, 28))   # legal only at 224x224
        midx = None
        if 0.1 <= self.mask_rate <= 0.9:
            masked = list(range(128))
            random.shuffle(masked)
            masked = masked[: int(self.mask_rate * 128)]
            midx = _upload_indices(masked, feat.device) if masked else None
        pe = self.positionalEncoding.pe[0] if self.pos_embed else None
        tokens = _TokensFn.apply(feat.view(B, 128, -1), pe, self.mask_token, midx)
        feat = snn.token_mean(self.transformer(tokens, None))                   # [B,196]
        lin = self.regressor[0]
        return _HeadLoopFn.apply(feat, self.mean_params.reshape(-1), lin.weight, lin.bias, self.iteration)


class EncoderTransformerCoarse(nn.Module):
    """models/hand_net.py:216-311 — the ``train_coarse.py`` network: same backbone and token path, the
    attention-returning transformer, a single ``Linear(1027 -> 3)`` camera regressor instead of the iterative
    loop, and a 3- or 4-tuple ``(pred_params, feat_visual, attn[B,8,21,21][, pl_term])``."""

    def __init__(self, opt, mean_params):
        super().__init__()
        self.mean_params = mean_params.clone().cuda()
        self.pl = opt.pl_reg
        self.full_content = 21
        self.conv1x1_channel_reduction = snn.Conv2d(512, 21, 1, 1, 0, bias=False)
        self.transformer = vision_transformer_attn.Transformer(dim=784, depth=3, heads=8, dim_head=64, mlp_dim=392,
                                                               dropout=0.0)
        self.main_encoder = get_model("resnet50")
        self.iteration = opt.iteration
        self.pos_embed = opt.pos_embed
        self.positionalEncoding = PositionalEncoding(784, max_len=21)
        self.mask_token = nn.Parameter(torch.randn(1, 1, 784))
        self.mask_rate = opt.mask_rate
        self.regressor = snn.Linear(1024 + 3, 3)
        self._midx_cache = {}

    _draw_mask = None   # bound below (same python-random draw as EncoderTransformer)

    def forward(self, main_input):
        auto_attach(self)

        def token_path(x2):
            feat_visual = self.conv1x1_channel_reduction(x2)
            B = feat_visual.size(0)
            midx = self._draw_mask(feat_visual.device)
            pe = self.positionalEncoding.pe[0] if self.pos_embed else None
            tokens = _TokensFn.apply(feat_visual.view(B, 21, -1), pe, self.mask_token, midx)
            self.transformer._holder.want_tape = bool(self.pl)
            feat_out, attn = self.transformer(tokens, None)
            aliased = pe is None and midx is not None     # hand_net.py:273-284, see EncoderTransformer._token_path
            if aliased:
                feat_visual = tokens.view_as(feat_visual)
            pl_term = None
            if self.pl:
                dtok = self.transformer.input_grad(_ones_like(feat_out))
                if aliased:
                    pl_term = dtok.contiguous().view_as(feat_visual)
                else:
                    pl_term = ops.tokens_bwd(dtok.contiguous(), midx, want_dmask=False)[0].view_as(feat_visual)
            return feat_visual, feat_out, attn, pl_term

        main_feat, (feat_visual, feat_out, attn, pl_term) = _backbone_with_tokens(self.main_encoder, main_input,
                                                                                  token_path)
        B = feat_visual.size(0)
        mean = self.mean_params.reshape(-1)
        # joints: mean template + offsets, root-relative (no refinement loop on this variant: iters = 0)
        joints = _RegressorFn.apply(main_feat, feat_out.reshape(B, -1), mean, self._dummy_w(), self._dummy_b(), 0)
        cameras = self.regressor(torch.cat((main_feat, mean[:3].expand(B, 3)), dim=1))   # hand_net.py:296
        pred_params = torch.cat((cameras, joints[:, 3:]), dim=1)
        if self.pl:
            return pred_params, feat_visual, attn, pl_term
        return pred_params, feat_visual, attn

    def _dummy_w(self):
        if not hasattr(self, "_zw") or self._zw.device != self.mask_token.device:
            self._zw = torch.zeros((66, 1024 + 66), device=self.mask_token.device)
            self._zb = torch.zeros((66,), device=self.mask_token.device)
        return self._zw

    def _dummy_b(self):
        self._dummy_w()
        return self._zb


class EncoderTransformer(nn.Module):
    def __init__(self, opt, mean_params):
        super().__init__()
        self.mean_params = mean_params.clone().cuda()
        heads = opt.vit_heads
        self.pl = opt.pl_reg
        self.full_content = 21
        self.conv1x1_channel_reduction = snn.Conv2d(512, 21, 1, 1, 0, bias=False)
        self.transformer = vision_transformer.Transformer(dim=784, depth=3, heads=heads, dim_head=64, mlp_dim=392,
                                                          dropout=0.0)
        self.main_encoder = get_model("resnet50")
        self.iteration = opt.iteration
        self.pos_embed = opt.pos_embed
        self.positionalEncoding = PositionalEncoding(784, max_len=21)
        self.mask_token = nn.Parameter(torch.randn(1, 1, 784))
        self.mask_rate = opt.mask_rate
        self.regressor = snn.Linear(1024 + 66, 66)
        self._midx_cache = {}

    def _draw_mask(self, device):
        """python-``random`` draw sequence of models/hand_net.py:369-372."""
        if not (0.1 <= self.mask_rate <= 0.9):
            return None
        masked = list(range(self.full_content))
        random.shuffle(masked)
        masked = tuple(masked[: int(self.mask_rate * self.full_content)])
        if not masked:
            return None
        key = (masked, str(device))
        t = self._midx_cache.get(key)
        if t is None:
            if len(self._midx_cache) > 4096:
                self._midx_cache.clear()
            t = _upload_indices(masked, device)
            self._midx_cache[key] = t
        return t

    def _token_path(self, x2):
        """everything that depends on x2 only (hand_net.py:362-378, 395-396): 1x1 reduction, +PE, mask-token scatter,
        the dim-halving transformer and, with pl_reg, the pose-length term"""
        feat_visual = self.conv1x1_channel_reduction(x2)                      # [B,21,28,28]
        B = feat_visual.size(0)
        midx = self._draw_mask(feat_visual.device)
        pe = self.positionalEncoding.pe[0] if self.pos_embed else None
        tokens = _TokensFn.apply(feat_visual.view(B, 21, -1), pe, self.mask_token, midx)
        self.transformer._holder.want_tape = bool(self.pl)
        feat_out = self.transformer(tokens, None)                             # [B,21,3]
        # hand_net.py:364-373: without the positional table ``feat`` is a VIEW of ``feat_visual``, so the reference's
        # in-place mask-token write lands in the tensor it returns, and ``autograd.grad(.., feat_visual)`` is then taken
        # at the post-scatter tensor (non-zero in the masked channels).  Unreachable from the CLI (``type=bool`` makes
        # ``--pos_embed False`` truthy, config.py:44) but kept: golden ``encoder_nope.npz``.
        aliased = pe is None and midx is not None
        if aliased:
            feat_visual = tokens.view_as(feat_visual)
        pl_term = None
        if self.pl:
            # d sum(feat_out) / d feat_visual, no graph (hand_net.py:396): replay the mixer tape for the
            # input gradient only, then undo the token scatter.
            dtok = self.transformer.input_grad(_ones_like(feat_out))
            if aliased:
                pl_term = dtok.contiguous().view_as(feat_visual)
            else:
                pl_term = ops.tokens_bwd(dtok.contiguous(), midx, want_dmask=False)[0].view_as(feat_visual)
        return feat_visual, feat_out, pl_term

    def forward(self, main_input):
        auto_attach(self)   # WORLD_SIZE > 1: data-parallel gradient averaging without touching train.py
        main_feat, (feat_visual, feat_out, pl_term) = _backbone_with_tokens(self.main_encoder, main_input,
                                                                            self._token_path)
        B = feat_visual.size(0)
        pred_params = _RegressorFn.apply(main_feat, feat_out.reshape(B, -1), self.mean_params.reshape(-1),
                                         self.regressor.weight, self.regressor.bias, self.iteration)
        if self.pl:
            return pred_params, feat_visual, pl_term
        return pred_params, feat_visual


EncoderTransformerCoarse._draw_mask = EncoderTransformer._draw_mask


class EncoderPerformer(nn.Module):
    """BASELINE config 5: the SCAT token path with FAVOR+ linear attention.  The reference ships only the block
    (``models.vision_performer.performer_attn_block``) and a patch model (``ViP``); it never wires the block into
    ``hand_net``.  Wiring chosen here (documented in DESIGN.md): everything of
    ``EncoderTransformer`` up to the tokens [B,21,784] (ResNet-50, 1x1 reduction of x2, +PE, mask-token scatter),
    then ``depth`` x ``performer_attn_block(emb_s = 784 // heads, head = heads)`` (dim-preserving, dropout as in the
    block), a per-token ``Linear(784 -> 3)`` giving the 63 joint offsets, and the same iterative
    ``Linear(1090 -> 66)`` regressor.  ``opt.vit_heads`` = 16 -> emb_s = 49, m = 24 random features per head."""

    def __init__(self, opt, mean_params, depth=3):
        super().__init__()
        from . import vision_performer

        self.mean_params = mean_params.clone().cuda()
        heads = opt.vit_heads
        assert 784 % heads == 0, "784 token features must split evenly over the heads (16 -> emb_s 49)"
        self.full_content = 21
        self.conv1x1_channel_reduction = snn.Conv2d(512, 21, 1, 1, 0, bias=False)
        self.blocks = nn.ModuleList([vision_performer.performer_attn_block(784 // heads, heads) for _ in range(depth)])
        self.to_offsets = snn.Linear(784, 3)
        self.main_encoder = get_model("resnet50")
        self.iteration = opt.iteration
        self.pos_embed = opt.pos_embed
        self.positionalEncoding = PositionalEncoding(784, max_len=21)
        self.mask_token = nn.Parameter(torch.randn(1, 1, 784))
        self.mask_rate = opt.mask_rate
        self.regressor = snn.Linear(1024 + 66, 66)
        self._midx_cache = {}

    _draw_mask = EncoderTransformer._draw_mask

    def forward(self, main_input):
        auto_attach(self)

        def token_path(x2):
            feat_visual = self.conv1x1_channel_reduction(x2)
            B = feat_visual.size(0)
            midx = self._draw_mask(feat_visual.device)
            pe = self.positionalEncoding.pe[0] if self.pos_embed else None
            tok = _TokensFn.apply(feat_visual.view(B, 21, -1), pe, self.mask_token, midx)
            for blk in self.blocks:
                tok = blk(tok)
            return feat_visual, self.to_offsets(tok)                         # [B,21,28,28], [B,21,3]

        main_feat, (feat_visual, feat_out) = _backbone_with_tokens(self.main_encoder, main_input, token_path)
        B = feat_visual.size(0)
        pred_params = _RegressorFn.apply(main_feat, feat_out.reshape(B, -1), self.mean_params.reshape(-1),
                                         self.regressor.weight, self.regressor.bias, self.iteration)
        return pred_params, feat_visual
