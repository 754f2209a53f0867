"""Executor for the two attention+MLP stacks of the reference, as explicit kernel sequences:

* dim-halving  ``vision_transformer.Transformer`` (models/vision_transformer.py:81-101):
  x += Attn(LN x); non-last: x = FF(LN x) (dim -> dim//2, no residual); last: x = FF3(x).
* dim-preserving ``vit.Transformer`` (models/vit.py:71-84): x = Attn(x)+x; x = FF(x)+x, no LN.

One autograd node for the whole stack.  Per layer: LayerNorm (wave-per-row), qkv GEMM (fp32
MFMA), LDS-resident attention core, out-projection GEMM with bias + residual fused in the
epilogue (C += A·B + b), FF GEMMs, exact-erf GELU.  ``input_grad`` replays the tape for the
input gradient only — that is the reference's pose-length term
``autograd.grad(sum(feat_out), feat_visual)`` (models/hand_net.py:396) without touching weights.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import List, Optional

import torch

from .. import _switches as _sw

from .. import ops


@dataclass
class LayerCfg:
    ln1: bool          # PreNorm in front of attention
    ff_ln: bool        # PreNorm in front of the MLP
    ff_res: bool       # residual around the MLP
    scale: float       # softmax scale
    heads: int
    dim_head: int
    post_ln: bool = False   # vision_transformer_attn: x = LN(Attn(x)) + x  (LayerNorm on the attention OUTPUT)

    def nparams(self):
        return (2 if self.ln1 else 0) + 3 + (2 if self.post_ln else 0) + (2 if self.ff_ln else 0) + 4


class Tape:
    __slots__ = ("cfgs", "params", "recs", "B", "n")


def mixer_forward(x: torch.Tensor, cfgs: List[LayerCfg], params: List[torch.Tensor], keep: bool):
    """x[B,n,dim] -> (y[B,n,dim_out], Tape|None)."""
    B, n, _ = x.shape
    M = B * n
    cur = x.contiguous().reshape(M, -1)
    recs = []
    pi = 0
    for cfg in cfgs:
        p = params[pi:pi + cfg.nparams()]
        pi += cfg.nparams()
        k = 0
        if cfg.ln1:
            g1, b1 = p[0], p[1]
            k = 2
            h, mu1, rs1 = ops.layernorm_fwd(cur, g1, b1)
        else:
            h, mu1, rs1 = cur, None, None
        wqkv, wout, bout = p[k], p[k + 1], p[k + 2]
        k += 3
        inner = cfg.heads * cfg.dim_head
        if h.is_cuda and ops.vit_fused_ok(n, h.shape[1], cfg.dim_head):
            # projection + attention in one launch (csrc/vit_fused.hip): the block's q, k, v stay in LDS
            qkv, ao, attn = ops.qkv_attention_fwd(h, wqkv, B, n, cfg.heads, cfg.scale)
        else:
            qkv = ops.linear_fwd(h, wqkv)
            ao, attn = ops.attention_fwd(qkv.view(B, n, 3 * inner), cfg.heads, cfg.dim_head, cfg.scale)
        if cfg.post_ln:
            gp, bp = p[k], p[k + 1]
            k += 2
            a1 = ops.linear_fwd(ao.view(M, inner), wout, bout)
            n1, mup, rsp = ops.layernorm_fwd(a1, gp, bp)
            x1 = ops.axpy(n1, cur, 1.0, out=n1)
        else:
            a1 = mup = rsp = None
            x1 = cur.clone()
            ops.linear_fwd(ao.view(M, inner), wout, bout, out=x1, accumulate=True)   # x + ao·Wᵀ + b in the epilogue
        if cfg.ff_ln:
            g2, b2 = p[k], p[k + 1]
            k += 2
            h2, mu2, rs2 = ops.layernorm_fwd(x1, g2, b2)
        else:
            h2, mu2, rs2 = x1, None, None
        w0, b0, w2, bb2 = p[k], p[k + 1], p[k + 2], p[k + 3]
        u = ops.linear_fwd(h2, w0, b0)
        a = ops.gelu_fwd(u)
        if cfg.ff_res:
            x2 = x1.clone()
            ops.linear_fwd(a, w2, bb2, out=x2, accumulate=True)
        else:
            x2 = ops.linear_fwd(a, w2, bb2)
        if keep:
            recs.append((cur, h, mu1, rs1, qkv, attn, ao, x1, h2, mu2, rs2, u, a, a1, mup, rsp))
        cur = x2
    y = cur.view(B, n, -1)
    tape = None
    if keep:
        tape = Tape()
        tape.cfgs, tape.params, tape.recs, tape.B, tape.n = cfgs, params, recs, B, n
    last_attn[0] = attn
    return y, tape


last_attn = [None]   # softmax probabilities [B,h,n,n] of the last layer of the most recent forward


def mixer_backward(tape: Tape, dy: torch.Tensor, want_param_grads: bool = True):
    """-> (dx[B,n,dim], [param grads in ``params`` order] or None)."""
    B, n = tape.B, tape.n
    M = B * n
    d = dy.contiguous().reshape(M, -1)
    grads: List[Optional[torch.Tensor]] = [None] * len(tape.params)
    pi = len(tape.params)
    # The parameter gradients (weight-gradient contractions, bias column sums) are not read by anything in this
    # backward: they run on the backbone's side stream next to the data-gradient chain, whose kernels (M = 2016
    # tokens) are far too small to fill the GPU on their own.  Joined before the node returns.
    main = side = None
    from .. import graphed
    if want_param_grads and d.is_cuda and _sw.ab("SCAT_SIDE_HEAD", True) and graphed.fork_ok():
        from . import resnet as _rn

        side = _rn._side_stream(d.device, "tokens")
        main = torch.cuda.current_stream()

    # the weight-gradient contractions (four per layer, independent of everything in this backward) are collected and
    # issued as ONE grouped launch at the end instead of twelve small ones next to the data-gradient chain
    group = [] if (want_param_grads and d.is_cuda and ops.GROUP_WGRAD) else None

    def wgrad(slot, dy2d, x2d):
        if group is None:
            grads[slot] = pgrad(lambda: ops.linear_wgrad(dy2d, x2d), dy2d, x2d)
        else:
            group.append((slot, dy2d, x2d))

    # ... and so are the bias gradients (three column sums per layer)
    bias_group = [] if group is not None else None

    def bgrad(slot, dy2d):
        if bias_group is None:
            grads[slot] = pgrad(lambda: ops.colsum(dy2d), dy2d)
        else:
            bias_group.append((slot, dy2d))

    def pgrad(fn, *inputs):
        if side is None:
            return fn()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out = fn()
        for t in inputs:
            t.record_stream(side)
        for o in (out if isinstance(out, (list, tuple)) else (out,)):
            o.record_stream(main)
        return out

    for cfg, rec in zip(reversed(tape.cfgs), reversed(tape.recs)):
        cur, h, mu1, rs1, qkv, attn, ao, x1, h2, mu2, rs2, u, a, a1, mup, rsp = rec
        npar = cfg.nparams()
        pi -= npar
        p = tape.params[pi:pi + npar]
        k = (2 if cfg.ln1 else 0)
        wqkv, wout = p[k], p[k + 1]
        kp = k + 3                               # post-LN params (if any)
        kn = kp + (2 if cfg.post_ln else 0)      # FF pre-norm params (if any)
        kf = kn + (2 if cfg.ff_ln else 0)
        w0, w2 = p[kf], p[kf + 2]
        inner = cfg.heads * cfg.dim_head
        # ---- MLP
        da = ops.linear_dgrad(d, w2)
        if want_param_grads:
            wgrad(pi + kf + 2, d, a)
            bgrad(pi + kf + 3, d)
        du = ops.gelu_bwd(da, u)
        if want_param_grads:
            wgrad(pi + kf, du, h2)
            bgrad(pi + kf + 1, du)
        dh2 = ops.linear_dgrad(du, w0)
        if cfg.ff_ln:
            dx1, dg2, db2 = ops.layernorm_bwd(dh2, x1, p[kn], mu2, rs2, want_param_grads)
            if want_param_grads:
                grads[pi + kn], grads[pi + kn + 1] = dg2, db2
        else:
            dx1 = dh2
        if cfg.ff_res:
            dx1 = ops.axpy(dx1, d, 1.0, out=dx1)
        # ---- attention: x1 = x + ao·Woutᵀ + b   (post_ln: x1 = x + LN(ao·Woutᵀ + b))
        if cfg.post_ln:
            da1, dgp, dbp = ops.layernorm_bwd(dx1, a1, p[kp], mup, rsp, want_param_grads)
            if want_param_grads:
                grads[pi + kp], grads[pi + kp + 1] = dgp, dbp
        else:
            da1 = dx1
        if want_param_grads:
            wgrad(pi + k + 1, da1, ao.view(M, inner))
            bgrad(pi + k + 2, da1)
        dao = ops.linear_dgrad(da1, wout)
        dqkv = ops.attention_bwd(dao.view(B, n, inner), qkv.view(B, n, 3 * inner), attn, cfg.heads, cfg.dim_head,
                                 cfg.scale).view(M, 3 * inner)
        if want_param_grads:
            wgrad(pi + k, dqkv, h)
        dh = ops.linear_dgrad(dqkv, wqkv)
        if cfg.ln1:
            dx, dg1, db1 = ops.layernorm_bwd(dh, cur, p[0], mu1, rs1, want_param_grads)
            if want_param_grads:
                grads[pi], grads[pi + 1] = dg1, db1
        else:
            dx = dh
        d = ops.axpy(dx, dx1, 1.0, out=dx)
    if group:
        pairs = [(dy2d, x2d) for _, dy2d, x2d in group]
        outs = pgrad(lambda: ops.linear_wgrad_group(pairs), *[t for pr in pairs for t in pr])
        for (slot, _, _), g in zip(group, outs):
            grads[slot] = g
    if bias_group:
        xs = [t for _, t in bias_group]
        outs = pgrad(lambda: ops.colsum_group(xs), *xs)
        for (slot, _), g in zip(bias_group, outs):
            grads[slot] = g
    if side is not None:
        main.wait_stream(side)
    return d.view(B, n, -1), (grads if want_param_grads else None)


class TapeHolder:
    """Mutable slot the autograd node fills so the owning module can replay the tape."""
    tape: Optional[Tape] = None
    want_tape: bool = False   # keep the tape even when nothing requires grad (pose-length term in eval mode)


class _MixerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, holder, cfgs, *params):
        keep = any(ctx.needs_input_grad) or holder.want_tape
        y, tape = mixer_forward(x, cfgs, list(params), keep)
        ctx.tape = tape
        holder.tape = tape
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.tape is None:
            raise RuntimeError("scat_amd: mixer backward without a recorded forward")
        dx, grads = mixer_backward(ctx.tape, dy, True)
        return (dx if ctx.needs_input_grad[0] else None, None, None, *grads)


def run_mixer(x, holder, cfgs, params):
    return _MixerFn.apply(x, holder, cfgs, *params)
