"""CPU oracle for the SCAT reg_transformer hot path — TEST INFRASTRUCTURE ONLY.

A functional (state_dict-keyed) PyTorch-CPU fp32 restatement of the reference's
arithmetic. Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this; the product (``scat_amd``) never does.

Pinned: ``oracle/gen_golden.py`` runs the real reference modules (imported from
/root/reference in the build container) on tensors from ``scat_amd.synth`` and
stores their outputs in ``tests/golden``; ``tests/test_oracle_golden.py`` checks
this restatement against those files.

Every function cites the reference lines it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
import random
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOM = 0.1
LN_EPS = 1e-5


# --------------------------------------------------------------------------
# backbone — models/resnet.py
# --------------------------------------------------------------------------

def batch_norm(sd, key, x, training):
    """nn.BatchNorm2d as used at models/resnet.py:68-73,108 (momentum .1, eps 1e-5).
    Train mode normalises with batch stats and updates running stats in ``sd`` in place."""
    rm, rv = sd[key + ".running_mean"], sd[key + ".running_var"]
    y = F.batch_norm(x, rm, rv, sd[key + ".weight"], sd[key + ".bias"], training, BN_MOM, BN_EPS)
    if training and (key + ".num_batches_tracked") in sd:
        sd[key + ".num_batches_tracked"] += 1
    return y


def bottleneck(sd, key, x, stride, training):
    """Bottleneck.forward, models/resnet.py:78-98 (stride on the 3x3, :69)."""
    out = F.conv2d(x, sd[key + ".conv1.weight"])
    out = F.relu(batch_norm(sd, key + ".bn1", out, training))
    out = F.conv2d(out, sd[key + ".conv2.weight"], stride=stride, padding=1)
    out = F.relu(batch_norm(sd, key + ".bn2", out, training))
    out = F.conv2d(out, sd[key + ".conv3.weight"])
    out = batch_norm(sd, key + ".bn3", out, training)
    if (key + ".downsample.0.weight") in sd:
        res = F.conv2d(x, sd[key + ".downsample.0.weight"], stride=stride)
        res = batch_norm(sd, key + ".downsample.1", res, training)
    else:
        res = x
    return F.relu(out + res)


def resnet_forward(sd, x, prefix="", training=True):
    """ResNet.forward, models/resnet.py:142-162 → (feat[B,1024], x1, x2, x3, x4)."""
    p = prefix
    x = F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3)
    x = F.relu(batch_norm(sd, p + "bn1", x, training))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li in range(4):
        bi = 0
        while f"{p}layer{li + 1}.{bi}.conv1.weight" in sd:
            stride = 2 if (li > 0 and bi == 0) else 1
            x = bottleneck(sd, f"{p}layer{li + 1}.{bi}", x, stride, training)
            bi += 1
        feats.append(x)
    x = F.avg_pool2d(feats[3], 7, 1)
    x = F.relu(x.reshape(x.size(0), -1))
    x = F.relu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
    return (x, *feats)


# --------------------------------------------------------------------------
# HRNet backbone — models/hrnet.py
# --------------------------------------------------------------------------

def _cbn(sd, kc, kb, x, training, stride=1, pad=1, relu=False):
    y = batch_norm(sd, kb, F.conv2d(x, sd[kc + ".weight"], stride=stride, padding=pad), training)
    return F.relu(y) if relu else y


def hr_basic_block(sd, key, x, training):
    """BasicBlock.forward, models/hrnet.py:61-77."""
    out = _cbn(sd, key + ".conv1", key + ".bn1", x, training, relu=True)
    out = _cbn(sd, key + ".conv2", key + ".bn2", out, training)
    return F.relu(out + x)


def hr_stage_module(sd, key, xs, stage, out_branches, training):
    """StageModule.forward, models/hrnet.py:128-144 (fuse layers built at :99-124)."""
    xs = list(xs)
    for i in range(stage):
        for b in range(4):
            xs[i] = hr_basic_block(sd, f"{key}.branches.{i}.{b}", xs[i], training)
    fused = []
    for i in range(out_branches):
        acc = None
        for j in range(stage):
            fk = f"{key}.fuse_layers.{i}.{j}"
            if i == j:
                t = xs[j]
            elif i < j:   # 1x1 conv + BN + nearest upsample by 2**(j-i)
                t = _cbn(sd, fk + ".0", fk + ".1", xs[j], training, pad=0)
                t = F.interpolate(t, scale_factor=float(2 ** (j - i)), mode="nearest")
            else:         # chain of stride-2 3x3 convs, ReLU on all but the last
                t = xs[j]
                for k in range(i - j):
                    t = _cbn(sd, f"{fk}.{k}.0", f"{fk}.{k}.1", t, training, stride=2, relu=(k < i - j - 1))
            acc = t if acc is None else acc + t
        fused.append(F.relu(acc))
    return fused


def hrnet_forward(sd, x, prefix="", training=True):
    """HRNet.forward, models/hrnet.py:230-261 -> [B,nof_joints,56,56]."""
    p = prefix
    x = _cbn(sd, p + "conv1", p + "bn1", x, training, stride=2, relu=True)
    x = _cbn(sd, p + "conv2", p + "bn2", x, training, stride=2, relu=True)
    for b in range(4):   # layer1: Bottlenecks (hrnet.py:10-45), first with a 1x1 downsample
        k = f"{p}layer1.{b}"
        out = _cbn(sd, k + ".conv1", k + ".bn1", x, training, pad=0, relu=True)
        out = _cbn(sd, k + ".conv2", k + ".bn2", out, training, relu=True)
        out = _cbn(sd, k + ".conv3", k + ".bn3", out, training, pad=0)
        res = _cbn(sd, k + ".downsample.0", k + ".downsample.1", x, training, pad=0) \
            if (k + ".downsample.0.weight") in sd else x
        x = F.relu(out + res)
    xs = [_cbn(sd, p + "transition1.0.0", p + "transition1.0.1", x, training, relu=True),
          _cbn(sd, p + "transition1.1.0.0", p + "transition1.1.0.1", x, training, stride=2, relu=True)]
    xs = hr_stage_module(sd, p + "stage2.0", xs, 2, 2, training)
    xs = [xs[0], xs[1], _cbn(sd, p + "transition2.2.0.0", p + "transition2.2.0.1", xs[-1], training, stride=2,
                             relu=True)]
    for m in range(4):
        xs = hr_stage_module(sd, f"{p}stage3.{m}", xs, 3, 3, training)
    xs = [xs[0], xs[1], xs[2], _cbn(sd, p + "transition3.3.0.0", p + "transition3.3.0.1", xs[-1], training,
                                    stride=2, relu=True)]
    xs = hr_stage_module(sd, p + "stage4.0", xs, 4, 4, training)
    xs = hr_stage_module(sd, p + "stage4.1", xs, 4, 4, training)
    xs = hr_stage_module(sd, p + "stage4.2", xs, 4, 1, training)
    return F.conv2d(xs[0], sd[p + "final_layer.weight"], sd[p + "final_layer.bias"])


def encoder_transformer_hrnet_forward(sd, mean_params, x, heads=8, depth=3, iteration=3, pos_embed=True,
                                      mask_rate=0.2, training=True, masked=None):
    """EncoderTransformerHRNet.forward, models/hand_net.py:176-213, with the dim-preserving
    models/vit.py transformer (the only wiring in which the wrapper's shapes agree)."""
    f = hrnet_forward(sd, x, "main_encoder.", training)
    b = f.size(0)
    feat = F.conv2d(f.reshape(b, 512, 28, 28), sd["conv1x1_channel_reduction.weight"], stride=2, padding=1)
    feat = feat.reshape(b, 128, -1)
    if pos_embed:
        feat = feat + sd["positionalEncoding.pe"][: feat.size(0)]
    if masked is None:
        masked = mask_indices(mask_rate, 128)
    if len(masked):
        feat = feat.clone()
        feat[:, masked, :] = sd["mask_token"]
    feat = vit_forward(sd, feat, "transformer.", depth, heads).mean(dim=1)
    pred = mean_params.repeat(b, 1)
    for _ in range(iteration):
        pred = pred + F.linear(torch.cat([feat, pred], dim=-1), sd["regressor.0.weight"], sd["regressor.0.bias"])
    return pred


# --------------------------------------------------------------------------
# token mixers
# --------------------------------------------------------------------------

def attention(x, w_qkv, w_out, b_out, heads, scale):
    """Attention.forward, models/vision_transformer.py:59-79 (mask is always None,
    models/hand_net.py:375). Returns (out, attn[B,h,n,n])."""
    b, n, _ = x.shape
    qkv = F.linear(x, w_qkv)
    inner = qkv.shape[-1] // 3
    d = inner // heads
    q, k, v = (t.reshape(b, n, heads, d).permute(0, 2, 1, 3) for t in qkv.split(inner, dim=-1))
    dots = torch.matmul(q, k.transpose(-1, -2)) * scale
    attn = dots.softmax(dim=-1)
    out = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(b, n, inner)
    return F.linear(out, w_out, b_out), attn


def vt_forward(sd, x, prefix, depth=3, heads=8, dim_head=64):
    """vision_transformer.Transformer.forward, models/vision_transformer.py:81-101:
    x += Attn(LN x); non-last: x = FF(LN x) (dim halves); last: x = FF_last(x) (no LN)."""
    scale = dim_head ** -0.5
    for l in range(depth):
        k = f"{prefix}layers.{l}"
        dim = x.shape[-1]
        h = F.layer_norm(x, (dim,), sd[k + ".0.fn.norm.weight"], sd[k + ".0.fn.norm.bias"], LN_EPS)
        a, _ = attention(h, sd[k + ".0.fn.fn.to_qkv.weight"], sd[k + ".0.fn.fn.to_out.0.weight"],
                         sd[k + ".0.fn.fn.to_out.0.bias"], heads, scale)
        x = a + x
        if l == depth - 1:
            f = "%s.1.net" % k
            h = x
        else:
            f = "%s.1.fn.net" % k
            h = F.layer_norm(x, (dim,), sd[k + ".1.norm.weight"], sd[k + ".1.norm.bias"], LN_EPS)
        h = F.gelu(F.linear(h, sd[f + ".0.weight"], sd[f + ".0.bias"]))
        x = F.linear(h, sd[f + ".2.weight"], sd[f + ".2.bias"])
    return x


def vt_attn_forward(sd, x, prefix, depth=3, heads=8, dim_head=64):
    """vision_transformer_attn.Transformer.forward, models/vision_transformer_attn.py:106-113:
    x1, attn = Attention(x); x = LN(x1) + x; x = FF(LN x) | FF3(x); returns (x, last attn)."""
    scale = dim_head ** -0.5
    attn = None
    for l in range(depth):
        k = f"{prefix}layers.{l}"
        dim = x.shape[-1]
        a, attn = attention(x, sd[k + ".0.to_qkv.weight"], sd[k + ".0.to_out.0.weight"], sd[k + ".0.to_out.0.bias"],
                            heads, scale)
        x = F.layer_norm(a, (dim,), sd[k + ".1.norm.weight"], sd[k + ".1.norm.bias"], LN_EPS) + x
        if l == depth - 1:
            f, h = k + ".2.net", x
        else:
            f = k + ".2.fn.net"
            h = F.layer_norm(x, (dim,), sd[k + ".2.norm.weight"], sd[k + ".2.norm.bias"], LN_EPS)
        h = F.gelu(F.linear(h, sd[f + ".0.weight"], sd[f + ".0.bias"]))
        x = F.linear(h, sd[f + ".2.weight"], sd[f + ".2.bias"])
    return x, attn


def vit_forward(sd, x, prefix, depth=3, heads=8):
    """vit.Transformer.forward, models/vit.py:71-84: x = Attn(x)+x; x = FF(x)+x;
    no LayerNorm; scale = dim**-0.5 (models/vit.py:41). Dropout p=0."""
    scale = x.shape[-1] ** -0.5
    for l in range(depth):
        k = f"{prefix}layers.{l}"
        a, _ = attention(x, sd[k + ".0.fn.to_qkv.weight"], sd[k + ".0.fn.to_out.0.weight"],
                         sd[k + ".0.fn.to_out.0.bias"], heads, scale)
        x = a + x
        h = F.gelu(F.linear(x, sd[k + ".1.fn.net.0.weight"], sd[k + ".1.fn.net.0.bias"]))
        x = F.linear(h, sd[k + ".1.fn.net.3.weight"], sd[k + ".1.fn.net.3.bias"]) + x
    return x


def hash_dropout_mask(n, p, seed):
    """The keep mask of the library's Dropout (scat_amd/csrc/misc.hip dropout_kernel, include/scat_hip.h scat_dropout):
    element e is kept iff u(e) >= p with u the top 24 bits of a splitmix64-style hash of e * 0xD1342543DE82EF95 + seed.
    The reference's nn.Dropout (models/vision_performer.py:18,28) draws from torch's RNG stream, which no device kernel
    can replay; this restatement makes the TRAIN-mode values of the performer path checkable against an oracle that uses
    the same Bernoulli(1 - p) mask, element for element.  -> float64 tensor of 0 / 1."""
    import numpy as np

    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + np.uint64(seed)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return torch.from_numpy((u >= np.float32(p)).astype(np.float64))


def performer_block(sd, x, prefix, emb_s, head, dropout_p=0.0):
    """performer_attn_block.forward, models/vision_performer.py:34-68. Split order k, q, v (:47).
    dropout_p = 0: eval mode.  dropout_p > 0: train mode (:18 after proj, :28 at the end of the mlp) with the library's
    hash mask: one 63-bit seed per Dropout call from python ``random``, in call order, as scat_amd.nn.Dropout draws them."""
    def drop(t):
        if dropout_p <= 0:
            return t
        keep = hash_dropout_mask(t.numel(), dropout_p, random.getrandbits(63)).reshape(t.shape).to(t.dtype)
        scale = torch.tensor(1.0, dtype=torch.float32) / (torch.tensor(1.0, dtype=torch.float32) - torch.tensor(dropout_p, dtype=torch.float32))
        return t * keep * scale.to(t.dtype)

    emb = emb_s * head
    w = sd[prefix + "w"]
    m = w.shape[0]

    def prm_exp(t):  # :34-43
        xd = (t * t).sum(dim=-1, keepdim=True) / 2
        return torch.exp(t @ w.t() - xd) / math.sqrt(m)

    h = F.layer_norm(x, (emb,), sd[prefix + "ln1.weight"], sd[prefix + "ln1.bias"], LN_EPS)
    outs = []
    for t in h.split(emb_s, dim=-1):  # :59-60
        k, q, v = F.linear(t, sd[prefix + "kqv.weight"], sd[prefix + "kqv.bias"]).split(emb_s, dim=-1)
        kp, qp = prm_exp(k), prm_exp(q)
        D = (qp * kp.sum(dim=1, keepdim=True)).sum(dim=-1, keepdim=True)  # :49
        kptv = torch.einsum("bin,bim->bnm", v, kp)  # :50
        outs.append(torch.einsum("bti,bni->btn", qp, kptv) / D)  # :52
    x = x + drop(F.linear(torch.cat(outs, dim=-1), sd[prefix + "proj.weight"], sd[prefix + "proj.bias"]))
    h = F.layer_norm(x, (emb,), sd[prefix + "ln2.weight"], sd[prefix + "ln2.bias"], LN_EPS)
    h = F.gelu(F.linear(h, sd[prefix + "mlp.0.weight"], sd[prefix + "mlp.0.bias"]))
    return x + drop(F.linear(h, sd[prefix + "mlp.2.weight"], sd[prefix + "mlp.2.bias"]))


def vip_forward(sd, mean_params, x, heads, emb_s, depth=3, iteration=3, patch=4):
    """ViP.forward in eval mode, models/vision_performer.py:102-116."""
    b = x.shape[0]
    t = F.unfold(x, kernel_size=patch, stride=patch).transpose(1, 2)                       # :104
    t = F.linear(t, sd["patch_emb.weight"], sd["patch_emb.bias"]) + sd["pos_emb"]
    t = torch.cat([sd["cls_token"].repeat(b, 1, 1), t], dim=1)
    for l in range(depth):
        t = performer_block(sd, t, f"mains.{l}.", emb_s, heads)
    feat = t.mean(dim=1)
    pred = mean_params.repeat(b, 1)
    for _ in range(iteration):
        pred = pred + F.linear(torch.cat([feat, pred], dim=1), sd["head.weight"], sd["head.bias"])
    return pred


# --------------------------------------------------------------------------
# model — models/hand_net.py:315-398
# --------------------------------------------------------------------------

def mask_indices(mask_rate, full_content=21):
    """models/hand_net.py:369-372 — consumes python ``random`` exactly like the reference."""
    if 0.1 <= mask_rate <= 0.9:
        masked = list(range(full_content))
        random.shuffle(masked)
        return masked[: int(mask_rate * full_content)]
    return []


def encoder_transformer_forward(sd, mean_params, x, heads=8, iteration=3, pos_embed=True,
                                mask_rate=0.2, pl_reg=True, training=True, masked=None):
    """EncoderTransformer.forward, models/hand_net.py:355-398."""
    feat1024, x1, x2, x3, x4 = resnet_forward(sd, x, "main_encoder.", training)
    feat_visual = F.conv2d(x2, sd["conv1x1_channel_reduction.weight"])  # :363
    b = feat_visual.size(0)
    feat = feat_visual.reshape(b, 21, -1)
    if pos_embed:
        feat = feat + sd["positionalEncoding.pe"][: feat.size(0)]  # :74-75 (slices dim 0 of [1,21,784])
    if masked is None:
        masked = mask_indices(mask_rate)
    if len(masked):
        feat = feat.clone()
        feat[:, masked, :] = sd["mask_token"]  # :373
        if not pos_embed:
            # :364 makes ``feat`` a VIEW of ``feat_visual`` and :367 is skipped, so the in-place write of :373 lands in
            # the tensor the reference returns and :396 differentiates with respect to the post-write tensor
            feat_visual = feat.view_as(feat_visual)
            feat = feat_visual.view(b, 21, -1)
    feat_out = vt_forward(sd, feat, "transformer.", 3, heads, 64).reshape(b, -1)  # :375-377
    pred = mean_params.repeat(b, 1).clone()
    pred[:, 3:] = pred[:, 3:] + feat_out  # :383
    for _ in range(iteration):  # :385-387
        pred = pred + F.linear(torch.cat((feat1024, pred), dim=1), sd["regressor.weight"], sd["regressor.bias"])
    j = pred[:, 3:66].reshape(-1, 21, 3)
    j = j - j[:, 1:2, :]  # :389-391
    pred = torch.cat((pred[:, :3], j.reshape(-1, 63)), dim=1)
    if pl_reg:
        pl = torch.autograd.grad(feat_out.sum(), feat_visual, retain_graph=True)[0]  # :396
        return pred, feat_visual, pl
    return pred, feat_visual


def encoder_transformer_coarse_forward(sd, mean_params, x, pos_embed=True, mask_rate=0.2, pl_reg=True,
                                       training=True, masked=None):
    """EncoderTransformerCoarse.forward, models/hand_net.py:262-311."""
    feat1024, x1, x2, x3, x4 = resnet_forward(sd, x, "main_encoder.", training)
    feat_visual = F.conv2d(x2, sd["conv1x1_channel_reduction.weight"])
    b = feat_visual.size(0)
    feat = feat_visual.reshape(b, 21, -1)
    if pos_embed:
        feat = feat + sd["positionalEncoding.pe"][: feat.size(0)]
    if masked is None:
        masked = mask_indices(mask_rate)
    if len(masked):
        feat = feat.clone()
        feat[:, masked, :] = sd["mask_token"]
    feat_out, attn = vt_attn_forward(sd, feat, "transformer.", 3, 8, 64)
    feat_out = feat_out.reshape(b, -1)
    pred = mean_params.repeat(b, 1).clone()
    pred[:, 3:] = pred[:, 3:] + feat_out
    cameras = F.linear(torch.cat((feat1024, pred[:, :3]), dim=1), sd["regressor.weight"], sd["regressor.bias"])
    j = pred[:, 3:66].reshape(-1, 21, 3)
    j = j - j[:, 1:2, :]
    pred = torch.cat((cameras, j.reshape(-1, 63)), dim=1)
    if pl_reg:
        pl = torch.autograd.grad(feat_out.sum(), feat_visual, retain_graph=True)[0]
        return pred, feat_visual, attn, pl
    return pred, feat_visual, attn


# --------------------------------------------------------------------------
# train step — train.py:112-120,158-209
# --------------------------------------------------------------------------

def scat_loss(outputs, labels, pl_term=None, w3d=100000.0, w2d=10.0):
    """train.py:165-203. Returns (loss, l_3d, l_2d, l_pl)."""
    cam = outputs[:, :3].reshape(-1, 1, 3)
    j3 = outputs[:, 3:66].reshape(-1, 21, 3)
    xt = j3[:, :, :2] + cam[:, :, 1:]  # :115
    j2 = (cam[:, :, 0] * xt.reshape(xt.size(0), -1)).reshape(xt.size(0), 21, 2) * 112 + 112  # :116-120
    if pl_term is not None:  # :178-183
        pl_len = pl_term.square().sum(dim=[2, 3]).mean(dim=[1]).sqrt()
        pl_mean = 0.0 + 0.01 * (pl_len.mean() - 0.0)
        l_pl = (pl_len - pl_mean).square().mean()
    else:
        l_pl = torch.zeros((), dtype=outputs.dtype)
    if labels.size(1) == 105:  # :188-192
        g3, g2 = labels[:, :63], labels[:, 63:]
    else:  # :193-198
        g3, g2 = labels[:, 61:124], labels[:, 124:]
    l3 = F.mse_loss(j3.reshape(-1, 63), g3)
    l2 = F.l1_loss(j2.reshape(-1, 42), g2)
    loss = w3d * l3 + w2d * l2
    if pl_term is not None:
        loss = loss + 10 * l_pl
    return loss, l3, l2, l_pl


def mpjpe_mm(pred66, gt63):
    """3-D MPJPE in mm, eval.py:753 formula."""
    p = pred66[:, 3:66].reshape(-1, 21, 3)
    g = gt63.reshape(-1, 21, 3)
    return (p - g).norm(dim=-1).mean() * 1000.0


PARAM_SKIP = ("running_mean", "running_var", "num_batches_tracked", "positionalEncoding.pe")


def trainable(sd):
    return OrderedDict((k, v) for k, v in sd.items() if not k.endswith(PARAM_SKIP))


def adam_update(params, grads, state, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (train.py:60), single-tensor formula."""
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    for k in params:
        g = grads[k]
        m, v = state.setdefault(k, (torch.zeros_like(g), torch.zeros_like(g)))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m, denom, value=-lr / bc1)


def train_step(sd, mean_params, x, labels, adam_state, step, lr=5e-4, masked=None, **kw):
    """One Trainer.train inner iteration (train.py:154-209): forward, loss, backward, Adam.
    ``sd`` tensors are updated in place. Returns dict of scalars + grads."""
    params = trainable(sd)
    for p in params.values():
        p.requires_grad_(True)
    out = encoder_transformer_forward(sd, mean_params, x, masked=masked, **kw)
    pl = out[2] if len(out) == 3 else None
    loss, l3, l2, lpl = scat_loss(out[0], labels, pl)
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    grads = OrderedDict((k, g if g is not None else torch.zeros_like(p))
                        for (k, p), g in zip(params.items(), grads))
    with torch.no_grad():
        for p in params.values():
            p.requires_grad_(False)
        adam_update(params, grads, adam_state, lr, step)
    return {"loss": loss.detach(), "l3d": l3.detach(), "l2d": l2.detach(), "lpl": lpl.detach(),
            "pred": out[0].detach(), "feat_visual": out[1].detach(), "pl": pl, "grads": grads}
