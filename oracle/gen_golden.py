"""Generate tests/golden/*.npz by running the REAL reference modules on CPU.

Runs only in the build container (needs /root/reference). The reference's files
are imported in place, never copied; only inputs' seeds and expected OUTPUTS are
stored. Inputs/weights are regenerated from scat_amd.synth on every machine.

    python oracle/gen_golden.py [--only NAME]

Two host shims are required because the reference hard-codes CUDA + network
(models/hand_net.py:321 ``.cuda()``; models/resnet.py:194 ``model_zoo.load_url``).
"""
from __future__ import annotations

import argparse
import os
import random
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = os.environ.get("SCAT_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

torch.Tensor.cuda = lambda self, *a, **k: self  # hand_net.py:321
import torch.utils.model_zoo as _mz  # noqa: E402

_mz.load_url = lambda *a, **k: {}  # resnet.py:194-195 (strict=False load of {})

from scat_amd import synth  # noqa: E402
from oracle import scat_oracle as O  # noqa: E402
from oracle.util import digest  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.manual_seed(0)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load_strict(mod, sd):
    missing = mod.load_state_dict(sd, strict=True)
    return missing


def opt_ns(**kw):
    d = dict(vit_heads=8, pl_reg=True, iteration=3, pos_embed=True, mask_rate=0.2, vit_depth=3)
    d.update(kw)
    return SimpleNamespace(**d)


# --------------------------------------------------------------------------

def g_vt():
    """G1/G2: vision_transformer.Transformer(784,3,8,64,392) fwd + all grads."""
    from models import vision_transformer as VT

    net = VT.Transformer(dim=784, depth=3, heads=8, dim_head=64, mlp_dim=392, dropout=0.0)
    sd = synth.to_torch(synth.vt_state(11, "", 784, 3, 8, 64))
    load_strict(net, sd)
    x = T(synth.normal_like(12, "x", (2, 21, 784))).requires_grad_(True)
    cot = T(synth.normal_like(13, "cot", (2, 21, 3)))
    y = net(x, None)
    (y * cot).sum().backward()
    out = {"y": y.detach().numpy(), "dx": digest(x.grad), "dx_head": x.grad[0, :2, :8].numpy()}
    for k, p in net.named_parameters():
        out["g:" + k] = digest(p.grad)
    out["g_full:layers.2.1.net.2.weight"] = net.layers[2][1].net[2].weight.grad.numpy()
    out["g_full:layers.2.1.net.2.bias"] = net.layers[2][1].net[2].bias.grad.numpy()
    # single attention block per dim (G1)
    for dim in (784, 392, 196):
        att = VT.Attention(dim, heads=8, dim_head=64, dropout=0.0)
        asd = synth.to_torch(synth.vt_state(21, "", dim, 1, 8, 64))
        att.load_state_dict({"to_qkv.weight": asd["layers.0.0.fn.fn.to_qkv.weight"],
                             "to_out.0.weight": asd["layers.0.0.fn.fn.to_out.0.weight"],
                             "to_out.0.bias": asd["layers.0.0.fn.fn.to_out.0.bias"]}, strict=True)
        xa = T(synth.normal_like(22, f"xa{dim}", (2, 21, dim))).requires_grad_(True)
        ya = att(xa)
        (ya * T(synth.normal_like(23, f"ca{dim}", (2, 21, dim)))).sum().backward()
        out[f"attn{dim}:y"] = digest(ya)
        out[f"attn{dim}:y_head"] = ya[0, :3, :8].detach().numpy()
        out[f"attn{dim}:dx"] = digest(xa.grad)
        out[f"attn{dim}:dwqkv"] = digest(att.to_qkv.weight.grad)
    np.savez(os.path.join(GOLD, "vt.npz"), **out)


def g_bottleneck():
    """G3: one Bottleneck(64→64→256, downsample) B=4 8×8: train fwd/bwd + running stats, eval fwd."""
    from models import resnet as R

    ds = torch.nn.Sequential(torch.nn.Conv2d(64, 256, 1, 1, bias=False), torch.nn.BatchNorm2d(256))
    blk = R.Bottleneck(64, 64, 1, ds)
    full = synth.resnet_state(31, "", (1, 0, 0, 0))
    sd = {k[len("layer1.0."):]: v for k, v in synth.to_torch(full).items() if k.startswith("layer1.0.")}
    load_strict(blk, sd)
    x = T(synth.normal_like(32, "x", (4, 64, 8, 8))).requires_grad_(True)
    cot = T(synth.normal_like(33, "cot", (4, 256, 8, 8)))
    blk.train()
    y = blk(x)
    (y * cot).sum().backward()
    out = {"y_train": y.detach().numpy()[:, ::37], "y_train_d": digest(y), "dx": digest(x.grad),
           "dx_head": x.grad[0, :2].numpy()}
    for k, p in blk.named_parameters():
        out["g:" + k] = digest(p.grad)
    for k, b in blk.named_buffers():
        out["buf:" + k] = b.double().numpy()
    blk.eval()
    out["y_eval_d"] = digest(blk(x))
    np.savez(os.path.join(GOLD, "bottleneck.npz"), **out)


def g_resnet():
    """G4: resnet50 B=2 224×224 train + eval forward."""
    from models import resnet as R

    net = R.resnet50(pretrained=True, num_classes=512)
    load_strict(net, synth.to_torch(synth.resnet_state(41, "")))
    x = T(synth.images(42, 2))
    out = {}
    for mode in ("train", "eval"):
        net.train(mode == "train")
        with torch.no_grad():
            feat, x1, x2, x3, x4 = net(x)
        out[f"{mode}:feat"] = feat.numpy()
        for n, t in (("x1", x1), ("x2", x2), ("x3", x3), ("x4", x4)):
            out[f"{mode}:{n}"] = digest(t, 64)
            out[f"{mode}:{n}_chsum"] = t.double().sum(dim=(0, 2, 3)).numpy()
        if mode == "train":
            out["bn1.running_mean"] = net.bn1.running_mean.numpy().copy()
            out["bn1.running_var"] = net.bn1.running_var.numpy().copy()
            out["layer4.2.bn3.running_var"] = net.layer4[2].bn3.running_var.numpy().copy()
    np.savez(os.path.join(GOLD, "resnet50.npz"), **out)


def _enc(seed=51, heads=8, **kw):
    from models.hand_net import EncoderTransformer

    mp = T(synth.mean_params(seed))
    net = EncoderTransformer(opt_ns(vit_heads=heads, **kw), mp)
    sd = synth.to_torch(synth.encoder_transformer_state(seed, heads))
    assert len(sd) == 356, len(sd)
    load_strict(net, sd)
    return net, mp


def g_encoder():
    """G5 + G10: full EncoderTransformer B=4, heads 8, iteration 3, pl_reg, mask .2, random.seed(3)."""
    net, mp = _enc()
    x = T(synth.images(52, 4))
    lab = T(synth.labels(53, 4))
    random.seed(3)
    net.train()
    pred, fv, pl = net(x)
    loss, l3, l2, lpl = O.scat_loss(pred, lab, pl)
    loss.backward()
    out = {"pred": pred.detach().numpy(), "fv": digest(fv, 64), "pl": digest(pl, 64),
           "fv_chsum": fv.detach().double().sum(dim=(0, 2, 3)).numpy(),
           "pl_chsum": pl.detach().double().sum(dim=(0, 2, 3)).numpy(),
           "loss": np.array([loss.item(), l3.item(), l2.item(), lpl.item()]),
           "mpjpe": np.array(O.mpjpe_mm(pred.detach(), lab[:, :63]).item())}
    full = ("regressor.weight", "regressor.bias", "mask_token", "conv1x1_channel_reduction.weight",
            "transformer.layers.2.1.net.2.weight")
    for k, p in net.named_parameters():
        out["g:" + k] = digest(p.grad, 8)
        if k in full:
            out["g_full:" + k] = p.grad.numpy()
    out["g_head:main_encoder.conv1.weight"] = net.main_encoder.conv1.weight.grad[:4].numpy()
    # eval-mode forward too (masking still active, hand_net.py:369)
    net2, _ = _enc()
    net2.eval()
    random.seed(3)
    pe, fve, ple = net2(x)
    out["eval:pred"] = pe.detach().numpy()
    out["eval:fv"] = digest(fve, 64)
    np.savez(os.path.join(GOLD, "encoder.npz"), **out)


def g_encoder_nope():
    """hand_net.py:364-373 with pos_embed=False and masking on: ``feat`` is a view of ``feat_visual``, so the mask
    token is written into the returned ``feat_visual`` and ``pl_term`` is taken at the post-write tensor."""
    net, mp = _enc(pos_embed=False)
    x = T(synth.images(52, 2))
    random.seed(3)
    net.train()
    pred, fv, pl = net(x)
    (pred * T(synth.normal_like(54, "cot", (2, 66)))).sum().backward()
    out = {"pred": pred.detach().numpy(), "fv": digest(fv, 64), "pl": digest(pl, 64),
           "fv_chsum": fv.detach().double().sum(dim=(0, 2, 3)).numpy(),
           "pl_chsum": pl.detach().double().abs().sum(dim=(0, 2, 3)).numpy(),
           "g_full:mask_token": net.mask_token.grad.numpy(),
           "g_full:conv1x1_channel_reduction.weight": net.conv1x1_channel_reduction.weight.grad.numpy()}
    np.savez(os.path.join(GOLD, "encoder_nope.npz"), **out)


def g_h3dw():
    """H3DWEncoder (hand_net.py:28-58; imported by eval.py:26), batch 1 (the only batch it runs at), eval and train."""
    from models.hand_net import H3DWEncoder

    mp = T(synth.normal_like(81, "mean61", (1, 61))) * 0.1
    net = H3DWEncoder(opt_ns(), mp)
    sd = {k: v for k, v in synth.to_torch(synth.encoder_transformer_state(81, 8)).items() if k.startswith("main_encoder.")}
    for k, shp, s in (("feat_encoder.1.weight", (1024, 1024), 1024 ** -0.5), ("feat_encoder.1.bias", (1024,), 0.05),
                      ("regressor.0.weight", (61, 1085), 1085 ** -0.5), ("regressor.0.bias", (61,), 0.05)):
        sd[k] = T(synth.normal_like(82, k, shp)) * s
    load_strict(net, sd)
    x = T(synth.images(83, 1))
    out = {"keys": np.array(list(net.state_dict().keys()))}
    net.eval()
    feat, pred = net(x)
    out["eval:feat"] = feat.detach().numpy()
    out["eval:pred"] = pred.detach().numpy()
    # head gradients on a frozen, eval-mode backbone (batch-1 train-mode BatchNorm statistics of a 7x7 map are noise;
    # fine-tuning the regressor on fixed features is the one way this class is trained at batch 1)
    net.train()
    net.main_encoder.eval()
    for p in net.main_encoder.parameters():
        p.requires_grad_(False)
    feat, pred = net(x)
    (pred * T(synth.normal_like(84, "cot", (1, 61)))).sum().backward()
    out["train:pred"] = pred.detach().numpy()
    for k in ("feat_encoder.1.weight", "feat_encoder.1.bias", "regressor.0.weight", "regressor.0.bias"):
        out["g:" + k] = digest(dict(net.named_parameters())[k].grad, 16)
    np.savez(os.path.join(GOLD, "h3dw.npz"), **out)


def g_trainstep():
    """G6: two full train steps (reference net + torch.optim.Adam + restated train.py loss)."""
    net, mp = _enc(seed=61)
    optim = torch.optim.Adam(net.parameters(), lr=5e-4)
    net.train()
    random.seed(5)
    out = {}
    for step in (1, 2):
        x = T(synth.images(62 + step, 4))
        lab = T(synth.labels(72 + step, 4))
        optim.zero_grad()
        pred, fv, pl = net(x)
        loss, l3, l2, lpl = O.scat_loss(pred, lab, pl)
        loss.backward()
        optim.step()
        out[f"s{step}:loss"] = np.array([loss.item(), l3.item(), l2.item(), lpl.item()])
        out[f"s{step}:pred"] = pred.detach().numpy()
        out[f"s{step}:regressor.weight"] = digest(net.regressor.weight, 32)
        out[f"s{step}:regressor.bias"] = net.regressor.bias.detach().numpy().copy()
        out[f"s{step}:bn1.running_mean"] = net.main_encoder.bn1.running_mean.numpy().copy()
        out[f"s{step}:conv1.weight"] = digest(net.main_encoder.conv1.weight, 32)
        out[f"s{step}:layer3.0.conv2.weight"] = digest(net.main_encoder.layer3[0].conv2.weight, 32)
    out["nbt"] = np.array(net.main_encoder.bn1.num_batches_tracked.item())
    np.savez(os.path.join(GOLD, "trainstep.npz"), **out)


def g_vit():
    """G8a: vit.Transformer(196,3,8,64,392,0.0) on [2,128,196]."""
    from models import vit as V

    net = V.Transformer(196, 3, 8, 64, 392, 0.0)
    load_strict(net, synth.to_torch(synth.vit_state(81, "")))
    x = T(synth.normal_like(82, "x", (2, 128, 196))).requires_grad_(True)
    y = net(x)
    (y * T(synth.normal_like(83, "cot", (2, 128, 196)))).sum().backward()
    out = {"y": digest(y, 64), "y_head": y[0, :4, :16].detach().numpy(), "dx": digest(x.grad, 64)}
    for k, p in net.named_parameters():
        out["g:" + k] = digest(p.grad, 8)
    np.savez(os.path.join(GOLD, "vit.npz"), **out)


def g_performer():
    """G7: performer_attn_block(49,16) eval fwd/bwd on [2,21,784]."""
    from models import vision_performer as P

    blk = P.performer_attn_block(49, 16)
    load_strict(blk, synth.to_torch(synth.performer_state(91, "")))
    blk.eval()
    x = T(synth.normal_like(92, "x", (2, 21, 784), std=0.5)).requires_grad_(True)
    y = blk(x)
    (y * T(synth.normal_like(93, "cot", (2, 21, 784)))).sum().backward()
    out = {"y": digest(y, 64), "y_head": y[0, :4, :16].detach().numpy(), "dx": digest(x.grad, 64)}
    for k, p in blk.named_parameters():
        if p.grad is not None:
            out["g:" + k] = digest(p.grad, 8)
    # ViP(heads=16, emb_s=49, iteration=5) on 64x64 images, eval mode (dropout off)
    vip = P.ViP(opt_ns(iteration=5), T(synth.mean_params(94, 10)), heads=16, emb_s=49)
    load_strict(vip, synth.to_torch(synth.vip_state(95, vip.state_dict())))
    vip.eval()
    xi = T(synth.images(96, 2, 64))
    pv = vip(xi)
    pv.square().sum().backward()
    assert all(torch.isfinite(q.grad).all() for q in vip.parameters() if q.grad is not None), "NaN in reference"
    out["vip:pred"] = pv.detach().numpy()
    out["vip:g:head.weight"] = digest(vip.head.weight.grad, 8)
    out["vip:g:patch_emb.weight"] = digest(vip.patch_emb.weight.grad, 8)
    out["vip:g:mains.0.kqv.weight"] = digest(vip.mains[0].kqv.weight.grad, 8)
    out["vip:g:cls_token"] = digest(vip.cls_token.grad, 8)
    out["vip:g:pos_emb"] = digest(vip.pos_emb.grad, 8)
    np.savez(os.path.join(GOLD, "performer.npz"), **out)


def g_hrnet():
    """G8b: HRNet(c=32, nof_joints=128) B=1 224x224 train-mode fwd + bwd (reference modules), and the wrapper
    (HRNet c=24 -> view 512x28x28 -> conv3x3/2 -> tokens -> vit.Transformer -> mean -> 3x Linear(257->61))."""
    from models import hrnet as H
    from models import vit as V
    from models.hand_net import EncoderTransformerHRNet

    net = H.HRNet(c=32, nof_joints=128, bn_momentum=0.1)
    load_strict(net, synth.to_torch(synth.fill_state(101, net.state_dict())))
    net.train()
    x = T(synth.images(102, 1))
    y = net(x)
    (y * T(synth.normal_like(103, "cot", tuple(y.shape)))).sum().backward()
    out = {"y": digest(y, 64), "y_chsum": y.detach().double().sum(dim=(0, 2, 3)).numpy(),
           "y_head": y[0, :4, :4, :8].detach().numpy(),
           "stage4.2.bn.rm": net.stage4[2].branches[0][3].bn2.running_mean.numpy().copy()}
    for k in ("conv1.weight", "final_layer.weight", "final_layer.bias", "stage3.1.fuse_layers.0.2.0.weight",
              "stage4.2.fuse_layers.0.3.1.weight", "stage2.0.branches.1.2.conv1.weight", "transition2.2.0.0.weight",
              "layer1.0.downsample.0.weight"):
        out["g:" + k] = digest(dict(net.named_parameters())[k].grad, 8)
    net.eval()
    with torch.no_grad():
        out["y_eval"] = digest(net(x), 64)
    # wrapper (reference class with the working transformer swapped in)
    w = EncoderTransformerHRNet(opt_ns(), T(synth.mean_params(104, 61)))
    w.transformer = V.Transformer(196, 3, 8, 64, 392, 0.0)
    load_strict(w, synth.to_torch(synth.hrnet_wrapper_state(105, w.state_dict())))
    w.train()
    random.seed(7)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):   # the reference prints feat.shape every call
        p = w(T(synth.images(106, 2)))
    p.square().sum().backward()
    out["wrap:pred"] = p.detach().numpy()
    out["wrap:g:regressor.0.weight"] = digest(w.regressor[0].weight.grad, 8)
    out["wrap:g:mask_token"] = digest(w.mask_token.grad, 8)
    out["wrap:g:conv1x1_channel_reduction.weight"] = digest(w.conv1x1_channel_reduction.weight.grad, 8)
    out["wrap:g:transformer.layers.0.0.fn.to_qkv.weight"] = digest(w.transformer.layers[0][0].fn.to_qkv.weight.grad, 8)
    np.savez(os.path.join(GOLD, "hrnet.npz"), **out)


def g_coarse():
    """train_coarse.py network: EncoderTransformerCoarse B=2 (reference module), fwd 4-tuple + a few grads."""
    from models.hand_net import EncoderTransformerCoarse

    net = EncoderTransformerCoarse(opt_ns(), T(synth.mean_params(111)))
    load_strict(net, synth.to_torch(synth.fill_state(112, net.state_dict())))
    net.train()
    random.seed(9)
    x, lab = T(synth.images(113, 2)), T(synth.labels(114, 2))
    pred, fv, attn, pl = net(x)
    loss, *_ = O.scat_loss(pred, lab, pl)
    loss.backward()
    out = {"pred": pred.detach().numpy(), "attn": digest(attn, 64), "attn_head": attn[0, :2, :4, :8].detach().numpy(),
           "fv": digest(fv, 64), "pl": digest(pl, 64), "loss": np.array(loss.item())}
    for k in ("regressor.weight", "regressor.bias", "mask_token", "transformer.layers.0.1.norm.weight",
              "transformer.layers.2.2.net.2.weight", "transformer.layers.1.0.to_qkv.weight",
              "conv1x1_channel_reduction.weight"):
        out["g:" + k] = digest(dict(net.named_parameters())[k].grad, 8)
    np.savez(os.path.join(GOLD, "coarse.npz"), **out)


def g_dp():
    """G9: the data-parallel parity definition.  Two shards of 4 images go through the REAL reference
    network independently from identical weights (train.py has no distributed code: this is what N independent
    train.py processes would compute), the two gradient sets are averaged, Adam is applied once.  An N-replica run
    must reproduce the averaged gradients and the updated weights.  Seeds as tests/test_gpu_dp.py uses them."""
    shards = []
    for r in range(2):
        net, mp = _enc(seed=43)
        net.train()
        x = T(synth.images(700 + r, 4))
        lab = T(synth.labels(710 + r, 4))
        random.seed(11)
        pred, fv, pl = net(x)
        loss, l3, l2, lpl = O.scat_loss(pred, lab, pl)
        loss.backward()
        shards.append((net, loss.item(), pred.detach().numpy()))
    out = {"loss": np.array([s[1] for s in shards]), "pred0": shards[0][2], "pred1": shards[1][2]}
    net = shards[0][0]
    full = ("regressor.weight", "regressor.bias", "mask_token", "conv1x1_channel_reduction.weight",
            "transformer.layers.2.1.net.2.weight", "transformer.layers.0.0.fn.norm.weight")
    for (k, p), (_, q) in zip(net.named_parameters(), shards[1][0].named_parameters()):
        g = (p.grad + q.grad) / 2
        out["g:" + k] = digest(g, 8)
        out["gnorm:" + k] = np.array([g.norm().item(), p.grad.norm().item(), q.grad.norm().item()])
        if k in full:
            out["g_full:" + k] = g.numpy()
        p.grad = g
    for r in range(2):      # BatchNorm statistics stay per replica (no SyncBN; what DDP does)
        out[f"bn1.running_mean:{r}"] = shards[r][0].main_encoder.bn1.running_mean.numpy().copy()
    optim = torch.optim.Adam(net.parameters(), lr=1e-4)
    optim.step()
    for k in full + ("main_encoder.fc1.bias", "main_encoder.layer4.2.bn3.weight"):
        w = dict(net.named_parameters())[k].detach()
        out["w:" + k] = w.numpy().copy() if w.numel() <= 4096 else digest(w, 64)
    np.savez(os.path.join(GOLD, "dp.npz"), **out)


def _eval_py_functions(*names):
    """eval.py cannot be imported (it needs torchvision, cv2 and modules the repository does not ship: eval.py:37-48),
    but its metric functions are self-contained: take their definitions out of the file where it lies (ast -> compile
    -> exec in a namespace that has torch and numpy) and run THEM.  Nothing is copied into this repository."""
    import ast

    path = os.path.join(REF, "eval.py")
    tree = ast.parse(open(path).read(), path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(body) == len(names), [n.name for n in body]
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def metric_inputs():
    """(pred[B,21,3], gt[B,21,3]) of the metric goldens: a rotated, scaled, shifted, noisy copy of gt (metres)"""
    B = 12
    gt = synth.normal_like(901, "gt", (B, 21, 3), 0.05)
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], dtype=np.float32)
    pred = (1.3 * gt @ R.T + 0.02 + synth.normal_like(902, "n", (B, 21, 3), 0.004)).astype(np.float32)
    return pred, gt


def g_metrics():
    """G10+: the reference's own evaluation formulae on seeded joints — MPJPE (eval.py:753), Procrustes
    alignment (eval.py:110-161, as applied at :953) and PA-MPJPE, PCK / AUC (eval.py:300-340 with the thresholds of
    :708/:744), acceleration error (data_utils/eval_utils.py:23-48, imported)."""
    from data_utils.eval_utils import compute_accel, compute_error_accel

    pa, cal_pck, area = _eval_py_functions("batch_compute_similarity_transform_torch", "cal_PCK", "_area_under_curve")
    pred, gt = metric_inputs()
    tp, tg = T(pred), T(gt)
    out = {"mpjpe_per_sample": torch.sqrt(((tp - tg) ** 2).sum(dim=-1)).mean(dim=-1).numpy()}       # eval.py:753
    out["mpjpe_mm"] = np.array(1000 * out["mpjpe_per_sample"].mean())                                 # eval.py:1050
    aligned = pa(tp, tg)
    out["pa_aligned"] = aligned.numpy()
    out["pa_mpjpe_mm"] = np.array(1000 * torch.sqrt(((aligned - tg) ** 2).sum(dim=-1)).mean(dim=-1).numpy().mean())
    rnge = np.arange(20, 51, 1.0)
    pck = cal_pck(tp, tg, rnge)
    out["rnge"] = rnge
    out["pck"] = pck[:, -1]
    out["auc"] = np.array(area(rnge / rnge.max(), pck[:, -1]))
    out["pck_pa"] = cal_pck(aligned, tg, rnge)[:, -1]
    out["accel_err"] = compute_error_accel(joints_gt=gt, joints_pred=pred)
    out["accel"] = compute_accel(pred)
    np.savez(os.path.join(GOLD, "metrics.npz"), **out)


def g_ckpt():
    """f4: the on-disk format.  The reference's own EncoderTransformer, loaded with the synthesised
    weights, is saved with torch.save(net.state_dict()) exactly as train.py:237-246 does and read back; the golden
    keeps the checkpoint's key order, shapes and a 4-number digest per tensor (not the 118 MB of values)."""
    import tempfile

    net, mp = _enc()
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "ref.pth")
        torch.save(net.state_dict(), f)
        sd = torch.load(f, map_location="cpu")
    keys = list(sd.keys())
    shapes = np.full((len(keys), 4), -1, dtype=np.int64)
    for i, k in enumerate(keys):
        shapes[i, :sd[k].dim()] = list(sd[k].shape)
    np.savez(os.path.join(GOLD, "ckpt_keys.npz"), keys=np.array(keys), shapes=shapes,
             digests=np.stack([digest(sd[k].float(), 4)[:4] for k in keys]))


ALL = {"encoder_nope": g_encoder_nope, "h3dw": g_h3dw, "coarse": g_coarse, "hrnet": g_hrnet, "vt": g_vt, "bottleneck": g_bottleneck, "resnet": g_resnet, "encoder": g_encoder,
       "trainstep": g_trainstep, "vit": g_vit, "performer": g_performer, "dp": g_dp, "metrics": g_metrics, "ckpt": g_ckpt}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    for name, fn in ALL.items():
        if a.only and a.only != name:
            continue
        print("golden:", name, flush=True)
        fn()
    print("done")
