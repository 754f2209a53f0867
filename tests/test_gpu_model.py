"""Model-level parity on a real MI355X, through the C ABI: the HIP path against
(a) the committed outputs of the REAL reference modules (tests/golden, made by oracle/gen_golden.py)
(b) the CPU oracle on the same seeded inputs at other sizes.

Bar (BASELINE.json north_star): 21-joint offsets pred[:,3:66] within 1e-4 relative (norm-wise,
max|a-b|/max|b|) of the reference CPU fp32 path; MPJPE within 1e-4 relative."""
import random
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import scat_oracle as O
from oracle.util import digest, digest_err, rel_err
from scat_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def opt_ns(**kw):
    d = dict(vit_heads=8, pl_reg=True, iteration=3, pos_embed=True, mask_rate=0.2, vit_depth=3)
    d.update(kw)
    return SimpleNamespace(**d)


def make_encoder(seed, **kw):
    from scat_amd.models.hand_net import EncoderTransformer

    net = EncoderTransformer(opt_ns(**kw), T(synth.mean_params(seed)))
    net.load_state_dict(synth.to_torch(synth.encoder_transformer_state(seed, kw.get("vit_heads", 8))), strict=True)
    return net.cuda()


def test_vt_transformer_golden(golden):
    from scat_amd.models import vision_transformer as VT

    g = golden("vt")
    net = VT.Transformer(dim=784, depth=3, heads=8, dim_head=64, mlp_dim=392, dropout=0.0)
    net.load_state_dict(synth.to_torch(synth.vt_state(11, "", 784, 3, 8, 64)), strict=True)
    net.cuda()
    x = T(synth.normal_like(12, "x", (2, 21, 784))).cuda().requires_grad_(True)
    y = net(x, None)
    assert rel_err(y, g["y"]) < 2e-5
    (y * T(synth.normal_like(13, "cot", (2, 21, 3))).cuda()).sum().backward()
    assert digest_err(digest(x.grad), g["dx"]) < 5e-5
    assert rel_err(x.grad[0, :2, :8], g["dx_head"]) < 5e-5
    for k, p in net.named_parameters():
        assert digest_err(digest(p.grad), g["g:" + k]) < 5e-5, k
    # input-gradient replay == autograd (the pose-length term path)
    ig = net.input_grad(T(synth.normal_like(13, "cot", (2, 21, 3))).cuda())
    assert rel_err(ig, x.grad) < 1e-6


def test_small_modules_standalone(golden):
    """Attention / FeedForward / PreNorm usable on their own like the reference classes."""
    from scat_amd.models import vision_transformer as VT

    g = golden("vt")
    for dim in (784, 392, 196):
        att = VT.Attention(dim, heads=8, dim_head=64, dropout=0.0)
        asd = synth.to_torch(synth.vt_state(21, "", dim, 1, 8, 64))
        att.load_state_dict({"to_qkv.weight": asd["layers.0.0.fn.fn.to_qkv.weight"],
                             "to_out.0.weight": asd["layers.0.0.fn.fn.to_out.0.weight"],
                             "to_out.0.bias": asd["layers.0.0.fn.fn.to_out.0.bias"]}, strict=True)
        att.cuda()
        xa = T(synth.normal_like(22, f"xa{dim}", (2, 21, dim))).cuda().requires_grad_(True)
        ya = att(xa)
        (ya * T(synth.normal_like(23, f"ca{dim}", (2, 21, dim))).cuda()).sum().backward()
        assert digest_err(digest(ya), g[f"attn{dim}:y"]) < 2e-5
        assert digest_err(digest(xa.grad), g[f"attn{dim}:dx"]) < 5e-5
        assert digest_err(digest(att.to_qkv.weight.grad), g[f"attn{dim}:dwqkv"]) < 5e-5


def test_vit_golden(golden):
    from scat_amd.models import vit as V

    g = golden("vit")
    net = V.Transformer(196, 3, 8, 64, 392, 0.0)
    net.load_state_dict(synth.to_torch(synth.vit_state(81, "")), strict=True)
    net.cuda()
    x = T(synth.normal_like(82, "x", (2, 128, 196))).cuda().requires_grad_(True)
    y = net(x)
    assert digest_err(digest(y, 64), g["y"]) < 2e-5
    (y * T(synth.normal_like(83, "cot", (2, 128, 196))).cuda()).sum().backward()
    assert digest_err(digest(x.grad, 64), g["dx"]) < 5e-5
    for k, p in net.named_parameters():
        assert digest_err(digest(p.grad, 8), g["g:" + k]) < 5e-5, k


def test_resnet50_golden(golden):
    from scat_amd.models import resnet as R

    g = golden("resnet50")
    net = R.resnet50(pretrained=True, num_classes=512)
    net.load_state_dict(synth.to_torch(synth.resnet_state(41, "")), strict=True)
    net.cuda()
    x = T(synth.images(42, 2)).cuda()
    for mode in ("train", "eval"):
        net.train(mode == "train")
        with torch.no_grad():
            feat, x1, x2, x3, x4 = net(x)
        assert rel_err(feat, g[f"{mode}:feat"]) < 5e-5, mode
        for n, t in (("x1", x1), ("x2", x2), ("x3", x3), ("x4", x4)):
            # 1e-4 = the path's stated fp32 parity bar; x4 sits behind 52 convolutions and (train mode, batch 2)
            # 98-sample batch statistics, so a different-but-valid fp32 summation order moves it by ~5e-5
            assert digest_err(digest(t, 64), g[f"{mode}:{n}"]) < 1e-4, (mode, n)
            assert rel_err(t.double().sum(dim=(0, 2, 3)), g[f"{mode}:{n}_chsum"]) < 5e-5
        if mode == "train":
            assert rel_err(net.bn1.running_mean, g["bn1.running_mean"]) < 1e-5
            assert rel_err(net.bn1.running_var, g["bn1.running_var"]) < 1e-5
            assert rel_err(net.layer4[2].bn3.running_var, g["layer4.2.bn3.running_var"]) < 1e-5
            assert int(net.bn1.num_batches_tracked) == 1


def test_encoder_transformer_golden(golden):
    """G5/G10: forward tuple, loss, every parameter gradient, MPJPE."""
    from scat_amd.trainer import pose_length_term, scat_loss

    g = golden("encoder")
    net = make_encoder(51)
    x, lab = T(synth.images(52, 4)).cuda(), T(synth.labels(53, 4)).cuda()
    random.seed(3)
    net.train()
    pred, fv, pl = net(x)
    assert rel_err(pred[:, 3:66], g["pred"][:, 3:66]) < 1e-4          # the north_star bar
    assert rel_err(pred, g["pred"]) < 1e-4
    assert float(pred[:, 6:9].abs().max()) == 0.0
    assert not pl.requires_grad
    assert digest_err(digest(fv, 64), g["fv"]) < 5e-5
    assert rel_err(fv.double().sum(dim=(0, 2, 3)), g["fv_chsum"]) < 5e-5
    assert digest_err(digest(pl, 64), g["pl"]) < 1e-4
    loss, parts = scat_loss(pred, lab)
    lpl = pose_length_term(pl)
    total = loss + 10 * lpl
    ref = g["loss"]
    assert abs(total.item() - ref[0]) / abs(ref[0]) < 1e-4
    assert abs(parts[1].item() - ref[1]) / abs(ref[1]) < 1e-4 and abs(parts[2].item() - ref[2]) / abs(ref[2]) < 1e-4
    assert abs(lpl.item() - ref[3]) / abs(ref[3]) < 1e-3
    mp = O.mpjpe_mm(pred.detach().cpu(), lab[:, :63].cpu()).item()
    assert abs(mp - float(g["mpjpe"])) / float(g["mpjpe"]) < 1e-4
    total.backward()
    # Gradients of this random-weight network are ill-conditioned in fp32: the reference's own CPU
    # fp32 gradients sit 2 % (median) to 9 % (worst parameter) from an fp64 evaluation of the same graph
    # (ReLU-mask / max-pool arg-max flips + train-mode BN at B=4; measured in test_against_oracle_config1).
    # Against the reference's fp32 goldens the backbone is therefore checked to that noise level, the
    # head (transformer, regressor, mask token, 1x1 reduction) tightly.
    named = dict(net.named_parameters())
    worst_bb, worst_head = ("", 0.0), ("", 0.0)
    for k, p in named.items():
        assert p.grad is not None, k
        e = digest_err(digest(p.grad, 8), g["g:" + k])
        if k.startswith("main_encoder."):
            worst_bb = max(worst_bb, (k, e), key=lambda t: t[1])
        else:
            worst_head = max(worst_head, (k, e), key=lambda t: t[1])
    assert worst_head[1] < 5e-4, worst_head
    assert worst_bb[1] < 0.25, worst_bb
    for k in ("regressor.weight", "regressor.bias", "mask_token", "conv1x1_channel_reduction.weight",
              "transformer.layers.2.1.net.2.weight"):
        assert rel_err(named[k].grad, g["g_full:" + k]) < 5e-4, k
    # eval mode: masking still active, BN uses running stats
    net2 = make_encoder(51)
    net2.eval()
    random.seed(3)
    with torch.no_grad():
        pe, fve, ple = net2(x)
    assert rel_err(pe, g["eval:pred"]) < 1e-4
    assert digest_err(digest(fve, 64), g["eval:fv"]) < 5e-5


def test_trainstep_golden(golden):
    """G6: two full train.py iterations (fused loss, flat buckets, fused Adam)."""
    from scat_amd.trainer import TrainStep

    g = golden("trainstep")
    net = make_encoder(61)
    net.train()
    ts = TrainStep(net, lr=5e-4)
    random.seed(5)
    for step in (1, 2):
        x, lab = T(synth.images(62 + step, 4)).cuda(), T(synth.labels(72 + step, 4)).cuda()
        loss, parts, lpl, pred = ts(x, lab)
        ref = g[f"s{step}:loss"]
        # step 1 is pure forward parity; step 2 runs on Adam-updated weights, and Adam's first update is
        # +-lr per weight whatever |g| is, so fp32 gradient noise (see above) moves the second loss at 1e-4..1e-3
        tol = 1e-4 if step == 1 else 3e-3
        assert abs(loss.item() - ref[0]) / abs(ref[0]) < tol, (step, loss.item(), ref)
        assert rel_err(pred, g[f"s{step}:pred"]) < (1e-4 if step == 1 else 2e-2)
        assert rel_err(net.regressor.bias, g[f"s{step}:regressor.bias"]) < 1e-3
        assert rel_err(net.main_encoder.bn1.running_mean, g[f"s{step}:bn1.running_mean"]) < 1e-4
        # (Adam moves every weight by +-lr = 5e-4 whatever |g| is: a near-zero gradient whose sign differs between two
        # fp32 evaluations moves a sampled weight by 1e-3 -- measured 1.1e-3 .. 2.5e-3 across math modes / stem kernels)
        assert digest_err(digest(net.regressor.weight, 32), g[f"s{step}:regressor.weight"]) < (2e-3 if step == 1 else 5e-3)
        assert digest_err(digest(net.main_encoder.layer3[0].conv2.weight, 32),
                          g[f"s{step}:layer3.0.conv2.weight"]) < (1e-4 if step == 1 else 5e-2)
    assert int(net.main_encoder.bn1.num_batches_tracked) == int(g["nbt"]) == 2


def test_against_oracle_config1():
    """BASELINE configs[0] geometry (heads 8, iteration 3; B=4 to keep the fp64 CPU pass short): HIP vs the
    CPU oracle on fresh seeds, forward + backward, no pose-length term / positional encoding / masking
    (exercises the flag paths).  Gradient criterion: the HIP fp32 gradient must be as close to an fp64
    evaluation of the same graph as the CPU fp32 oracle is (3x its distance + 1e-4)."""
    from scat_amd.trainer import scat_loss

    B = 4
    kw = dict(pl_reg=False, pos_embed=False, mask_rate=0.0)
    net = make_encoder(7, **kw)
    net.train()
    x, lab = T(synth.images(8, B)), T(synth.labels(9, B))

    def oracle(dt):
        sd = {k: (v.to(dt) if v.dtype == torch.float32 else v)
              for k, v in synth.to_torch(synth.encoder_transformer_state(7, 8)).items()}
        params = O.trainable(sd)
        for p in params.values():
            p.requires_grad_(True)
        pr, _ = O.encoder_transformer_forward(sd, T(synth.mean_params(7)).to(dt), x.to(dt), pl_reg=False,
                                              pos_embed=False, mask_rate=0.0)
        l, *_ = O.scat_loss(pr, lab.to(dt))
        l.backward()
        return pr.detach(), l.item(), {k: p.grad for k, p in params.items() if p.grad is not None}

    p32, l32, g32 = oracle(torch.float32)
    p64, l64, g64 = oracle(torch.float64)
    pred, fv = net(x.cuda())
    assert rel_err(pred[:, 3:66], p32[:, 3:66]) < 1e-4
    assert rel_err(pred[:, 3:66], p64[:, 3:66]) < 1e-4
    loss, _ = scat_loss(pred, lab.cuda())
    loss.backward()
    assert abs(loss.item() - l32) / abs(l32) < 1e-4
    named = dict(net.named_parameters())
    e_hip, e_cpu = [], []
    for k, gref in g64.items():
        if k == "mask_token":
            continue
        e_hip.append(rel_err(named[k].grad, gref))
        e_cpu.append(rel_err(g32[k], gref))
    e_hip, e_cpu = np.array(e_hip), np.array(e_cpu)
    # two fp32 evaluations land at independent random distances from the fp64 gradient: compare the
    # distributions, not parameter by parameter
    assert np.median(e_hip) <= 2 * np.median(e_cpu) + 1e-4, (np.median(e_hip), np.median(e_cpu))
    assert e_hip.max() <= 3 * e_cpu.max() + 1e-4, (e_hip.max(), e_cpu.max())
    assert np.mean(e_hip) <= 2 * np.mean(e_cpu) + 1e-4, (np.mean(e_hip), np.mean(e_cpu))
    assert named["mask_token"].grad is None or float(named["mask_token"].grad.abs().max()) == 0.0


@pytest.mark.timeout(600)
def test_full_size_batch96_against_oracle():
    """BASELINE configs[1] at its FULL size: batch 96 (the tile counts, split-K plans and BatchNorm reductions the bench
    runs, which no B <= 8 golden reaches), pl_reg, mask 0.2, positional encoding.  One training-mode forward + loss on
    the HIP path against the CPU oracle on the same 96 images (the oracle takes a few seconds at this size): the
    21-joint offsets within 1e-4 norm-wise — the north star's bar —, loss, MPJPE, feat_visual, the pose-length term,
    the running statistics after the step; then the size-independent properties of the step: root joint exactly zero,
    every gradient finite, Adam moves every trained parameter."""
    from scat_amd.trainer import TrainStep

    B = 96
    net = make_encoder(1)
    net.train()
    x, lab = T(synth.images(2, B)), T(synth.labels(3, B))
    sd = synth.to_torch(synth.encoder_transformer_state(1, 8))
    random.seed(3)
    with torch.no_grad():      # (the pose-length term needs autograd; pred and feat_visual do not depend on it)
        pr, fv = O.encoder_transformer_forward(sd, T(synth.mean_params(1)), x, pl_reg=False)
    l_ref, l3, l2, _ = O.scat_loss(pr, lab, None)
    before = net.regressor.weight.detach().clone()
    ts = TrainStep(net, lr=5e-4)
    random.seed(3)
    total, parts, lpl, pred = ts(x.cuda(), lab.cuda())
    assert rel_err(pred[:, 3:66], pr[:, 3:66]) < 1e-4
    assert rel_err(pred[:, :3], pr[:, :3]) < 1e-4
    assert float(pred[:, 6:9].abs().max()) == 0.0
    assert abs(parts[0].item() - l_ref.item()) / abs(l_ref.item()) < 1e-4
    assert abs(O.mpjpe_mm(pred.cpu(), lab[:, :63]).item() - O.mpjpe_mm(pr, lab[:, :63]).item()) < 1e-4 * O.mpjpe_mm(pr, lab[:, :63]).item()
    assert rel_err(net.main_encoder.bn1.running_mean, sd["main_encoder.bn1.running_mean"]) < 1e-4
    assert rel_err(net.main_encoder.layer4[2].bn3.running_var, sd["main_encoder.layer4.2.bn3.running_var"]) < 1e-4
    assert int(net.main_encoder.bn1.num_batches_tracked) == 1
    assert torch.isfinite(ts.buckets.flat_grad).all() and torch.isfinite(total)
    moved = (net.regressor.weight.detach() - before).abs()
    assert float(moved.max()) <= 5e-4 * 1.001 and float(moved.mean()) > 1e-4            # Adam's first step is +-lr


@pytest.mark.timeout(1800)
def test_full_size_batch96_gradients_against_fp64():
    """The BACKWARD of BASELINE configs[1] at its full size (VERDICT r02 "Next" item 1): the dgrad / wgrad instantiations,
    split-K plans and BatchNorm-backward reductions of a batch-96 step exist only at this size.  One HIP train step
    (pl_reg, mask 0.2, positional table; gradients read from the flat buckets) against the CPU oracle evaluated in fp32
    AND in fp64 on the same 96 images (about one and two minutes of host time):
      * head gradients (transformer, regressor, mask token, 1x1 reduction) within 5e-4 of fp64;
      * backbone gradients: the HIP fp32 gradient is as close to the fp64 one as the CPU fp32 gradient is — median and
        mean of the per-parameter distances within 2x, the worst within 3x (two fp32 evaluations land at independent
        distances from fp64, so the distributions are compared, not parameter by parameter);
      * the measured conditioning is written to gpurun_out/b96_gradient_conditioning.json."""
    import json
    import os
    from scat_amd.trainer import TrainStep

    B = 96
    x, lab = T(synth.images(2, B)), T(synth.labels(3, B))

    def oracle(dt):
        sd = {k: (v.to(dt) if v.dtype == torch.float32 else v)
              for k, v in synth.to_torch(synth.encoder_transformer_state(1, 8)).items()}
        params = O.trainable(sd)
        for p in params.values():
            p.requires_grad_(True)
        random.seed(3)
        pr, fv, pl = O.encoder_transformer_forward(sd, T(synth.mean_params(1)).to(dt), x.to(dt))
        loss, *_ = O.scat_loss(pr, lab.to(dt), pl)
        loss.backward()
        return pr.detach(), loss.item(), {k: p.grad for k, p in params.items() if p.grad is not None}

    net = make_encoder(1)
    net.train()
    ts = TrainStep(net, lr=5e-4)
    random.seed(3)
    total, parts, lpl, pred = ts(x.cuda(), lab.cuda())
    named = dict(net.named_parameters())
    g_hip = {k: p.grad.detach().cpu().clone() for k, p in named.items() if p.grad is not None}
    p32, l32, g32 = oracle(torch.float32)
    p64, l64, g64 = oracle(torch.float64)
    assert rel_err(pred[:, 3:66], p64[:, 3:66]) < 1e-4
    assert abs(total.item() - l64) / abs(l64) < 1e-4
    assert set(g_hip) == set(g64)
    head, bb_hip, bb_cpu, rows = [], [], [], {}
    for k, gref in g64.items():
        e_h, e_c = rel_err(g_hip[k], gref), rel_err(g32[k], gref)
        rows[k] = (e_h, e_c)
        if k.startswith("main_encoder."):
            bb_hip.append(e_h)
            bb_cpu.append(e_c)
        else:
            head.append((e_h, k))
    bb_hip, bb_cpu = np.array(bb_hip), np.array(bb_cpu)
    report = {"head_max_hip": max(head)[0], "head_max_hip_param": max(head)[1],
              "head_max_cpu": max(rows[k][1] for _, k in head),
              "backbone_hip": {"median": float(np.median(bb_hip)), "mean": float(bb_hip.mean()), "max": float(bb_hip.max())},
              "backbone_cpu_fp32": {"median": float(np.median(bb_cpu)), "mean": float(bb_cpu.mean()), "max": float(bb_cpu.max())},
              "worst_backbone_hip": max((v[0], k) for k, v in rows.items() if k.startswith("main_encoder."))[1]}
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump(report, open(os.path.join(out, "b96_gradient_conditioning.json"), "w"), indent=1)
    except OSError:
        pass
    print("batch-96 gradient conditioning:", report)
    assert max(head)[0] < 5e-4, report
    assert np.median(bb_hip) <= 2 * np.median(bb_cpu) + 1e-5, report
    assert bb_hip.mean() <= 2 * bb_cpu.mean() + 1e-5, report
    assert bb_hip.max() <= 3 * bb_cpu.max() + 1e-5, report


def test_hrnet_golden(golden):
    """BASELINE config 4 backbone: HRNet-W32 on the HIP kernels vs the reference modules (train fwd+bwd, eval)."""
    from scat_amd.models import hrnet as H

    g = golden("hrnet")
    net = H.HRNet(c=32, nof_joints=128, bn_momentum=0.1)
    net.load_state_dict(synth.to_torch(synth.fill_state(101, net.state_dict())), strict=True)
    net.cuda().train()
    x = T(synth.images(102, 1)).cuda()
    y = net(x)
    assert tuple(y.shape) == (1, 128, 56, 56)
    assert digest_err(digest(y, 64), g["y"]) < 1e-4
    assert rel_err(y[0, :4, :4, :8], g["y_head"]) < 1e-4
    assert rel_err(y.double().sum(dim=(0, 2, 3)), g["y_chsum"]) < 1e-4
    (y * T(synth.normal_like(103, "cot", tuple(y.shape))).cuda()).sum().backward()
    named = dict(net.named_parameters())
    for k in ("final_layer.weight", "final_layer.bias"):
        assert digest_err(digest(named[k].grad, 8), g["g:" + k]) < 1e-3, k      # at the output: tight
    for k in ("stage4.2.fuse_layers.0.3.1.weight", "conv1.weight", "stage3.1.fuse_layers.0.2.0.weight", "stage2.0.branches.1.2.conv1.weight",
              "transition2.2.0.0.weight", "layer1.0.downsample.0.weight"):
        assert digest_err(digest(named[k].grad, 8), g["g:" + k]) < 0.25, k      # B=1 train-mode BN: fp32 noise level
    assert rel_err(net.stage4[2].branches[0][3].bn2.running_mean, g["stage4.2.bn.rm"]) < 1e-4
    net.eval()
    with torch.no_grad():
        assert digest_err(digest(net(x), 64), g["y_eval"]) < 1e-4


def test_hrnet_wrapper_golden(golden):
    from scat_amd.models.hand_net import EncoderTransformerHRNet

    g = golden("hrnet")
    net = EncoderTransformerHRNet(opt_ns(), T(synth.mean_params(104, 61)))
    net.load_state_dict(synth.to_torch(synth.hrnet_wrapper_state(105, net.state_dict())), strict=True)
    net.cuda().train()
    random.seed(7)
    p = net(T(synth.images(106, 2)).cuda())
    assert rel_err(p, g["wrap:pred"]) < 1e-4
    p.square().sum().backward()
    assert digest_err(digest(net.regressor[0].weight.grad, 8), g["wrap:g:regressor.0.weight"]) < 1e-3
    assert digest_err(digest(net.mask_token.grad, 8), g["wrap:g:mask_token"]) < 5e-3
    assert digest_err(digest(net.transformer.layers[0][0].fn.to_qkv.weight.grad, 8),
                      g["wrap:g:transformer.layers.0.0.fn.to_qkv.weight"]) < 5e-3


def test_performer_block_golden(golden):
    """BASELINE config 5 block: performer_attn_block(49, 16) eval-mode fwd/bwd vs the reference module."""
    from scat_amd.models import vision_performer as P

    g = golden("performer")
    blk = P.performer_attn_block(49, 16)
    blk.load_state_dict(synth.to_torch(synth.performer_state(91, "")), strict=True)
    blk.cuda().eval()
    x = T(synth.normal_like(92, "x", (2, 21, 784), std=0.5)).cuda().requires_grad_(True)
    y = blk(x)
    assert digest_err(digest(y, 64), g["y"]) < 2e-5
    assert rel_err(y[0, :4, :16], g["y_head"]) < 2e-5
    (y * T(synth.normal_like(93, "cot", (2, 21, 784))).cuda()).sum().backward()
    assert digest_err(digest(x.grad, 64), g["dx"]) < 5e-5
    assert blk.w.grad is None
    for k, p in blk.named_parameters():
        if p.requires_grad:
            assert digest_err(digest(p.grad, 8), g["g:" + k]) < 1e-4, k


def test_performer_block_train_mode_values():
    """BASELINE configs[4] in TRAIN mode, value for value (VERDICT r03 "missing" 5; models/vision_performer.py:18,28: the
    two Dropout(0.1) of a block are active whenever the module trains).  The reference draws its masks from torch's RNG
    stream, which a device kernel cannot replay; the library's masks come from a counter hash seeded from python
    ``random`` (scat_dropout), and the oracle restates that hash (oracle.hash_dropout_mask) — so with the same ``random``
    seed both sides drop the SAME elements and the train-mode forward, input gradient and every parameter gradient can
    be compared at the eval-mode tolerances, in fp64 on the oracle side."""
    from scat_amd.models import vision_performer as P

    blk = P.performer_attn_block(49, 16)
    state = synth.to_torch(synth.performer_state(91, ""))
    blk.load_state_dict(state, strict=True)
    blk.cuda().train()
    x = T(synth.normal_like(92, "x", (2, 21, 784), std=0.5))
    cot = T(synth.normal_like(93, "cot", (2, 21, 784)))
    sd = {k: v.double().requires_grad_(k != "w") for k, v in state.items()}
    xr = x.double().requires_grad_(True)
    random.seed(77)
    yr = O.performer_block(sd, xr, "", 49, 16, dropout_p=0.1)
    (yr * cot.double()).sum().backward()
    xg = x.cuda().requires_grad_(True)
    random.seed(77)
    y = blk(xg)
    dropped = float((O.hash_dropout_mask(x.numel(), 0.1, 12345) == 0).double().mean())
    assert 0.08 < dropped < 0.12                              # Bernoulli(0.9) keep mask
    assert rel_err(y, yr.detach()) < 2e-5, rel_err(y, yr.detach())
    (y * cot.cuda()).sum().backward()
    assert rel_err(xg.grad, xr.grad) < 5e-5
    for k, p in blk.named_parameters():
        if p.requires_grad:
            assert rel_err(p.grad, sd[k].grad) < 1e-4, k
    # and the eval-mode path of the same oracle function is the golden-pinned one (dropout_p = 0 changes nothing)
    with torch.no_grad():
        assert torch.equal(O.performer_block({k: v.detach() for k, v in sd.items()}, x.double(), "", 49, 16),
                           O.performer_block({k: v.detach() for k, v in sd.items()}, x.double(), "", 49, 16, 0.0))


def test_vip_golden(golden):
    from scat_amd.models.vision_performer import ViP

    g = golden("performer")
    net = ViP(opt_ns(iteration=5), T(synth.mean_params(94, 10)), heads=16, emb_s=49)
    net.load_state_dict(synth.to_torch(synth.vip_state(95, net.state_dict())), strict=True)
    net.cuda().eval()
    p = net(T(synth.images(96, 2, 64)).cuda())
    assert rel_err(p, g["vip:pred"]) < 1e-4
    p.square().sum().backward()
    named = dict(net.named_parameters())
    for k in ("head.weight", "patch_emb.weight", "mains.0.kqv.weight", "cls_token", "pos_emb"):
        assert digest_err(digest(named[k].grad, 8), g["vip:g:" + k]) < 2e-3, k
    # train mode runs (dropout p = 0.1 with the hash mask) and keeps the expected scale
    net.train()
    random.seed(1)
    pt = net(T(synth.images(96, 2, 64)).cuda())
    assert torch.isfinite(pt).all() and rel_err(pt, g["vip:pred"]) < 0.5


def test_coarse_golden(golden):
    """train_coarse.py's network: 4-tuple incl. the last layer's attention map."""
    from scat_amd.models.hand_net import EncoderTransformerCoarse
    from scat_amd.trainer import pose_length_term, scat_loss

    g = golden("coarse")
    net = EncoderTransformerCoarse(opt_ns(), T(synth.mean_params(111)))
    net.load_state_dict(synth.to_torch(synth.fill_state(112, net.state_dict())), strict=True)
    net.cuda().train()
    random.seed(9)
    x, lab = T(synth.images(113, 2)).cuda(), T(synth.labels(114, 2)).cuda()
    pred, fv, attn, pl = net(x)
    assert tuple(attn.shape) == (2, 8, 21, 21)
    assert rel_err(pred, g["pred"]) < 1e-4
    assert digest_err(digest(attn, 64), g["attn"]) < 1e-4 and rel_err(attn[0, :2, :4, :8], g["attn_head"]) < 1e-4
    assert digest_err(digest(fv, 64), g["fv"]) < 1e-4
    assert digest_err(digest(pl, 64), g["pl"]) < 5e-4
    loss, _ = scat_loss(pred, lab)
    total = loss + 10 * pose_length_term(pl)
    assert abs(total.item() - float(g["loss"])) / abs(float(g["loss"])) < 1e-4
    total.backward()
    named = dict(net.named_parameters())
    for k in ("regressor.weight", "regressor.bias", "mask_token", "transformer.layers.0.1.norm.weight",
              "transformer.layers.2.2.net.2.weight", "transformer.layers.1.0.to_qkv.weight",
              "conv1x1_channel_reduction.weight"):
        assert digest_err(digest(named[k].grad, 8), g["g:" + k]) < 2e-3, k


def test_split_products_train_like_fp32_mfma():
    """The same 6 training steps with the two product modes (fp32 MFMA / three-term bf16 split): the first forward
    must agree to fp32 rounding, and the loss trajectories must stay together as far as two fp32 evaluation orders of
    this network do (gradient noise at 1e-2 relative, see the module docstring, moves later steps at ~1e-3)."""
    from scat_amd import ops
    from scat_amd.trainer import TrainStep

    traj = {}
    saved = ops.get_math_mode()
    try:
        for mode in (0, 1):
            ops.set_math_mode(mode)
            net = make_encoder(131)
            net.train()
            ts = TrainStep(net, lr=1e-4)
            random.seed(9)
            losses = []
            for step in range(6):
                x, lab = T(synth.images(140 + step, 8)).cuda(), T(synth.labels(150 + step, 8)).cuda()
                loss, parts, lpl, pred = ts(x, lab)
                losses.append(loss.item())
                if step == 0:
                    first = pred.detach().cpu()
            traj[mode] = (losses, first)
    finally:
        ops.set_math_mode(saved)
    (l0, p0), (l1, p1) = traj[0], traj[1]
    assert rel_err(p1, p0) < 2e-5                      # forward: both are fp32-accurate
    assert abs(l1[0] - l0[0]) / abs(l0[0]) < 2e-5
    # one Adam step later: as close as two fp32-MFMA summation orders of the same network are (mode 0 with and
    # without the packed stride-2 shortcut input, i.e. a different fp32 kernel for 3 convolutions: 6.5e-4 apart here)
    assert abs(l1[1] - l0[1]) / abs(l0[1]) < 2e-3
    for a, b in zip(l0[2:], l1[2:]):                   # then the run is chaotic (the loss swings 7x in 6 steps at this
        assert abs(a - b) / abs(a) < 3e-2, (l0, l1)    # init): measured 0.3 % .. 1.2 % apart, as two fp32 orders are


@pytest.mark.parametrize("math", [1, 0])
def test_bottleneck_golden(golden, math):
    """G3: ONE Bottleneck(64->64->256, downsample) taken out of the network, B=4, 8x8 — small enough for
    the gradients to be well conditioned, so the COMPOSED fused block (bn1/bn2 folded into operand loads, folded
    bn3 backward, sign mask, shortcut BatchNorm applied while adding, shortcut gradient accumulated on top of conv1's)
    is held to the real reference's outputs at 5e-5: forward, dx, every dW / dgamma / dbeta, running statistics, eval
    mode.  models/resnet.py:78-98."""
    from scat_amd import ops
    from scat_amd.models import resnet as R
    from scat_amd import nn as snn

    g = golden("bottleneck")
    saved = ops.get_math_mode()
    ops.set_math_mode(math)
    try:
        ds = torch.nn.Sequential(snn.Conv2d(64, 256, 1, 1, bias=False), snn.BatchNorm2d(256))
        blk = R.Bottleneck(64, 64, 1, ds)
        full = synth.to_torch(synth.resnet_state(31, "", (1, 0, 0, 0)))
        blk.load_state_dict({k[len("layer1.0."):]: v for k, v in full.items() if k.startswith("layer1.0.")}, strict=True)
        blk.cuda().train()
        x = T(synth.normal_like(32, "x", (4, 64, 8, 8))).cuda().requires_grad_(True)
        cot = T(synth.normal_like(33, "cot", (4, 256, 8, 8))).cuda()
        y = blk(x)
        assert digest_err(digest(y), g["y_train_d"]) < 2e-5
        assert rel_err(y.detach().cpu().numpy()[:, ::37], g["y_train"]) < 2e-5
        cot0 = cot.clone()
        (y * cot).sum().backward()
        assert torch.equal(cot, cot0)                      # the caller's gradient tensor is not modified in place
        assert digest_err(digest(x.grad), g["dx"]) < 5e-5
        assert rel_err(x.grad[0, :2], g["dx_head"]) < 5e-5
        for k, p in blk.named_parameters():
            assert digest_err(digest(p.grad), g["g:" + k]) < 5e-5, k
        for k, b in blk.named_buffers():
            assert rel_err(b.double(), g["buf:" + k]) < 5e-6, k
        blk.eval()
        with torch.no_grad():
            assert digest_err(digest(blk(x)), g["y_eval_d"]) < 2e-5
    finally:
        ops.set_math_mode(saved)


def test_layer4_standalone_on_7x7_maps():
    """``net.layer4(x)`` taken out of the network, train mode, forward + backward (models/resnet.py:78-98, 125-140): its
    block outputs are 7x7 maps, which the 1-bit sign mask cannot pack (H*W % 4 != 0) — the stand-alone block node
    must then keep the output for the ReLU sign of ``relu(bn3(.) + residual)`` instead of deriving it from bn3 alone
    (a wrong sign mask is an O(1) error).  Against the oracle's three Bottlenecks in fp64, with the
    oracle in fp32 as the yardstick: nine batch-statistics BatchNorms over 392 samples and the ReLUs between them make
    these gradients ill conditioned (the CPU fp32 evaluation itself is ~1 % from fp64), so the HIP gradients must be as
    close to fp64 as the CPU fp32 ones are (3x + 2e-4), the forward within 2e-5."""
    from scat_amd.models import resnet as R

    net = R.resnet50(pretrained=False, num_classes=512)
    full = synth.to_torch(synth.resnet_state(41, ""))
    net.load_state_dict(full, strict=True)
    layer4 = net.layer4.cuda().train()
    B = 8
    x = T(synth.normal_like(43, "x3", (B, 1024, 14, 14))).abs_()       # (a block input is a ReLU output)
    cot = T(synth.normal_like(44, "cot4", (B, 2048, 7, 7)))

    def oracle(dt, blocks, xin, prefix):
        sd = {k[len(prefix):]: (v.detach().clone().to(dt) if v.is_floating_point() else v.clone())
              for k, v in full.items() if k.startswith(prefix)}
        for k, v in sd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
        xr = xin.detach().clone().to(dt).requires_grad_(True)
        yr = xr
        for key, stride in blocks:
            yr = O.bottleneck(sd, key, yr, stride, True)
        (yr * cot.to(dt)).sum().backward()
        return yr.detach(), xr.grad, {k: v.grad for k, v in sd.items() if v.is_floating_point() and v.grad is not None}

    def held(got_x, got_p, r32, r64, what):
        e_h, e_c = rel_err(got_x, r64[1]), rel_err(r32[1], r64[1])
        assert e_h <= 3 * e_c + 2e-4, (what, "dx", e_h, e_c)
        eh = np.array([rel_err(got_p[k], r64[2][k]) for k in r64[2]])
        ec = np.array([rel_err(r32[2][k], r64[2][k]) for k in r64[2]])
        assert np.median(eh) <= 2 * np.median(ec) + 2e-4 and eh.max() <= 3 * ec.max() + 2e-4, (what, eh.max(), ec.max())

    blocks = (("0", 2), ("1", 1), ("2", 1))
    r32, r64 = oracle(torch.float32, blocks, x, "layer4."), oracle(torch.float64, blocks, x, "layer4.")
    xg = x.cuda().requires_grad_(True)
    y = layer4(xg)
    assert tuple(y.shape) == (B, 2048, 7, 7)
    assert rel_err(y, r64[0]) < 2e-5
    (y * cot.cuda()).sum().backward()
    held(xg.grad, {k: p.grad for k, p in layer4.named_parameters()}, r32, r64, "layer4")
    # one block alone, no shortcut convolution (the residual is the block input itself)
    blk = layer4[2]
    xin = y.detach().cpu()
    xb = xin.cuda().requires_grad_(True)
    for p in blk.parameters():
        p.grad = None
    yb = blk(xb)
    (yb * cot.cuda()).sum().backward()
    b32, b64 = oracle(torch.float32, (("2", 1),), xin, "layer4."), oracle(torch.float64, (("2", 1),), xin, "layer4.")
    assert rel_err(yb, b64[0]) < 2e-5
    held(xb.grad, {"2." + k: p.grad for k, p in blk.named_parameters()}, b32, b64, "layer4.2")


@pytest.mark.timeout(1700)
@pytest.mark.parametrize("key,cin,H,stride", [("layer1.0", 64, 56, 1), ("layer2.0", 256, 56, 2), ("layer3.1", 1024, 14, 1),
                                              ("layer4.2", 2048, 7, 1)])
def test_bottleneck_batch96_against_fp64(key, cin, H, stride):
    """The COMPOSED fused block at the benchmarked size (VERDICT r03 "Next" item 2; models/resnet.py:78-98): the block
    executor of the fused backbone (resnet._block_forward / _block_backward, what Bottleneck.forward runs) on one
    Bottleneck at batch 96, train mode — BatchNorm sums in the convolution epilogues and their finish, bn1 / bn2 folded
    into the next operand load, bn3 (+ the shortcut's BatchNorm) + residual + ReLU with the 1-bit sign mask, and in the
    backward bn_bwd_pre / the folded-BatchNorm gradient kernels (56x56 / 28x28 planes), the one-pass BatchNorm backward
    (14x14, 7x7), the packed stride-2 shortcut, accumulate-on-residual — against the oracle's Bottleneck in fp64 on the
    same tensors.  Blocks: layer1.0 (56x56, shortcut convolution, folded bn3 backward), layer2.0 (stride 2, packed
    shortcut), layer3.1 (14x14, one-pass BatchNorm backward), layer4.2 (7x7, no sign mask: the output is kept).

    How the gradient gate is made TIGHT.  At this size a block has 10-40 million ReLU inputs, a few dozen of them within
    one fp32 rounding of zero: any fp32 evaluation flips those against fp64 and every flip is an O(1) change of the
    gradient at that element — measured, the oracle itself in fp32 on the CPU is 4e-2 (max) / 1e-3 (L2) from fp64 on
    layer1.0's dx (gpurun_out/b96_block_parity.json).  That is the network's conditioning, not the kernels'.  So:
      1. forward against plain fp64: 2e-5; running statistics 1e-5;
      2. the three ReLU sign patterns of the HIP run (bn1 / bn2: sign of fma(c, scale, shift) exactly as the kernels
         form it; block output: its sign) may disagree with fp64's only where fp64's pre-activation is itself within
         2e-5 of zero (relative to its largest value), and in fewer than 2e-5 of the elements;
      3. every gradient (dx, dW, dgamma, dbeta) against the fp64 evaluation of the same block WITH THOSE sign patterns
         (relu(z) -> z * mask): no flips left, and the gate is 2e-4 norm-wise maximum — what a wrong reduction grid, a
         missing 1/N or a misplaced mask would exceed by orders of magnitude (the gate below is 5e-5: measured 2e-5 at worst)."""
    import json
    import os
    import torch.nn.functional as F
    from scat_amd.models import resnet as R

    B = 96
    net = R.resnet50(pretrained=False, num_classes=512)
    full = synth.to_torch(synth.resnet_state(51, ""))
    net.load_state_dict(full, strict=True)
    li, bi = key.split(".")
    blk = getattr(net, li)[int(bi)].cuda().train()
    x = T(synth.normal_like(52, "x" + key, (B, cin, H, H))).abs_()       # (a block input is a ReLU output)
    Ho = H // stride
    cot = T(synth.normal_like(53, "cot" + key, (B, blk.conv3.weight.shape[0], Ho, Ho)))

    # ---- HIP: the block executor itself (so that the tape's raw convolution outputs are at hand for the sign patterns)
    R._NBT.clear()
    rec = R._block_forward(blk, x.cuda(), True, None)
    torch._foreach_add_(R._NBT, 1)
    R._NBT.clear()
    _, _, c1, s1, c2, s2, c3, s3, cd, sd, out, omask = rec
    m1 = (c1.double() * s1.scale.double().view(1, -1, 1, 1) + s1.shift.double().view(1, -1, 1, 1) > 0).cpu()
    m2 = (c2.double() * s2.scale.double().view(1, -1, 1, 1) + s2.shift.double().view(1, -1, 1, 1) > 0).cpu()
    m3 = (out > 0).cpu()
    bc = R._Bwd(None, out.device, None)
    dx, _ = R._block_backward(bc, rec, cot.cuda().clone())
    bc.join()
    torch.cuda.synchronize()
    got = {"dx": dx}
    got.update({k: bc.grads[p] for k, p in blk.named_parameters()})

    # ---- fp64: plain (forward, statistics, its own sign patterns) ...
    def state():
        return {("b." + k[len(key) + 1:]): (v.detach().clone().double() if v.is_floating_point() else v.clone())
                for k, v in full.items() if k.startswith(key + ".")}

    sd64 = state()
    with torch.no_grad():
        z = {}

        def bn(k, t):
            return O.batch_norm(sd64, "b." + k, t, True)

        xd = x.double()
        z1 = bn("bn1", F.conv2d(xd, sd64["b.conv1.weight"]))
        z2 = bn("bn2", F.conv2d(F.relu(z1), sd64["b.conv2.weight"], stride=stride, padding=1))
        z3 = bn("bn3", F.conv2d(F.relu(z2), sd64["b.conv3.weight"]))
        res = xd
        if "b.downsample.0.weight" in sd64:
            res = bn("downsample.1", F.conv2d(xd, sd64["b.downsample.0.weight"], stride=stride))
        z3 = z3 + res
        y64 = F.relu(z3)
    assert rel_err(out, y64) < 2e-5, rel_err(out, y64)
    flips = {}
    for name, zz, m in (("bn1", z1, m1), ("bn2", z2, m2), ("out", z3, m3)):
        bad = (zz > 0) != m
        nbad = int(bad.sum())
        worst = float(zz[bad].abs().max() / zz.abs().max()) if nbad else 0.0
        flips[name] = {"disagreeing": nbad, "of": zz.numel(), "largest_fp64_preactivation_rel": worst}
        assert nbad <= 2e-5 * zz.numel() and worst < 2e-5, (key, name, flips[name])
    del z1, z2, z3, y64

    # ---- ... and with the HIP run's sign patterns in place of its own: the gradient reference
    sdm = state()
    leaves = {k[2:]: v.requires_grad_(True) for k, v in sdm.items() if v.is_floating_point() and "running" not in k}
    xr = x.double().requires_grad_(True)
    a1 = O.batch_norm(sdm, "b.bn1", F.conv2d(xr, sdm["b.conv1.weight"]), True) * m1
    a2 = O.batch_norm(sdm, "b.bn2", F.conv2d(a1, sdm["b.conv2.weight"], stride=stride, padding=1), True) * m2
    o3 = O.batch_norm(sdm, "b.bn3", F.conv2d(a2, sdm["b.conv3.weight"]), True)
    res = xr
    if "b.downsample.0.weight" in sdm:
        res = O.batch_norm(sdm, "b.downsample.1", F.conv2d(xr, sdm["b.downsample.0.weight"], stride=stride), True)
    ((o3 + res) * m3 * cot.double()).sum().backward()
    ref = {"dx": xr.grad}
    ref.update({k: v.grad for k, v in leaves.items()})

    rows = {k: rel_err(got[k], ref[k]) for k in ref}
    outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, "b96_block_parity.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[key] = {"forward_max": rel_err(out, F.relu(o3 + res).detach()), "sign_patterns": flips,
                "gradients_max_vs_fp64_with_the_same_signs": rows}
    json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    worst = max(rows, key=rows.get)
    assert rows[worst] < 5e-5, (key, worst, rows)       # (measured: 2e-5 at worst — bn1.bias of layer1.0; dx 6e-7 .. 1.2e-6)
    for k, v in blk.named_buffers():
        if "running" in k:
            assert rel_err(v, sd64["b." + k]) < 1e-5, (key, k)
        elif "num_batches_tracked" in k:
            assert int(v) == int(sd64["b." + k]) == 1


def test_config0_as_stated():
    """BASELINE configs[0] as it is written: reg_transformer, vit_heads 8, iteration 3, batch 8, a uint8[8,3,64,64]
    source batch -> the on-device input pipeline (normalise + bilinear resize to 224) -> one train
    step, against the CPU oracle fed the SAME resized fp32 input (the resize itself is pinned against torch's
    F.interpolate in test_preprocess_u8): joint offsets 1e-4, loss 1e-4, and Adam moves the weights."""
    from scat_amd import ops
    from scat_amd.trainer import TrainStep

    B = 8
    net = make_encoder(5)
    net.train()
    u8 = T(synth.randint_u8(70, "cfg0", (B, 3, 64, 64)))
    lab = T(synth.labels(71, B))
    xin = ops.preprocess_u8(u8.cuda())
    assert tuple(xin.shape) == (B, 3, 224, 224)
    sd = synth.to_torch(synth.encoder_transformer_state(5, 8))
    random.seed(11)
    with torch.no_grad():
        pr, fv = O.encoder_transformer_forward(sd, T(synth.mean_params(5)), xin.cpu(), pl_reg=False)
    l_ref, *_ = O.scat_loss(pr, lab, None)
    before = net.regressor.weight.detach().clone()
    ts = TrainStep(net, lr=5e-4)
    random.seed(11)
    total, parts, lpl, pred = ts(xin, lab.cuda())
    assert rel_err(pred[:, 3:66], pr[:, 3:66]) < 1e-4
    assert float(pred[:, 6:9].abs().max()) == 0.0
    assert abs(parts[0].item() - l_ref.item()) / abs(l_ref.item()) < 1e-4
    assert torch.isfinite(ts.buckets.flat_grad).all() and torch.isfinite(total)
    assert float((net.regressor.weight.detach() - before).abs().mean()) > 1e-4


def test_backbone_modules_standalone():
    """Every attribute of the ResNet mirror is a working module on its own, as in the reference (models/resnet.py:
    105-116, 142-162): the stem pieces, a whole nn.Sequential layer of Bottlenecks, AvgPool2d(7) — composed by hand
    they give ResNet.forward's outputs (the fused node), forward and input gradient."""
    from scat_amd.models import resnet as R

    net = R.resnet50()
    net.load_state_dict(synth.to_torch(synth.resnet_state(41, "")), strict=True)
    net.cuda().eval()
    x = T(synth.images(42, 2)).cuda()
    with torch.no_grad():
        feat, x1, x2, x3, x4 = net(x)
        h = net.maxpool(net.relu(net.bn1(net.conv1(x))))
        y1 = net.layer1(h)
        y2 = net.layer2(y1)
        y4 = net.layer4(net.layer3(y2))
        f = net.relu(net.fc1(net.relu(net.avgpool(y4).view(2, -1))))
    assert rel_err(y1, x1) < 1e-5 and rel_err(y2, x2) < 1e-5 and rel_err(y4, x4) < 1e-5 and rel_err(f, feat) < 1e-5
    # train mode, one layer: gradients through the per-block nodes == through the fused node's executor
    net.train()
    a = x1.detach().clone().requires_grad_(True)
    cot = T(synth.normal_like(43, "cot", tuple(x2.shape))).cuda()
    ya = net.layer2(a)
    (ya * cot).sum().backward()
    ga = {k: p.grad.clone() for k, p in net.layer2.named_parameters()}
    ref = torch.nn.functional.max_pool2d
    mp = ref(torch.relu(x1), 3, 2, 1)
    assert rel_err(net.maxpool(torch.relu(x1)), mp) == 0.0
    assert all(torch.isfinite(v).all() for v in ga.values()) and torch.isfinite(a.grad).all()
    with pytest.raises(RuntimeError):
        net.avgpool(x3)                                    # AvgPool2d(7) needs the 7x7 map, like the reference's fc1


def test_encoder_performer_config5():
    """BASELINE config 5 wiring (ResNet-50 tokens -> FAVOR+ blocks -> per-token offsets -> iterative regressor):
    the token path must be exactly performer_attn_block (golden-tested above against the reference's class) applied
    to EncoderTransformer's tokens.  Checked against a torch fp64 evaluation of the whole head on the device's own
    backbone outputs — tokens (1x1 reduction, +PE, mask-token scatter), three oracle performer blocks
    (models/vision_performer.py:34-68), the per-token Linear, the iteration-5 regressor loop and the root-relative
    shift (hand_net.py:379-393) — at 1e-4 on the 21-joint offsets; every trainable parameter must receive a finite
    gradient, and the head's gradients must match fp64 autograd of the same evaluation."""
    import torch.nn.functional as F

    from scat_amd.models.hand_net import EncoderPerformer

    torch.manual_seed(3)
    net = EncoderPerformer(opt_ns(vit_heads=16, iteration=5, pl_reg=False), T(synth.mean_params(3))).cuda().eval()
    x = T(synth.images(160, 2)).cuda()
    random.seed(11)
    pred, feat_visual = net(x)
    assert pred.shape == (2, 66) and feat_visual.shape == (2, 21, 28, 28)
    assert torch.isfinite(pred).all() and (pred[:, 6:9] == 0).all()          # root-relative: joint 1 is the origin

    def head_fp64(sd, main_feat, x2, midx):
        """the head in float64 from the backbone's outputs (differentiable w.r.t. sd's tensors)"""
        fv = F.conv2d(x2, sd["conv1x1_channel_reduction.weight"])
        tok = fv.view(2, 21, -1) + sd["positionalEncoding.pe"][0]
        tok = tok.clone()
        tok[:, midx, :] = sd["mask_token"][0, 0]
        for l in range(3):
            tok = O.performer_block(sd, tok, f"blocks.{l}.", 49, 16)
        off = F.linear(tok, sd["to_offsets.weight"], sd["to_offsets.bias"]).reshape(2, -1)
        p = T(synth.mean_params(3)).double().repeat(2, 1)
        p = torch.cat([p[:, :3], p[:, 3:] + off], dim=1)
        for _ in range(5):
            p = p + F.linear(torch.cat([main_feat, p], dim=1), sd["regressor.weight"], sd["regressor.bias"])
        j = p[:, 3:].view(2, 21, 3)
        return torch.cat([p[:, :3], (j - j[:, 1:2]).reshape(2, -1)], dim=1), fv

    with torch.no_grad():
        main_feat, _, x2, _, _ = net.main_encoder(x)
    random.seed(11)
    midx = O.mask_indices(0.2)
    sd = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
    ref, fv = head_fp64(sd, main_feat.double().cpu(), x2.double().cpu(), midx)
    assert rel_err(feat_visual, fv) < 2e-5
    assert rel_err(pred[:, 3:66], ref[:, 3:66]) < 1e-4, rel_err(pred[:, 3:66], ref[:, 3:66])
    assert rel_err(pred[:, :3], ref[:, :3]) < 1e-4

    net.train()
    for m in net.modules():                       # dropout off (vision_performer.py:18,28 are active in train mode):
        if m.__class__.__name__ == "Dropout":     # the comparison below wants the same function on both sides
            m.eval()
    random.seed(11)
    pred, _ = net(x)
    cot = T(synth.normal_like(161, "cot", (2, 66))).cuda()
    (pred * cot).sum().backward()
    for n, p in net.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
    assert net.blocks[0].w.grad is None                                      # frozen random features
    # head gradients against fp64 autograd of the same head on the (train-mode) backbone outputs
    with torch.no_grad():
        was = {n: b.clone() for n, b in net.main_encoder.named_buffers()}
        main_feat, _, x2, _, _ = net.main_encoder(x)       # batch statistics again: same outputs as in the step above
        for n, b in net.main_encoder.named_buffers():
            b.copy_(was[n])
    sd = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
    names = [n for n, p in net.named_parameters() if p.requires_grad and not n.startswith("main_encoder.")]
    for n in names:
        sd[n].requires_grad_(True)
    ref, _ = head_fp64(sd, main_feat.double().cpu(), x2.double().cpu(), midx)
    (ref * cot.double().cpu()).sum().backward()
    for n in names:
        g = dict(net.named_parameters())[n].grad
        assert rel_err(g, sd[n].grad) < 5e-4, (n, rel_err(g, sd[n].grad))


def test_prepared_weights_follow_training():
    """The backbone's prepared weights (ops.WeightPrep: one re-layout launch per step) must never go stale: two train
    steps (the optimiser's update is a raw kernel, invisible to autograd's version counters) and an inference forward
    in between / after give exactly the outputs of the same sequence with every convolution re-laying its own weights
    (ops.WPREP = False)."""
    from scat_amd import ops
    from scat_amd.trainer import TrainStep

    def sequence(use_prep):
        saved = ops.WPREP
        ops.WPREP = use_prep
        try:
            net = make_encoder(77)
            net.train()
            ts = TrainStep(net, lr=1e-3)
            outs = []
            xe = T(synth.images(171, 2)).cuda()
            for step in range(3):
                x, lab = T(synth.images(172 + step, 4)).cuda(), T(synth.labels(182 + step, 4)).cuda()
                random.seed(5 + step)
                loss, _, _, pred = ts(x, lab)
                outs.append(pred.detach().cpu())
                net.eval()
                with torch.no_grad():
                    for _ in range(2):            # second call: nothing changed, nothing is re-laid
                        random.seed(50 + step)
                        outs.append(net(xe)[0].cpu())
                net.train()
            if use_prep:
                wp = net.main_encoder._wprep
                assert len(wp.entries) > 100 and wp.table is not None and all(e[2] for e in wp.entries.values())
            return outs
        finally:
            ops.WPREP = saved

    a, b = sequence(True), sequence(False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_encoder_without_positional_table_aliases_feat_visual(golden):
    """hand_net.py:364-373 with ``pos_embed=False`` and masking on: the reference's ``feat`` is a view of
    ``feat_visual``, so the mask token lands in the returned map and the pose-length term is taken at the post-write
    tensor (non-zero in the masked channels).  Golden from the real reference (oracle/gen_golden.py::g_encoder_nope)."""
    g = golden("encoder_nope")
    net = make_encoder(51, pos_embed=False)
    net.train()
    x = T(synth.images(52, 2)).cuda()
    random.seed(3)
    pred, fv, pl = net(x)
    assert rel_err(pred, g["pred"]) < 1e-4
    assert digest_err(digest(fv, 64), g["fv"]) < 5e-5
    assert rel_err(fv.double().sum(dim=(0, 2, 3)), g["fv_chsum"]) < 5e-5
    random.seed(3)
    masked = list(range(21))
    random.shuffle(masked)
    for c in masked[:4]:
        assert torch.equal(fv[0, c].reshape(-1), net.mask_token.detach().reshape(-1)), c
        assert float(pl[:, c].abs().max()) > 0.0
    assert digest_err(digest(pl, 64), g["pl"]) < 1e-4
    assert rel_err(pl.double().abs().sum(dim=(0, 2, 3)), g["pl_chsum"]) < 1e-4
    assert not pl.requires_grad
    (pred * T(synth.normal_like(54, "cot", (2, 66))).cuda()).sum().backward()
    assert rel_err(net.mask_token.grad, g["g_full:mask_token"]) < 5e-4
    assert rel_err(net.conv1x1_channel_reduction.weight.grad, g["g_full:conv1x1_channel_reduction.weight"]) < 5e-4


def test_h3dw_encoder_golden(golden):
    """``H3DWEncoder`` (hand_net.py:28-58, imported by eval.py:26) against the real reference at batch 1."""
    from scat_amd.models.hand_net import H3DWEncoder

    g = golden("h3dw")
    mp = T(synth.normal_like(81, "mean61", (1, 61))) * 0.1
    net = H3DWEncoder(opt_ns(), mp)
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    sd = {k: v for k, v in synth.to_torch(synth.encoder_transformer_state(81, 8)).items()
          if k.startswith("main_encoder.")}
    for k, shp, s in (("feat_encoder.1.weight", (1024, 1024), 1024 ** -0.5), ("feat_encoder.1.bias", (1024,), 0.05),
                      ("regressor.0.weight", (61, 1085), 1085 ** -0.5), ("regressor.0.bias", (61,), 0.05)):
        sd[k] = T(synth.normal_like(82, k, shp)) * s
    net.load_state_dict(sd, strict=True)
    net.cuda().eval()
    x = T(synth.images(83, 1)).cuda()
    with torch.no_grad():
        feat, pred = net(x)
    assert rel_err(feat, g["eval:feat"]) < 1e-4 and rel_err(pred, g["eval:pred"]) < 1e-4
    net.train()
    net.main_encoder.eval()                      # head gradients on a frozen, eval-mode backbone (see gen_golden.g_h3dw)
    for p in net.main_encoder.parameters():
        p.requires_grad_(False)
    feat, pred = net(x)
    assert rel_err(pred, g["train:pred"]) < 1e-4
    (pred * T(synth.normal_like(84, "cot", (1, 61))).cuda()).sum().backward()
    named = dict(net.named_parameters())
    for k in ("feat_encoder.1.weight", "feat_encoder.1.bias", "regressor.0.weight", "regressor.0.bias"):
        assert digest_err(digest(named[k].grad, 16), g["g:" + k]) < 5e-4, k
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        net(torch.cat((x, x)))


def test_literal_train_py_sequence_through_dropin(golden):
    """The reference trainer's own step, statement for statement (train.py:124-125, 154-209), on the class the
    unmodified script would get: ``dropin/`` first on ``sys.path``, ``from models.hand_net import
    EncoderTransformer``, plain ``optim.Adam(net.parameters())``, torch-op projection / MSE / L1 / pose-length
    arithmetic, ``loss.backward()``, ``optimizer.step()`` — against golden G6 (the real reference net under the same
    sequence).  ``TrainStep``'s fused loss and flat-buffer Adam are NOT involved."""
    import os
    import sys
    import torch.nn as tnn
    import torch.optim as optim

    dropin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropin")
    sys.path.insert(0, dropin)
    try:
        for m in [m for m in sys.modules if m == "models" or m.startswith("models.")]:
            del sys.modules[m]
        from models.hand_net import EncoderTransformer   # train.py:13
    finally:
        sys.path.remove(dropin)
    g = golden("trainstep")
    opt = opt_ns()
    net = EncoderTransformer(opt, T(synth.mean_params(61))).cuda()                       # train.py:57
    net.load_state_dict(synth.to_torch(synth.encoder_transformer_state(61, 8)), strict=True)
    optimizer = optim.Adam(net.parameters(), lr=5e-4)                                   # train.py:60
    w3d, w2d = 100000.0, 10.0
    optimizer.zero_grad()
    optimizer.step()                                                                    # train.py:124-125
    random.seed(5)
    for step in (1, 2):
        inputs = T(synth.images(62 + step, 4)).cuda().float()
        labels = T(synth.labels(72 + step, 4)).cuda().float()
        optimizer.zero_grad()
        outputs, _, pl_term = net(inputs)
        cam = outputs[:, :3]
        j3d = outputs[:, 3:66].view(-1, 21, 3)
        cam_v = cam.view(-1, 1, 3)
        shifted = j3d[:, :, :2] + cam_v[:, :, 1:]
        j2d = (cam_v[:, :, 0] * shifted.view(shifted.size(0), -1)).view(shifted.size(0), shifted.size(1), -1)
        j2d = j2d * 112 + 112
        j3d, j2d = j3d.view(-1, 63), j2d.view(-1, 42)
        pl_lengths = torch.sum(torch.square(pl_term), dim=[2, 3]).mean(dim=[1]).sqrt()
        pl_mean = 0.0 + 0.01 * (torch.mean(pl_lengths) - 0.0)
        l_pl = torch.square(pl_lengths - pl_mean).mean()
        assert labels.size()[1] == 105
        l_3d = tnn.MSELoss()(j3d, labels[:, :63])
        l_2d = tnn.L1Loss()(j2d, labels[:, 63:])
        loss = w3d * l_3d + w2d * l_2d + 10 * l_pl
        loss.backward()
        optimizer.step()
        ref = g[f"s{step}:loss"]
        tol = 1e-4 if step == 1 else 3e-3      # (step 2 runs on Adam-updated weights: see test_trainstep_golden)
        assert abs(loss.item() - ref[0]) / abs(ref[0]) < tol, (step, loss.item(), ref)
        assert abs(l_3d.item() - ref[1]) / abs(ref[1]) < tol and abs(l_2d.item() - ref[2]) / abs(ref[2]) < tol
        assert abs(l_pl.item() - ref[3]) / abs(ref[3]) < 10 * tol
        assert rel_err(outputs, g[f"s{step}:pred"]) < (1e-4 if step == 1 else 2e-2)
        assert rel_err(net.regressor.bias, g[f"s{step}:regressor.bias"]) < 1e-3
        assert rel_err(net.main_encoder.bn1.running_mean, g[f"s{step}:bn1.running_mean"]) < 1e-4
        assert digest_err(digest(net.regressor.weight, 32), g[f"s{step}:regressor.weight"]) < (2e-3 if step == 1 else 5e-3)
        assert digest_err(digest(net.main_encoder.conv1.weight, 32), g[f"s{step}:conv1.weight"]) < (1e-4 if step == 1 else 5e-2)
    assert int(net.main_encoder.bn1.num_batches_tracked) == int(g["nbt"]) == 2


@pytest.mark.timeout(1200)
def test_backbone_gradients_against_fp64_with_the_runs_own_patterns():
    """The WHOLE fused ResNet-50 backbone (models/resnet.py:100-160: stem, max-pool, sixteen Bottlenecks, average pool, fc1),
    forward and backward through the real autograd node, held tight the way the single blocks are
    (test_bottleneck_batch96_against_fp64): the piecewise-linear pieces of the network — 49 ReLUs and the max-pool's
    arg-max — are taken from the HIP run (sign of fma(c, scale, shift) as the kernels form it, the sign of every block
    output, the pool's tap indices) and the oracle's layers are evaluated in fp64 WITH those patterns.  What is left is
    smooth, so every one of the 161 parameter gradients has to agree to rounding — against the goldens of the real
    reference the same gradients can only be held to 0.25, because one flipped ReLU of millions moves them by that much."""
    import torch.nn.functional as F
    from scat_amd import ops as OPS
    from scat_amd.models import resnet as R

    B = 12
    net = R.resnet50(pretrained=False, num_classes=512)
    full = synth.to_torch(synth.resnet_state(61, ""))
    net.load_state_dict(full, strict=True)
    net = net.cuda().train()
    x = T(synth.images(62, B))
    cot_f = T(synth.normal_like(63, "cot_feat", (B, 1024)))
    cot_2 = T(synth.normal_like(64, "cot_x2", (B, 512, 28, 28))) * 0.05

    recs, stem = [], {}
    bf, mp = R._block_forward, OPS.maxpool_fwd

    def bf_rec(*a, **k):
        rec = bf(*a, **k)
        recs.append(rec)
        return rec

    def mp_rec(c0, scale=None, shift=None, relu=False):
        y, idx = mp(c0, scale, shift, relu)
        stem.update(c0=c0, scale=scale, shift=shift, idx=idx)
        return y, idx

    R._block_forward, OPS.maxpool_fwd = bf_rec, mp_rec
    try:
        feat, x1, x2, x3, x4 = net(x.cuda())
        ((feat * cot_f.cuda()).sum() + (x2 * cot_2.cuda()).sum()).backward()
    finally:
        R._block_forward, OPS.maxpool_fwd = bf, mp
    torch.cuda.synchronize()
    assert len(recs) == 16 and stem
    got = {k: p.grad.detach().cpu().double() for k, p in net.named_parameters()}

    def fma_sign(c, s):
        return (c.double() * s.scale.double().view(1, -1, 1, 1) + s.shift.double().view(1, -1, 1, 1) > 0).cpu()

    # ---- the run's patterns
    m0 = (stem["c0"].double() * stem["scale"].double().view(1, -1, 1, 1) + stem["shift"].double().view(1, -1, 1, 1) > 0).cpu()
    idx = stem["idx"].cpu().long()                                   # tap kh * 3 + kw of the window's maximum
    OH, OW = idx.shape[2:]
    oy = torch.arange(OH).view(1, 1, OH, 1)
    ox = torch.arange(OW).view(1, 1, 1, OW)
    flat = ((2 * oy - 1 + idx // 3) * (2 * OW) + (2 * ox - 1 + idx % 3)).reshape(B, 64, -1)       # into the 112 x 112 plane
    masks = [(fma_sign(r[2], r[3]), fma_sign(r[4], r[5]), (r[10] > 0).cpu()) for r in recs]
    mpool = (x4.double().mean((2, 3)) > 0).cpu()
    mfeat = (feat > 0).cpu()

    # ---- the oracle's layers in fp64 with those patterns
    sd = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in full.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    xd = x.double()
    a0 = O.batch_norm(sd, "bn1", F.conv2d(xd, sd["conv1.weight"], stride=2, padding=3), True) * m0
    cur = a0.reshape(B, 64, -1).gather(2, flat).reshape(B, 64, OH, OW)
    feats, k = [], 0
    for li, (nblk, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), start=1):
        for bi in range(nblk):
            key, st = f"layer{li}.{bi}", (stride if bi == 0 else 1)
            m1, m2, m3 = masks[k]
            k += 1
            a1 = O.batch_norm(sd, key + ".bn1", F.conv2d(cur, sd[key + ".conv1.weight"]), True) * m1
            a2 = O.batch_norm(sd, key + ".bn2", F.conv2d(a1, sd[key + ".conv2.weight"], stride=st, padding=1), True) * m2
            o3 = O.batch_norm(sd, key + ".bn3", F.conv2d(a2, sd[key + ".conv3.weight"]), True)
            res = cur
            if key + ".downsample.0.weight" in sd:
                res = O.batch_norm(sd, key + ".downsample.1", F.conv2d(cur, sd[key + ".downsample.0.weight"], stride=st), True)
            cur = (o3 + res) * m3
        feats.append(cur)
    pooled = cur.mean((2, 3)) * mpool
    f64 = F.linear(pooled, sd["fc1.weight"], sd["fc1.bias"]) * mfeat
    fwd = {"feat": rel_err(feat, f64.detach()), "x2": rel_err(x2, feats[1].detach()), "x4": rel_err(x4, cur.detach())}
    print("forward against fp64 with the run's patterns:", fwd)
    # (fp32 rounding through 53 convolutions and BatchNorms whose 7 x 7 statistics see 588 samples per channel at this batch)
    assert fwd["x2"] < 2e-5 and fwd["x4"] < 3e-4 and fwd["feat"] < 3e-4, fwd
    ((f64 * cot_f.double()).sum() + (feats[1] * cot_2.double()).sum()).backward()
    rows = {name: rel_err(got[name], p.grad) for name, p in leaves.items()}
    worst = max(rows, key=rows.get)
    print("backbone gradients against fp64 with the run's patterns: worst", worst, rows[worst], "median",
          float(np.median(list(rows.values()))))
    assert set(rows) == set(got)
    assert rows[worst] < 2e-4, (worst, rows[worst], sorted(rows.items(), key=lambda kv: -kv[1])[:6])


@pytest.mark.timeout(1500)
def test_hrnet_gradients_against_fp64_with_the_runs_own_patterns():
    """HRNet-W32 (models/hrnet.py:150-261; BASELINE configs[3]'s backbone) the same way as the ResNet above: forward and
    backward through the real module tree (fused BasicBlocks, conv + BatchNorm + ReLU units, exchange units on their branch
    streams, layer1 on the Bottleneck executor), every ReLU's sign pattern taken from the HIP run — bn -> relu pairs from the
    raw convolution output and the scale / shift the kernels used, residual and exchange ReLUs from the stored outputs —
    and the oracle's layers (oracle/scat_oracle.py hrnet_forward, restated here with a mask where it has F.relu) evaluated in
    fp64 with them.  Every parameter gradient of the network then has to agree to rounding; against the reference's goldens
    the inner ones are held to 0.25 (test_hrnet_golden)."""
    import torch.nn.functional as F
    from scat_amd.models import hrnet as H
    from scat_amd.models import resnet as R

    B = 2
    net = H.HRNet(c=32, nof_joints=128, bn_momentum=0.1)
    full = synth.to_torch(synth.fill_state(111, net.state_dict()))
    net.load_state_dict(full, strict=True)
    net.cuda().train()
    mods = dict(net.named_modules())
    x = T(synth.images(112, B))
    cot = T(synth.normal_like(113, "cot", (B, 128, 56, 56)))

    bnrec, outrec = {}, {}

    class RecState(R._BNState):
        __slots__ = ()

        def __init__(self, c, bn, training):
            super().__init__(c, bn, training)
            bnrec[id(bn)] = (c, self.scale, self.shift)

    hooks = []
    for name, m in mods.items():
        if isinstance(m, (H.BasicBlock, H.Bottleneck, H.StageModule)):
            hooks.append(m.register_forward_hook(lambda mod, inp, out, name=name: outrec.__setitem__(name, out)))
    orig = R._BNState
    R._BNState = RecState
    try:
        y = net(x.cuda())
        (y * cot.cuda()).sum().backward()
    finally:
        R._BNState = orig
        for h in hooks:
            h.remove()
    torch.cuda.synchronize()
    got = {k: p.grad.detach().cpu().double() for k, p in net.named_parameters()}

    def bn_mask(kb):
        c, sc, sh = bnrec[id(mods[kb])]
        return (c.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) > 0).cpu()

    def out_mask(key, i=None):
        o = outrec[key]
        return ((o if i is None else o[i]) > 0).cpu()

    sd = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in full.items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}

    def cbn(kc, kb, t, stride=1, pad=1, relu=False):
        z = O.batch_norm(sd, kb, F.conv2d(t, sd[kc + ".weight"], stride=stride, padding=pad), True)
        return z * bn_mask(kb) if relu else z

    def basic(key, t):
        o = cbn(key + ".conv1", key + ".bn1", t, relu=True)
        o = cbn(key + ".conv2", key + ".bn2", o)
        return (o + t) * out_mask(key)

    def stage_module(key, xs, stage, out_branches):
        xs = list(xs)
        for i in range(stage):
            for b in range(4):
                xs[i] = basic(f"{key}.branches.{i}.{b}", xs[i])
        fused = []
        for i in range(out_branches):
            acc = None
            for j in range(stage):
                fk = f"{key}.fuse_layers.{i}.{j}"
                if i == j:
                    t = xs[j]
                elif i < j:
                    t = F.interpolate(cbn(fk + ".0", fk + ".1", xs[j], pad=0), scale_factor=float(2 ** (j - i)), mode="nearest")
                else:
                    t = xs[j]
                    for k in range(i - j):
                        t = cbn(f"{fk}.{k}.0", f"{fk}.{k}.1", t, stride=2, relu=(k < i - j - 1))
                acc = t if acc is None else acc + t
            fused.append(acc * out_mask(key, i))
        return fused

    t = cbn("conv1", "bn1", x.double(), stride=2, relu=True)
    t = cbn("conv2", "bn2", t, stride=2, relu=True)
    for b in range(4):
        k = f"layer1.{b}"
        o = cbn(k + ".conv1", k + ".bn1", t, pad=0, relu=True)
        o = cbn(k + ".conv2", k + ".bn2", o, relu=True)
        o = cbn(k + ".conv3", k + ".bn3", o, pad=0)
        res = cbn(k + ".downsample.0", k + ".downsample.1", t, pad=0) if (k + ".downsample.0.weight") in sd else t
        t = (o + res) * out_mask(k)
    xs = [cbn("transition1.0.0", "transition1.0.1", t, relu=True),
          cbn("transition1.1.0.0", "transition1.1.0.1", t, stride=2, relu=True)]
    xs = stage_module("stage2.0", xs, 2, 2)
    xs = [xs[0], xs[1], cbn("transition2.2.0.0", "transition2.2.0.1", xs[-1], stride=2, relu=True)]
    for m in range(4):
        xs = stage_module(f"stage3.{m}", xs, 3, 3)
    xs = [xs[0], xs[1], xs[2], cbn("transition3.3.0.0", "transition3.3.0.1", xs[-1], stride=2, relu=True)]
    xs = stage_module("stage4.0", xs, 4, 4)
    xs = stage_module("stage4.1", xs, 4, 4)
    xs = stage_module("stage4.2", xs, 4, 1)
    y64 = F.conv2d(xs[0], sd["final_layer.weight"], sd["final_layer.bias"])
    fwd = rel_err(y, y64.detach())
    (y64 * cot.double()).sum().backward()
    rows = {name: rel_err(got[name], p.grad) for name, p in leaves.items()}
    worst = max(rows, key=rows.get)
    print("HRNet-W32 against fp64 with the run's patterns: forward", fwd, "gradients: worst", worst, rows[worst], "median",
          float(np.median(list(rows.values()))), "of", len(rows))
    assert set(rows) == set(got)
    assert fwd < 1e-4, fwd
    assert rows[worst] < 3e-4, (worst, rows[worst], sorted(rows.items(), key=lambda kv: -kv[1])[:6])      # (measured 1.0e-4)


@pytest.mark.timeout(1700)
@pytest.mark.parametrize("B", [8, 96])
def test_train_step_gradients_against_fp64_with_the_runs_own_patterns(B):
    """One whole train.py iteration of BASELINE configs[1]'s network (train.py:152-209: forward with the mask-token draw, the
    loss, backward; at batch 8 and at the benchmarked batch 96, whose tile plans, split-K factors and epilogue reductions exist
    only at that size) — the real TrainStep: split backbone with the token path on its own stream, weight
    gradients on the side stream, flat buckets — against the oracle's EncoderTransformer in fp64 whose backbone is evaluated
    with the HIP run's ReLU sign patterns and max-pool taps (as in test_backbone_gradients_against_fp64_with_the_runs_own_patterns;
    the head has no piecewise-linear piece).  Prediction, loss and EVERY parameter gradient, head and backbone."""
    import torch.nn.functional as F
    from scat_amd import ops as OPS
    from scat_amd.models import resnet as R
    from scat_amd.trainer import TrainStep

    x, lab = T(synth.images(72, B)), T(synth.labels(73, B))
    net = make_encoder(1)
    net.train()
    ts = TrainStep(net, lr=5e-4)
    recs, stem, relus = [], {}, []
    bf, mp, rl = R._block_forward, OPS.maxpool_fwd, OPS.relu_fwd

    def bf_rec(*a, **k):
        rec = bf(*a, **k)
        recs.append(rec)
        return rec

    def mp_rec(c0, scale=None, shift=None, relu=False):
        y, idx = mp(c0, scale, shift, relu)
        stem.update(c0=c0, scale=scale, shift=shift, idx=idx)
        return y, idx

    def rl_rec(t):
        y = rl(t)
        relus.append(y)
        return y

    R._block_forward, OPS.maxpool_fwd, OPS.relu_fwd = bf_rec, mp_rec, rl_rec
    try:
        random.seed(3)
        total, parts, lpl, pred = ts(x.cuda(), lab.cuda())
    finally:
        R._block_forward, OPS.maxpool_fwd, OPS.relu_fwd = bf, mp, rl
    torch.cuda.synchronize()
    assert len(recs) == 16 and stem
    feat_hip = [t for t in relus if tuple(t.shape) == (B, 1024)][-1]
    g_hip = {k: p.grad.detach().cpu().double() for k, p in net.named_parameters() if p.grad is not None}

    def fma_sign(c, s):
        return (c.double() * s.scale.double().view(1, -1, 1, 1) + s.shift.double().view(1, -1, 1, 1) > 0).cpu()

    m0 = (stem["c0"].double() * stem["scale"].double().view(1, -1, 1, 1) + stem["shift"].double().view(1, -1, 1, 1) > 0).cpu()
    idx = stem["idx"].cpu().long()
    OH, OW = idx.shape[2:]
    oy, ox = torch.arange(OH).view(1, 1, OH, 1), torch.arange(OW).view(1, 1, 1, OW)
    flat = ((2 * oy - 1 + idx // 3) * (2 * OW) + (2 * ox - 1 + idx % 3)).reshape(B, 64, -1)
    masks = [(fma_sign(r[2], r[3]), fma_sign(r[4], r[5]), (r[10] > 0).cpu()) for r in recs]
    mpool = (recs[-1][10].double().mean((2, 3)) > 0).cpu()
    mfeat = (feat_hip > 0).cpu()

    def resnet_with_patterns(sd, xin, prefix="", training=True):
        p = prefix
        a0 = O.batch_norm(sd, p + "bn1", F.conv2d(xin, sd[p + "conv1.weight"], stride=2, padding=3), True) * m0
        cur = a0.reshape(B, 64, -1).gather(2, flat).reshape(B, 64, OH, OW)
        feats, k = [], 0
        for li, (nblk, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), start=1):
            for bi in range(nblk):
                key, st = f"{p}layer{li}.{bi}", (stride if bi == 0 else 1)
                m1, m2, m3 = masks[k]
                k += 1
                a1 = O.batch_norm(sd, key + ".bn1", F.conv2d(cur, sd[key + ".conv1.weight"]), True) * m1
                a2 = O.batch_norm(sd, key + ".bn2", F.conv2d(a1, sd[key + ".conv2.weight"], stride=st, padding=1), True) * m2
                o3 = O.batch_norm(sd, key + ".bn3", F.conv2d(a2, sd[key + ".conv3.weight"]), True)
                res = cur
                if key + ".downsample.0.weight" in sd:
                    res = O.batch_norm(sd, key + ".downsample.1",
                                       F.conv2d(cur, sd[key + ".downsample.0.weight"], stride=st), True)
                cur = (o3 + res) * m3
            feats.append(cur)
        f = F.linear(cur.mean((2, 3)) * mpool, sd[p + "fc1.weight"], sd[p + "fc1.bias"]) * mfeat
        return (f, *feats)

    sd = {k: (v.double() if v.dtype == torch.float32 else v)
          for k, v in synth.to_torch(synth.encoder_transformer_state(1, 8)).items()}
    params = O.trainable(sd)
    for p in params.values():
        p.requires_grad_(True)
    plain = O.resnet_forward
    O.resnet_forward = resnet_with_patterns
    try:
        random.seed(3)
        pr, fv, pl = O.encoder_transformer_forward(sd, T(synth.mean_params(1)).double(), x.double())
    finally:
        O.resnet_forward = plain
    loss, *_ = O.scat_loss(pr, lab.double(), pl)
    loss.backward()
    g64 = {k: p.grad for k, p in params.items() if p.grad is not None}
    assert rel_err(pred[:, 3:66], pr.detach()[:, 3:66]) < 2e-5
    assert abs(total.item() - loss.item()) / abs(loss.item()) < 2e-5
    assert set(g_hip) == set(g64)
    rows = {k: rel_err(g_hip[k], g64[k]) for k in g64}
    worst = max(rows, key=rows.get)
    head = {k: v for k, v in rows.items() if not k.startswith("main_encoder.")}
    print("train step against fp64 with the run's patterns: gradients worst", worst, rows[worst], "median",
          float(np.median(list(rows.values()))), "head worst", max(head.values()), "of", len(rows))
    assert rows[worst] < 5e-4, (worst, rows[worst], sorted(rows.items(), key=lambda kv: -kv[1])[:6])
