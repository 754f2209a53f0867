"""The oracle (oracle/scat_oracle.py) against outputs of the REAL reference modules
(tests/golden/*.npz, made by oracle/gen_golden.py). CPU only.

Tolerance: 2e-6 norm-wise relative (same torch CPU kernels, different op graph ⇒
only summation-order noise); gradients through 53 BN layers 2e-5."""
import random
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import scat_oracle as O
from oracle.util import digest, digest_err, rel_err
from scat_amd import synth

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def test_state_dict_layout():
    sd = synth.encoder_transformer_state(1)
    assert len(sd) == 356
    n = sum(v.size for k, v in sd.items())
    assert n == 29470944  # SURVEY §5 checkpoint row
    ntrain = sum(v.size for k, v in O.trainable(sd).items())
    assert ntrain == 29401307


def test_vt_transformer(golden):
    g = golden("vt")
    sd = synth.to_torch(synth.vt_state(11, "", 784, 3, 8, 64))
    for p in sd.values():
        p.requires_grad_(True)
    x = T(synth.normal_like(12, "x", (2, 21, 784))).requires_grad_(True)
    y = O.vt_forward(sd, x, "", 3, 8, 64)
    assert rel_err(y, g["y"]) < 2e-6
    (y * T(synth.normal_like(13, "cot", (2, 21, 3)))).sum().backward()
    assert digest_err(digest(x.grad), g["dx"]) < 2e-5
    for k, p in sd.items():
        assert digest_err(digest(p.grad), g["g:" + k]) < 2e-5, k
    assert rel_err(sd["layers.2.1.net.2.weight"].grad, g["g_full:layers.2.1.net.2.weight"]) < 2e-5


@pytest.mark.parametrize("dim", [784, 392, 196])
def test_attention(golden, dim):
    g = golden("vt")
    sd = synth.to_torch(synth.vt_state(21, "", dim, 1, 8, 64))
    x = T(synth.normal_like(22, f"xa{dim}", (2, 21, dim)))
    y, attn = O.attention(x, sd["layers.0.0.fn.fn.to_qkv.weight"], sd["layers.0.0.fn.fn.to_out.0.weight"],
                          sd["layers.0.0.fn.fn.to_out.0.bias"], 8, 64 ** -0.5)
    assert digest_err(digest(y), g[f"attn{dim}:y"]) < 2e-6
    assert rel_err(y[0, :3, :8], g[f"attn{dim}:y_head"]) < 2e-5
    assert torch.allclose(attn.sum(-1), torch.ones(2, 8, 21), atol=1e-5)


def test_bottleneck(golden):
    g = golden("bottleneck")
    full = synth.to_torch(synth.resnet_state(31, "", (1, 0, 0, 0)))
    sd = OrderedDict((k, v) for k, v in full.items() if k.startswith("layer1.0."))
    for k, p in sd.items():
        if p.dtype == torch.float32 and "running" not in k:
            p.requires_grad_(True)
    x = T(synth.normal_like(32, "x", (4, 64, 8, 8))).requires_grad_(True)
    y = O.bottleneck(sd, "layer1.0", x, 1, True)
    assert digest_err(digest(y), g["y_train_d"]) < 2e-6
    (y * T(synth.normal_like(33, "cot", (4, 256, 8, 8)))).sum().backward()
    assert digest_err(digest(x.grad), g["dx"]) < 2e-5
    assert rel_err(x.grad[0, :2], g["dx_head"]) < 2e-5
    for k, p in sd.items():
        kk = k[len("layer1.0."):]
        if p.grad is not None:
            assert digest_err(digest(p.grad), g["g:" + kk]) < 2e-5, k
        elif "running" in k:
            assert rel_err(p, g["buf:" + kk]) < 2e-6, k
    ye = O.bottleneck(sd, "layer1.0", x, 1, False)
    assert digest_err(digest(ye), g["y_eval_d"]) < 2e-6


def test_resnet50(golden):
    g = golden("resnet50")
    sd = synth.to_torch(synth.resnet_state(41, ""))
    x = T(synth.images(42, 2))
    for mode in ("train", "eval"):
        with torch.no_grad():
            feat, *xs = O.resnet_forward(sd, x, "", mode == "train")
        assert rel_err(feat, g[f"{mode}:feat"]) < 2e-6
        for n, t in zip(("x1", "x2", "x3", "x4"), xs):
            assert digest_err(digest(t, 64), g[f"{mode}:{n}"]) < 2e-6
        if mode == "train":
            assert rel_err(sd["bn1.running_mean"], g["bn1.running_mean"]) < 2e-6
            assert rel_err(sd["layer4.2.bn3.running_var"], g["layer4.2.bn3.running_var"]) < 2e-6


def test_encoder_transformer(golden):
    g = golden("encoder")
    sd = synth.to_torch(synth.encoder_transformer_state(51, 8))
    params = O.trainable(sd)
    for p in params.values():
        p.requires_grad_(True)
    mp = T(synth.mean_params(51))
    x, lab = T(synth.images(52, 4)), T(synth.labels(53, 4))
    random.seed(3)
    pred, fv, pl = O.encoder_transformer_forward(sd, mp, x)
    assert rel_err(pred, g["pred"]) < 2e-6
    assert float(pred[:, 6:9].abs().max()) == 0.0  # root joint, hand_net.py:389-391
    assert digest_err(digest(fv, 64), g["fv"]) < 2e-6
    assert digest_err(digest(pl, 64), g["pl"]) < 2e-5
    loss, l3, l2, lpl = O.scat_loss(pred, lab, pl)
    assert rel_err(np.array([loss.item(), l3.item(), l2.item(), lpl.item()]), g["loss"]) < 2e-6
    assert abs(O.mpjpe_mm(pred.detach(), lab[:, :63]).item() - float(g["mpjpe"])) < 1e-3
    loss.backward()
    for k, p in params.items():
        assert digest_err(digest(p.grad, 8), g["g:" + k]) < 5e-5, k
    assert rel_err(params["regressor.weight"].grad, g["g_full:regressor.weight"]) < 2e-5


def test_dp_parity_definition(golden):
    """G9: two shards through the network independently from identical weights, gradients averaged, one
    Adam step — the reference's modules wrote tests/golden/dp.npz; the oracle must reproduce it (it is what
    tests/test_gpu_dp.py holds the N-replica HIP run to)."""
    g = golden("dp")
    grads = []
    for r in range(2):
        sd = synth.to_torch(synth.encoder_transformer_state(43, 8))
        params = O.trainable(sd)
        for p in params.values():
            p.requires_grad_(True)
        x, lab = T(synth.images(700 + r, 4)), T(synth.labels(710 + r, 4))
        random.seed(11)
        pred, fv, pl = O.encoder_transformer_forward(sd, T(synth.mean_params(43)), x)
        assert rel_err(pred, g[f"pred{r}"]) < 2e-6
        loss = O.scat_loss(pred, lab, pl)[0]
        assert abs(loss.item() - g["loss"][r]) / abs(g["loss"][r]) < 2e-6
        loss.backward()
        assert rel_err(sd["main_encoder.bn1.running_mean"], g[f"bn1.running_mean:{r}"]) < 2e-6
        grads.append({k: p.grad for k, p in params.items()})
    for k in grads[0]:
        m = (grads[0][k] + grads[1][k]) / 2
        assert digest_err(digest(m, 8), g["g:" + k]) < 5e-5, k
        if "g_full:" + k in g:
            assert rel_err(m, g["g_full:" + k]) < 5e-5, k


def test_trainstep(golden):
    g = golden("trainstep")
    sd = synth.to_torch(synth.encoder_transformer_state(61, 8))
    mp = T(synth.mean_params(61))
    st = {}
    random.seed(5)
    for step in (1, 2):
        x, lab = T(synth.images(62 + step, 4)), T(synth.labels(72 + step, 4))
        r = O.train_step(sd, mp, x, lab, st, step)
        assert rel_err(np.array([r["loss"].item(), r["l3d"].item(), r["l2d"].item(), r["lpl"].item()]),
                       g[f"s{step}:loss"]) < 5e-5
        assert rel_err(r["pred"], g[f"s{step}:pred"]) < 5e-5
        assert rel_err(sd["regressor.bias"], g[f"s{step}:regressor.bias"]) < 1e-5
        assert rel_err(sd["main_encoder.bn1.running_mean"], g[f"s{step}:bn1.running_mean"]) < 2e-6
        # Adam's first steps move every weight by ±lr regardless of |g|: sign noise on ~0 grads ⇒ compare loosely
        assert digest_err(digest(sd["main_encoder.layer3.0.conv2.weight"], 32), g[f"s{step}:layer3.0.conv2.weight"]) < 1e-3
    assert int(sd["main_encoder.bn1.num_batches_tracked"]) == int(g["nbt"]) == 2


def test_vit(golden):
    g = golden("vit")
    sd = synth.to_torch(synth.vit_state(81, ""))
    for p in sd.values():
        p.requires_grad_(True)
    x = T(synth.normal_like(82, "x", (2, 128, 196))).requires_grad_(True)
    y = O.vit_forward(sd, x, "", 3, 8)
    assert digest_err(digest(y, 64), g["y"]) < 2e-6
    (y * T(synth.normal_like(83, "cot", (2, 128, 196)))).sum().backward()
    assert digest_err(digest(x.grad, 64), g["dx"]) < 2e-5
    for k, p in sd.items():
        assert digest_err(digest(p.grad, 8), g["g:" + k]) < 2e-5, k


def test_performer(golden):
    g = golden("performer")
    sd = synth.to_torch(synth.performer_state(91, ""))
    for k, p in sd.items():
        if k != "w":
            p.requires_grad_(True)
    x = T(synth.normal_like(92, "x", (2, 21, 784), std=0.5)).requires_grad_(True)
    y = O.performer_block(sd, x, "", 49, 16)
    assert digest_err(digest(y, 64), g["y"]) < 2e-6
    (y * T(synth.normal_like(93, "cot", (2, 21, 784)))).sum().backward()
    assert digest_err(digest(x.grad, 64), g["dx"]) < 2e-5
    for k, p in sd.items():
        if p.grad is not None:
            assert digest_err(digest(p.grad, 8), g["g:" + k]) < 2e-5, k


def test_hrnet(golden):
    """HRNet(c=32,128) + the HRNet wrapper: oracle vs the reference modules' outputs."""
    from scat_amd.models import hrnet as H

    g = golden("hrnet")
    tmpl = {k: tuple(v.shape) for k, v in H.HRNet(c=32, nof_joints=128).state_dict().items()}
    sd = synth.to_torch(synth.fill_state(101, tmpl))
    for k, p in sd.items():
        if p.dtype == torch.float32 and "running" not in k:
            p.requires_grad_(True)
    x = T(synth.images(102, 1))
    y = O.hrnet_forward(sd, x, "", True)
    assert digest_err(digest(y, 64), g["y"]) < 5e-6
    assert rel_err(y[0, :4, :4, :8], g["y_head"]) < 5e-6
    (y * T(synth.normal_like(103, "cot", tuple(y.shape)))).sum().backward()
    for k in ("conv1.weight", "final_layer.weight", "final_layer.bias", "stage3.1.fuse_layers.0.2.0.weight",
              "transition2.2.0.0.weight"):
        assert digest_err(digest(sd[k].grad, 8), g["g:" + k]) < 1e-3, k
    assert rel_err(sd["stage4.2.branches.0.3.bn2.running_mean"], g["stage4.2.bn.rm"]) < 5e-6
    with torch.no_grad():
        assert digest_err(digest(O.hrnet_forward(sd, x, "", False), 64), g["y_eval"]) < 5e-6


def test_hrnet_wrapper(golden):
    from types import SimpleNamespace

    from scat_amd.models.hand_net import EncoderTransformerHRNet

    g = golden("hrnet")
    torch.Tensor.cuda, keep = (lambda self, *a, **k: self), torch.Tensor.cuda
    try:
        net = EncoderTransformerHRNet(SimpleNamespace(vit_heads=8, vit_depth=3, iteration=3, pos_embed=True,
                                                      mask_rate=0.2), T(synth.mean_params(104, 61)))
    finally:
        torch.Tensor.cuda = keep
    sd = synth.to_torch(synth.hrnet_wrapper_state(105, {k: tuple(v.shape) for k, v in net.state_dict().items()}))
    random.seed(7)
    p = O.encoder_transformer_hrnet_forward(sd, T(synth.mean_params(104, 61)), T(synth.images(106, 2)))
    assert rel_err(p, g["wrap:pred"]) < 5e-6


def test_vip(golden):
    from types import SimpleNamespace

    from scat_amd.models.vision_performer import ViP

    g = golden("performer")
    torch.Tensor.cuda, keep = (lambda self, *a, **k: self), torch.Tensor.cuda
    try:
        net = ViP(SimpleNamespace(iteration=5), T(synth.mean_params(94, 10)), heads=16, emb_s=49)
    finally:
        torch.Tensor.cuda = keep
    sd = synth.to_torch(synth.vip_state(95, {k: tuple(v.shape) for k, v in net.state_dict().items()}))
    p = O.vip_forward(sd, T(synth.mean_params(94, 10)), T(synth.images(96, 2, 64)), 16, 49, 3, 5)
    assert rel_err(p, g["vip:pred"]) < 5e-6


def test_coarse(golden):
    g = golden("coarse")
    from types import SimpleNamespace

    from scat_amd.models.hand_net import EncoderTransformerCoarse

    torch.Tensor.cuda, keep = (lambda self, *a, **k: self), torch.Tensor.cuda
    try:
        net = EncoderTransformerCoarse(SimpleNamespace(vit_heads=8, pl_reg=True, iteration=3, pos_embed=True,
                                                       mask_rate=0.2), T(synth.mean_params(111)))
    finally:
        torch.Tensor.cuda = keep
    sd = synth.to_torch(synth.fill_state(112, {k: tuple(v.shape) for k, v in net.state_dict().items()}))
    for p in O.trainable(sd).values():
        p.requires_grad_(True)
    random.seed(9)
    pred, fv, attn, pl = O.encoder_transformer_coarse_forward(sd, T(synth.mean_params(111)), T(synth.images(113, 2)))
    assert rel_err(pred, g["pred"]) < 5e-6
    assert digest_err(digest(attn, 64), g["attn"]) < 5e-6
    assert digest_err(digest(pl, 64), g["pl"]) < 5e-5


def test_oracle_feat_visual_aliasing_without_positional_table(golden):
    """hand_net.py:364-373, pos_embed=False + masking: the oracle must return the post-write ``feat_visual`` and the
    pose-length term taken there, like the real reference (golden encoder_nope)."""
    g = golden("encoder_nope")
    sd = synth.to_torch(synth.encoder_transformer_state(51, 8))
    for p in O.trainable(sd).values():
        p.requires_grad_(True)
    random.seed(3)
    pred, fv, pl = O.encoder_transformer_forward(sd, T(synth.mean_params(51)), T(synth.images(52, 2)),
                                                 pos_embed=False)
    assert rel_err(pred.detach(), g["pred"]) < 2e-6
    assert digest_err(digest(fv, 64), g["fv"]) < 2e-6
    assert digest_err(digest(pl, 64), g["pl"]) < 2e-5
    assert rel_err(pl.double().abs().sum(dim=(0, 2, 3)), g["pl_chsum"]) < 2e-5
