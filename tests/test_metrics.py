"""scat_amd.metrics against independent numpy restatements of eval.py's formulas (CPU)."""
import numpy as np
import torch

from scat_amd import metrics as M
from scat_amd import synth


def _np_procrustes(S1, S2):   # one sample, [N,3]
    mu1, mu2 = S1.mean(0), S2.mean(0)
    X1, X2 = S1 - mu1, S2 - mu2
    K = X1.T @ X2
    U, s, Vt = np.linalg.svd(K)
    Z = np.eye(3)
    Z[-1, -1] = np.sign(np.linalg.det(U @ Vt))
    R = Vt.T @ Z @ U.T
    scale = np.trace(R @ K) / (X1 ** 2).sum()
    return scale * (R @ S1.T).T + (mu2 - scale * R @ mu1)


def test_metrics_match_numpy():
    B = 6
    gt = synth.normal_like(1, "gt", (B, 21, 3), 0.05).astype(np.float64)
    # pred = rotated + scaled + shifted gt plus noise: PA-MPJPE must remove the similarity part
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    pred = 1.3 * gt @ R.T + 0.02 + synth.normal_like(2, "n", (B, 21, 3), 0.002)
    tp, tg = torch.from_numpy(pred), torch.from_numpy(gt)
    ref_mpjpe = np.linalg.norm(pred - gt, axis=-1).mean() * 1000
    assert abs(M.mpjpe_mm(tp, tg).item() - ref_mpjpe) < 1e-9 * ref_mpjpe + 1e-9
    ref_pa = np.mean([np.linalg.norm(_np_procrustes(pred[b], gt[b]) - gt[b], axis=-1).mean() for b in range(B)]) * 1000
    assert abs(M.pa_mpjpe_mm(tp, tg).item() - ref_pa) < 1e-6 * ref_pa
    assert ref_pa < 0.1 * ref_mpjpe     # the similarity transform really was removed
    # [B,66]-style inputs are accepted too
    p66 = torch.cat([torch.zeros(B, 3, dtype=torch.float64), tp.reshape(B, 63)], 1)
    assert abs(M.mpjpe_mm(p66, tg.reshape(B, 63)).item() - ref_mpjpe) < 1e-9
    rng = np.arange(0, 51, 5.0)
    d = np.linalg.norm(pred - gt, axis=-1).reshape(-1) * 1000
    ref_pck = np.array([100.0 * np.mean(d <= r) for r in rng])
    got = M.pck(tp, tg, rng).numpy()
    assert np.allclose(got, ref_pck)
    assert abs(M.auc(rng, got) - np.trapz(ref_pck, rng) / np.trapz(np.ones_like(rng), rng)) < 1e-5
    ag = gt[:-2] - 2 * gt[1:-1] + gt[2:]
    ap = pred[:-2] - 2 * pred[1:-1] + pred[2:]
    assert np.allclose(M.accel_error(tg, tp).numpy(), np.linalg.norm(ap - ag, axis=2).mean(1))


def test_warmup_schedule_and_pretrained_ingest(tmp_path):
    """train.py:61-63,134 — linear ramp over 15 epochs (stepped with epoch+1), flat afterwards; and a torchvision-style
    ResNet checkpoint loads into the backbone with only the classifier left over (models/resnet.py:192-195)."""
    from scat_amd.schedule import WarmupSchedule, load_pretrained_backbone, warmup_lr

    assert warmup_lr(5e-4, 1) == 5e-4 / 15 and warmup_lr(5e-4, 15) == 5e-4 and warmup_lr(5e-4, 40) == 5e-4
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=5e-4)
    sch = WarmupSchedule(opt, 15)
    lrs = []
    for epoch in range(20):
        sch.step(epoch + 1)
        lrs.append(opt.param_groups[0]["lr"])
    assert np.allclose(lrs[:15], 5e-4 * np.arange(1, 16) / 15) and np.allclose(lrs[15:], 5e-4)

    class Bb(torch.nn.Module):          # key layout of a ResNet: conv/bn match, the head is fc1 (not fc)
        def __init__(self):
            super().__init__()
            self.conv1 = torch.nn.Conv2d(3, 4, 3, bias=False)
            self.bn1 = torch.nn.BatchNorm2d(4)
            self.fc1 = torch.nn.Linear(4, 2)

    src = {"conv1.weight": torch.ones(4, 3, 3, 3), "bn1.weight": torch.full((4,), 2.0), "bn1.bias": torch.zeros(4),
           "bn1.running_mean": torch.zeros(4), "bn1.running_var": torch.ones(4), "fc.weight": torch.zeros(1000, 4),
           "fc.bias": torch.zeros(1000)}
    f = tmp_path / "resnet.pth"
    torch.save(src, f)
    bb = Bb()
    missing, unexpected = load_pretrained_backbone(bb, str(f))
    assert set(unexpected) == {"fc.weight", "fc.bias"} and {"fc1.weight", "fc1.bias"} <= set(missing)
    assert torch.equal(bb.conv1.weight.data, src["conv1.weight"]) and torch.equal(bb.bn1.weight.data, src["bn1.weight"])


def test_pretrained_flag_is_honoured_or_loud(tmp_path, monkeypatch):
    """models/resnet.py:192-195 loads ImageNet weights when pretrained=True.  The mirror takes them from a local file
    (SCAT_RESNET50_CKPT / SCAT_PRETRAINED_DIR / torch hub cache) and must WARN when there is none instead of silently
    training the backbone from its random initialisation."""
    import warnings

    from scat_amd.models import resnet

    monkeypatch.delenv("SCAT_RESNET50_CKPT", raising=False)
    monkeypatch.delenv("SCAT_PRETRAINED_DIR", raising=False)
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "nohub"))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        resnet.resnet50(pretrained=True)
    assert any("RANDOM initialisation" in str(x.message) for x in w)
    f = tmp_path / "resnet50.pth"
    torch.save({"conv1.weight": torch.full((64, 3, 7, 7), 0.25), "bn1.running_var": torch.full((64,), 3.0),
                "fc.weight": torch.zeros(1000, 2048)}, f)
    monkeypatch.setenv("SCAT_RESNET50_CKPT", str(f))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        net = resnet.resnet50(pretrained=True)
    assert not any("RANDOM" in str(x.message) for x in w)
    assert torch.all(net.conv1.weight == 0.25) and torch.all(net.bn1.running_var == 3.0)
    monkeypatch.delenv("SCAT_RESNET50_CKPT")
    monkeypatch.setenv("SCAT_PRETRAINED_DIR", str(tmp_path))
    assert resnet.pretrained_checkpoint("resnet50") == str(f)


def metric_inputs():
    """the seeded joints of tests/golden/metrics.npz (oracle/gen_golden.py::metric_inputs)"""
    B = 12
    gt = synth.normal_like(901, "gt", (B, 21, 3), 0.05)
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], dtype=np.float32)
    pred = (1.3 * gt @ R.T + 0.02 + synth.normal_like(902, "n", (B, 21, 3), 0.004)).astype(np.float32)
    return pred, gt


def check_metrics_against_reference(golden, device):
    """scat_amd.metrics on ``device`` tensors against the outputs of the REFERENCE's own functions (eval.py:110-161,
    300-340, 753; data_utils/eval_utils.py:6-48) stored in tests/golden/metrics.npz by oracle/gen_golden.py."""
    g = golden("metrics")
    pred, gt = metric_inputs()
    tp, tg = torch.from_numpy(pred).to(device), torch.from_numpy(gt).to(device)
    rel = lambda a, b: float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / np.abs(b).max())
    assert rel(M.mpjpe_mm(tp, tg).item(), g["mpjpe_mm"]) < 2e-6
    assert rel(M.procrustes_align(tp, tg).cpu().numpy(), g["pa_aligned"]) < 2e-5
    assert rel(M.pa_mpjpe_mm(tp, tg).item(), g["pa_mpjpe_mm"]) < 1e-4
    pck = M.pck(tp, tg, g["rnge"]).cpu().numpy()
    assert np.abs(pck - g["pck"]).max() < 1e-4
    assert abs(M.auc(g["rnge"], pck) - float(g["auc"])) < 1e-4
    assert np.abs(M.pck(M.procrustes_align(tp, tg), tg, g["rnge"]).cpu().numpy() - g["pck_pa"]).max() < 1e-4
    assert rel(M.accel_error(tg, tp).cpu().numpy(), g["accel_err"]) < 2e-5
    # 66-wide network outputs (3 camera + 63 joints) and 63-wide labels, as eval.py slices them
    p66 = torch.cat([torch.zeros(12, 3, device=device), tp.reshape(12, 63)], 1)
    assert rel(M.mpjpe_mm(p66, tg.reshape(12, 63)).item(), g["mpjpe_mm"]) < 2e-6


def test_metrics_reference_golden(golden):
    check_metrics_against_reference(golden, "cpu")


def test_checkpoint_format_matches_reference(golden, tmp_path):
    """The reference's checkpoint (torch.save(net.state_dict()), train.py:237-246) as captured from its own module in
    tests/golden/ckpt_keys.npz: the mirror's state_dict has the same keys in the same order with the same shapes, and
    loads a reference-style .pth strictly (eval.py:399-400) — on CPU tensors (construction needs no GPU)."""
    from types import SimpleNamespace

    from oracle.util import digest

    ck = golden("ckpt_keys")
    keys = [str(k) for k in ck["keys"]]
    sd = synth.to_torch(synth.encoder_transformer_state(51, 8))
    assert sorted(sd.keys()) == sorted(keys) and len(keys) == 356
    ref_sd = {}
    for k, shp, dg in zip(keys, ck["shapes"], ck["digests"]):
        shape = tuple(int(v) for v in shp if v >= 0)
        assert sd[k].numel() == int(np.prod(shape)), k      # (the synthesiser keeps num_batches_tracked as [1])
        ref_sd[k] = sd[k].reshape(shape)                     # exactly the reference's checkpoint: order, shapes
        assert np.allclose(digest(ref_sd[k].float(), 4)[:4], dg, rtol=1e-6, atol=1e-9), k
    f = tmp_path / "ref.pth"
    torch.save(ref_sd, f)
    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self          # the ctor places mean_params with .cuda() (hand_net.py:321)
    try:
        from scat_amd.models.hand_net import EncoderTransformer

        opt = SimpleNamespace(vit_heads=8, pl_reg=True, iteration=3, pos_embed=True, mask_rate=0.2, vit_depth=3)
        net = EncoderTransformer(opt, torch.from_numpy(synth.mean_params(51)))
    finally:
        torch.Tensor.cuda = saved_cuda
    r = net.load_state_dict(torch.load(f, map_location="cpu"), strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    own = net.state_dict()
    assert list(own.keys()) == keys
    assert all(tuple(own[k].shape) == tuple(ref_sd[k].shape) for k in keys)
