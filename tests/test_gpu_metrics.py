"""Evaluation metrics and checkpoint interop on the device."""
import os

import numpy as np
import pytest
import torch

from scat_amd import synth

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def test_metrics_on_device(golden):
    """MPJPE / Procrustes / PA-MPJPE / PCK / AUC / acceleration error computed on cuda tensors (no per-frame host loop)
    equal the reference's own functions' outputs (tests/golden/metrics.npz)."""
    from tests.test_metrics import check_metrics_against_reference

    check_metrics_against_reference(golden, "cuda")


def test_reference_checkpoint_round_trip(golden, tmp_path):
    """train.py:237-246 saves ``net.state_dict()`` with torch.save, eval.py:399-400 loads it STRICT.  A checkpoint
    written by the reference's own EncoderTransformer (tests/golden/ckpt_keys.npz holds its key order, shapes and a
    digest per tensor, made by oracle/gen_golden.py from the real module) must load strict into the mirror from a .pth
    file, give the golden eval-mode prediction, and the mirror's own checkpoint must round-trip the same way with the
    same keys in the same order."""
    import random

    from oracle.util import digest, rel_err
    from tests.test_gpu_model import make_encoder, opt_ns

    ck = golden("ckpt_keys")
    keys = [str(k) for k in ck["keys"]]
    sd = synth.to_torch(synth.encoder_transformer_state(51, 8))
    assert sorted(sd.keys()) == sorted(keys)
    sd = {k: sd[k].reshape(tuple(int(v) for v in shp if v >= 0)) for k, shp in zip(keys, ck["shapes"])}
    f = tmp_path / "reference_style.pth"
    torch.save(sd, f)                                      # what train.py:242 writes: the reference's order and shapes
    from scat_amd.models.hand_net import EncoderTransformer

    net = EncoderTransformer(opt_ns(), T(synth.mean_params(51)))
    missing = net.load_state_dict(torch.load(f, map_location="cpu"), strict=True)      # eval.py:399-400
    assert not missing.missing_keys and not missing.unexpected_keys
    net.cuda().eval()
    g = golden("encoder")
    random.seed(3)
    with torch.no_grad():
        pred = net(T(synth.images(52, 4)).cuda())[0]
    assert rel_err(pred, g["eval:pred"]) < 1e-4
    # the mirror's own checkpoint: same keys, same order, same shapes and values as the reference module's
    own = net.state_dict()
    assert list(own.keys()) == keys
    for k, shp, dg in zip(keys, ck["shapes"], ck["digests"]):
        assert tuple(own[k].shape) == tuple(int(v) for v in shp if v >= 0), k
        assert np.allclose(digest(own[k].float(), 4)[:4], dg, rtol=1e-6, atol=1e-9), k
    f2 = tmp_path / "mirror.pth"
    torch.save(own, f2)
    net2 = EncoderTransformer(opt_ns(), T(synth.mean_params(51))).cuda()   # (mean_params is a plain attribute, not in
    net2.load_state_dict(torch.load(f2, map_location="cuda"), strict=True)  # the checkpoint: hand_net.py:321)
    net2.eval()
    random.seed(3)
    with torch.no_grad():
        assert torch.equal(net2(T(synth.images(52, 4)).cuda())[0], pred)
