"""CPU-side checks of the C ABI: the header parses, the library loads and exports every declared symbol
with the declared arity, error codes come back as exceptions.  No kernel is launched (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from scat_amd import build

    return build.build(verbose=False)


def test_header_declares_the_path():
    from scat_amd._lib import parse_header

    protos = parse_header()
    need = {"scat_conv2d_fwd", "scat_conv2d_dgrad", "scat_conv2d_dgrad_s2", "scat_conv2d_wgrad", "scat_gemm",
            "scat_bn_train_stats", "scat_bn_apply", "scat_bn_bwd", "scat_maxpool3x3s2_fwd", "scat_maxpool3x3s2_bwd",
            "scat_avgpool_fwd", "scat_layernorm_fwd", "scat_layernorm_bwd", "scat_attention_fwd", "scat_attention_bwd",
            "scat_gelu_fwd", "scat_tokens_fwd", "scat_regressor_fwd", "scat_regressor_bwd", "scat_loss_fwd_bwd",
            "scat_adam", "scat_last_error", "scat_version"}
    assert need <= set(protos), need - set(protos)
    # every data-path entry point is stream-ordered: last argument is the stream
    for name, (rt, args) in protos.items():
        if rt is ctypes.c_int and name not in ("scat_version", "scat_check_device", "scat_get_math_mode",
                                                "scat_set_math_mode",
                                                # host-side state of the NEXT launch of this thread, no device work
                                                "scat_epilogue_stats_arm", "scat_epilogue_stats_arm_shift", "scat_epilogue_stats_groups",
                                                "scat_epilogue_bnb_arm", "scat_epilogue_bnb_groups",
                                                "scat_streamk_arm",
                                                # host-side record of reduces to be flushed later (the flush takes the stream)
                                                "scat_splitk_defer", "scat_splitk_reduce_pending", "scat_splitk_reduce_discard"):
            assert args[-1][1] == "stream", name
    # every prototype cites the reference file it replaces somewhere in the header
    src = open(os.path.join(ROOT, "include", "scat_hip.h")).read()
    assert len(re.findall(r"models/\w+\.py:\d+|train\.py:\d+|hand_net\.py:\d+", src)) >= 12


def test_library_exports_every_symbol(built):
    from scat_amd._lib import lib, parse_header

    L = lib()
    for name in parse_header():
        assert hasattr(L.cdll, name), name
    assert L.scat_version() >= 100
    assert isinstance(L.scat_last_kernel(), bytes)


def test_errors_surface_without_a_gpu(built):
    """argument validation happens before any HIP call, so it is testable on CPU"""
    from scat_amd._lib import ScatError, lib

    L = lib()
    with pytest.raises(ScatError, match="kernel 5x5 unsupported"):
        L.scat_conv2d_fwd(1, 1, 0, 1, 1, 3, 8, 8, 4, 5, 5, 1, 2, 0, 0, 0, 0)
    with pytest.raises(ScatError, match="null pointer"):
        L.scat_conv2d_fwd(0, 0, 0, 0, 1, 3, 8, 8, 4, 3, 3, 1, 1, 0, 0, 0, 0)
    with pytest.raises(ScatError, match="dim_head must be 64"):
        L.scat_attention_fwd(1, 1, 1, 2, 21, 8, 32, 0.1, 0)
    assert L.scat_conv2d_wgrad_ws(96, 64, 56, 56, 64, 3, 3, 1, 1) > 0
    assert L.scat_gemm_ws(2016, 1536, 784) >= 0


def test_product_has_no_cpu_fallback():
    import torch

    from scat_amd import ops
    from scat_amd._lib import ScatError

    with pytest.raises(ScatError, match="no CPU fallback"):
        ops.conv2d_fwd(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3), 1, 1)
    # and nothing under scat_amd imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "scat_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_module_api_mirrors_reference():
    """class names / ctor signatures / state_dict keys of the drop-in"""
    import inspect

    from scat_amd.models import hand_net, resnet, vision_transformer, vit

    assert list(inspect.signature(hand_net.EncoderTransformer.__init__).parameters) == ["self", "opt", "mean_params"]
    assert list(inspect.signature(vision_transformer.Transformer.__init__).parameters) == \
        ["self", "dim", "depth", "heads", "dim_head", "mlp_dim", "dropout"]
    assert list(inspect.signature(vit.Transformer.__init__).parameters) == \
        ["self", "dim", "depth", "heads", "dim_head", "mlp_dim", "dropout"]
    net = resnet.resnet50(pretrained=True, num_classes=512)
    from scat_amd import synth

    ref = synth.resnet_state(1)
    sd = net.state_dict()
    assert set(sd) == set(ref) and all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in ref)
    t = vision_transformer.Transformer(784, 3, 8, 64, 392)
    assert set(t.state_dict()) == set(synth.vt_state(1, ""))
    v = vit.Transformer(196, 3, 8, 64, 392, 0.0)
    assert set(v.state_dict()) == set(synth.vit_state(1, ""))


def test_synth_is_deterministic():
    import numpy as np

    from scat_amd import synth

    a = synth.normal_like(3, "w", (5, 7))
    b = synth.normal_like(3, "w", (5, 7))
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert not np.array_equal(a, synth.normal_like(4, "w", (5, 7)))
    assert abs(float(synth.normal_like(1, "big", (200000,)).std()) - 1.0) < 0.01
    u = synth.uniform(1, "u", (100000,), 2.0, 3.0)
    assert u.min() >= 2.0 and u.max() < 3.0


def test_bench_roofline_lookup_matches_profiles():
    """bench.py prices its roofline kernel's HBM traffic from profiles/r*_traffic.json (rocprofv3 --pmc passes over the
    bench command): the kernel labels the library reports must map onto the instantiation names that file is keyed by,
    or `roofline.traffic` silently becomes null."""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    traffic, name = bench.latest_traffic()
    assert name is not None and traffic, "no profiles/r*_traffic.json"
    for label in ("conv1x1_split_128x128x32", "conv1x1_split_128x128x32_tf", "conv3x3_split_64x128x16",
                  "wgrad1x1_pw_128x256x32_split124", "wgrad1x1_pw_256x128x32_tf_split32", "wgrad1x1_pw_128x128x32_tf_bnb_split64",
                  "wgrad3x3_rows_64x576x16_tf_r1_split64", "wgrad3x3_s2_split_pc128x128x16_tf_split86",
                  "conv7x7_s2_split_64x128x32"):
        inst = bench.instantiation_of(label, traffic)
        assert inst in traffic, (label, inst, sorted(traffic)[:8])
        assert traffic[inst]["hbm_bytes_per_launch"] > 0
