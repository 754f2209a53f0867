"""Deterministic, RNG-library-independent tensor synthesis.

Weights, inputs and labels for parity tests and the benchmark are regenerated
on every machine from (seed, tensor-name, element-index) by a counter hash, so
nothing but small expected outputs has to be stored in ``tests/golden``.
Pure numpy (uint64 arithmetic); the same code runs in the build container
(where the reference is importable) and on the GPU box (where it is not).

The parameter *shapes* follow the reference's ``state_dict`` layout
(models/resnet.py:103-140, models/hand_net.py:319-353,
models/vision_transformer.py:81-96); the *values* are ours (kaiming-like
scales so activations stay O(1) through 53 conv+BN layers).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wraps mod 2^64)."""
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def _stream(seed: int, name: str, n: int) -> np.ndarray:
    """n uint64 words for (seed, name)."""
    key = (np.uint64(zlib.crc32(name.encode())) << np.uint64(32)) ^ np.uint64(seed & 0xFFFFFFFF)
    base = _mix64(np.array([key], dtype=np.uint64))[0]
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _mix64(idx * np.uint64(0xD1342543DE82EF95) + base)


def uniform(seed: int, name: str, shape, lo=-1.0, hi=1.0) -> np.ndarray:
    """float32 uniform in [lo, hi) — 24 random mantissa bits per element."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_stream(seed, name, n) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal_like(seed: int, name: str, shape, std=1.0) -> np.ndarray:
    """Zero-mean, variance std^2, bounded (sum of 4 uniforms, Irwin-Hall)."""
    n = int(np.prod(shape)) if len(shape) else 1
    w = _stream(seed, name, n)
    acc = np.zeros(n, dtype=np.float64)
    for s in (0, 16, 32, 48):
        acc += ((w >> np.uint64(s)) & np.uint64(0xFFFF)).astype(np.float64)
    acc = (acc * (1.0 / 65536.0) - 2.0) * np.sqrt(3.0)  # var of sum of 4 U(0,1) = 1/3
    return (acc * std).astype(np.float32).reshape(shape)


def randint_u8(seed: int, name: str, shape) -> np.ndarray:
    n = int(np.prod(shape))
    return (_stream(seed, name, n) >> np.uint64(56)).astype(np.uint8).reshape(shape)


# --------------------------------------------------------------------------
# state_dict synthesis
# --------------------------------------------------------------------------

def _conv(sd, seed, key, cout, cin, k):
    fan_in = cin * k * k
    sd[key + ".weight"] = normal_like(seed, key + ".weight", (cout, cin, k, k), std=np.sqrt(2.0 / fan_in))


def _bn(sd, seed, key, c):
    sd[key + ".weight"] = uniform(seed, key + ".weight", (c,), 0.5, 1.5)
    sd[key + ".bias"] = uniform(seed, key + ".bias", (c,), -0.3, 0.3)
    sd[key + ".running_mean"] = uniform(seed, key + ".running_mean", (c,), -0.2, 0.2)
    sd[key + ".running_var"] = uniform(seed, key + ".running_var", (c,), 0.6, 1.4)
    sd[key + ".num_batches_tracked"] = np.array(0, dtype=np.int64)


def _linear(sd, seed, key, nout, nin, bias=True, gain=1.0):
    sd[key + ".weight"] = normal_like(seed, key + ".weight", (nout, nin), std=gain * np.sqrt(1.0 / nin))
    if bias:
        sd[key + ".bias"] = uniform(seed, key + ".bias", (nout,), -0.1, 0.1)


def _ln(sd, seed, key, c):
    sd[key + ".weight"] = uniform(seed, key + ".weight", (c,), 0.7, 1.3)
    sd[key + ".bias"] = uniform(seed, key + ".bias", (c,), -0.2, 0.2)


RESNET_LAYERS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}


def resnet_state(seed: int, prefix: str = "", layers=(3, 4, 6, 3), width: int = 64) -> "OrderedDict[str, np.ndarray]":
    """Bottleneck ResNet state_dict (keys of models/resnet.py:103-140)."""
    sd = OrderedDict()
    p = prefix
    _conv(sd, seed, p + "conv1", width, 3, 7)
    _bn(sd, seed, p + "bn1", width)
    inplanes = width
    for li, nblk in enumerate(layers):
        planes = width * (2 ** li)
        for bi in range(nblk):
            stride = 2 if (li > 0 and bi == 0) else 1
            k = f"{p}layer{li + 1}.{bi}"
            _conv(sd, seed, k + ".conv1", planes, inplanes, 1)
            _bn(sd, seed, k + ".bn1", planes)
            _conv(sd, seed, k + ".conv2", planes, planes, 3)
            _bn(sd, seed, k + ".bn2", planes)
            _conv(sd, seed, k + ".conv3", planes * 4, planes, 1)
            _bn(sd, seed, k + ".bn3", planes * 4)
            if bi == 0 and (stride != 1 or inplanes != planes * 4):
                _conv(sd, seed, k + ".downsample.0", planes * 4, inplanes, 1)
                _bn(sd, seed, k + ".downsample.1", planes * 4)
            inplanes = planes * 4
    _linear(sd, seed, p + "fc1", 1024, inplanes)
    return sd


def vt_state(seed: int, prefix: str, dim=784, depth=3, heads=8, dim_head=64) -> "OrderedDict[str, np.ndarray]":
    """Dim-halving transformer (models/vision_transformer.py:81-96)."""
    sd = OrderedDict()
    inner = heads * dim_head
    for l in range(depth):
        k = f"{prefix}layers.{l}"
        _ln(sd, seed, k + ".0.fn.norm", dim)
        _linear(sd, seed, k + ".0.fn.fn.to_qkv", inner * 3, dim, bias=False)
        _linear(sd, seed, k + ".0.fn.fn.to_out.0", dim, inner)
        hid = dim * 3 // 4
        if l == depth - 1:
            _linear(sd, seed, k + ".1.net.0", hid, dim)
            _linear(sd, seed, k + ".1.net.2", 3, hid, gain=0.05)
        else:
            _ln(sd, seed, k + ".1.norm", dim)
            _linear(sd, seed, k + ".1.fn.net.0", hid, dim)
            _linear(sd, seed, k + ".1.fn.net.2", dim // 2, hid)
            dim //= 2
    return sd


def vit_state(seed: int, prefix: str, dim=196, depth=3, heads=8, dim_head=64, mlp_dim=392) -> "OrderedDict[str, np.ndarray]":
    """Dim-preserving transformer (models/vit.py:71-84)."""
    sd = OrderedDict()
    inner = heads * dim_head
    for l in range(depth):
        k = f"{prefix}layers.{l}"
        _linear(sd, seed, k + ".0.fn.to_qkv", inner * 3, dim, bias=False)
        _linear(sd, seed, k + ".0.fn.to_out.0", dim, inner, gain=0.5)
        _linear(sd, seed, k + ".1.fn.net.0", mlp_dim, dim)
        _linear(sd, seed, k + ".1.fn.net.3", dim, mlp_dim, gain=0.5)
    return sd


def performer_state(seed: int, prefix: str, emb_s=49, head=16, kernel_ratio=0.5) -> "OrderedDict[str, np.ndarray]":
    """performer_attn_block (models/vision_performer.py:12-32)."""
    sd = OrderedDict()
    emb = emb_s * head
    m = int(emb_s * kernel_ratio)
    _linear(sd, seed, prefix + "kqv", 3 * emb_s, emb_s, gain=0.5)
    _linear(sd, seed, prefix + "proj", emb, emb)
    _ln(sd, seed, prefix + "ln1", emb)
    _ln(sd, seed, prefix + "ln2", emb)
    _linear(sd, seed, prefix + "mlp.0", 4 * emb, emb)
    _linear(sd, seed, prefix + "mlp.2", emb, 4 * emb)
    sd[prefix + "w"] = normal_like(seed, prefix + "w", (m, emb_s), std=0.5)
    return sd


def positional_encoding(d_model: int, max_len: int) -> np.ndarray:
    """Sinusoidal table, float32 arithmetic as models/hand_net.py:61-72."""
    import torch

    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).numpy()


def encoder_transformer_state(seed: int, heads=8, layers=(3, 4, 6, 3), width=64) -> "OrderedDict[str, np.ndarray]":
    """Full EncoderTransformer state_dict (models/hand_net.py:319-353); 356 entries for ResNet-50."""
    sd = OrderedDict()
    _conv(sd, seed, "conv1x1_channel_reduction", 21, width * 8, 1)
    sd.update(vt_state(seed, "transformer.", 784, 3, heads, 64))
    sd.update(resnet_state(seed, "main_encoder.", layers, width))
    sd["positionalEncoding.pe"] = positional_encoding(784, 21)
    sd["mask_token"] = normal_like(seed, "mask_token", (1, 1, 784), std=1.0)
    _linear(sd, seed, "regressor", 66, 1024 + 66, gain=0.3)
    return sd


def mean_params(seed: int, n: int = 66) -> np.ndarray:
    """[1,n]: camera scale 5,0,0 then a hand-sized template (train.py:103-110)."""
    mp = np.zeros((1, n), dtype=np.float32)
    mp[0, 0] = 5.0
    mp[0, 3:] = normal_like(seed, "mean_params", (n - 3,), std=0.03)
    return mp


def images(seed: int, batch: int, size: int = 224) -> np.ndarray:
    """Network input [B,3,size,size] in [-1,1] (dataset/load_STB.py:48-67 normalisation)."""
    u8 = randint_u8(seed, "images", (batch, 3, size, size))
    return (u8.astype(np.float32) / 127.5 - 1.0).astype(np.float32)


def labels(seed: int, batch: int) -> np.ndarray:
    """[B,105] = 63 root-relative 3-D joints (m) + 42 2-D joints (px) (dataset/load_STB.py:286-294)."""
    g3 = normal_like(seed, "gt3d", (batch, 21, 3), std=0.03)
    g3 = g3 - g3[:, 1:2, :]
    g2 = uniform(seed, "gt2d", (batch, 42), 0.0, 224.0)
    return np.concatenate([g3.reshape(batch, 63), g2], axis=1).astype(np.float32)


def to_torch(sd, device="cpu"):
    import torch

    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)).to(device)) for k, v in sd.items())


def fill_state(seed: int, template) -> "OrderedDict[str, np.ndarray]":
    """Deterministic values for ANY module's state_dict, keyed by entry name and shape (template: name ->
    tensor/array/shape).  Conv/Linear weights get fan-in scaling, BN/LN affine and running stats get
    non-trivial values; used where the key list is long (HRNet: 1.7k entries) and comes from the module."""
    sd = OrderedDict()
    for k, t in template.items():
        shape = tuple(t) if isinstance(t, (tuple, list)) else tuple(t.shape)
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            sd[k] = np.array(0, dtype=np.int64)
        elif leaf == "running_mean":
            sd[k] = uniform(seed, k, shape, -0.2, 0.2)
        elif leaf == "running_var":
            sd[k] = uniform(seed, k, shape, 0.6, 1.4)
        elif leaf == "pe":
            sd[k] = positional_encoding(shape[-1], shape[-2])
        elif len(shape) == 1 and leaf == "weight":       # BN / LN scale
            sd[k] = uniform(seed, k, shape, 0.5, 1.5)
        elif len(shape) == 1:                            # biases
            sd[k] = uniform(seed, k, shape, -0.2, 0.2)
        elif len(shape) == 4:                            # conv weight
            sd[k] = normal_like(seed, k, shape, std=np.sqrt(2.0 / (shape[1] * shape[2] * shape[3])))
        elif len(shape) == 2:                            # linear weight
            sd[k] = normal_like(seed, k, shape, std=np.sqrt(1.0 / shape[1]))
        else:                                            # tokens and the like
            sd[k] = normal_like(seed, k, shape, std=1.0)
    return sd


def hrnet_wrapper_state(seed: int, template) -> "OrderedDict[str, np.ndarray]":
    """fill_state for EncoderTransformerHRNet, with the token-producing conv scaled so tokens are O(1):
    the wrapper's transformer (models/vit.py) has no LayerNorm and softmax scale dim**-0.5, so with
    O(60) tokens its output moves 0.5 % under a 1e-6 input perturbation (measured) and no fp32
    implementation can be compared against another."""
    sd = fill_state(seed, template)
    sd["conv1x1_channel_reduction.weight"] = (sd["conv1x1_channel_reduction.weight"] * 0.02).astype(np.float32)
    return sd


def vip_state(seed: int, template) -> "OrderedDict[str, np.ndarray]":
    """fill_state for ViP with Linear weights scaled to the reference's own init regime (std 0.02-ish,
    vision_performer.py:93-100): FAVOR+ has no stabiliser in exp(w.x - |x|^2/2) and no eps on its
    denominator, so O(1) projections overflow to NaN gradients in the reference itself."""
    sd = fill_state(seed, template)
    for k, v in sd.items():
        if v.ndim == 2 and k.endswith("weight"):
            sd[k] = (v * 0.25).astype(np.float32)
    return sd
