"""Thin tensor-level wrappers over the C ABI (no autograd here).

PyTorch is plumbing only: it owns device memory and the stream. Every function passes
``tensor.data_ptr()`` + explicit sizes to libscat_hip and returns torch tensors that view
caller-owned buffers. There is no fallback path: tensors must be fp32, contiguous, on the GPU.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _switches as _sw

from ._lib import ScatError, lib

_ws_cache = {}

# bench.py sets PROFILE = [] for one instrumented step: every contraction-engine call then appends
# (kernel label, algorithmic FLOPs, start event, end event) — HIP events on the launch stream.
PROFILE = None


def _prof(flops, fn, *args):
    if PROFILE is None:
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn(*args)
    e1.record()
    PROFILE.append((lib().scat_last_kernel().decode(), float(flops), e0, e1))
    return r


# the same for the HBM-bound BatchNorm passes: (entry point, algorithmic bytes, start event, end event)
PROFILE_HBM = None


def _prof_hbm(name, nbytes, fn, *args):
    if PROFILE_HBM is None:
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn(*args)
    e1.record()
    PROFILE_HBM.append((name, float(nbytes), e0, e1))
    return r


# torch.cuda.current_stream() builds a Stream object and walks the device-index helpers (8-10 us a call, 1 300 calls in
# an HRNet-W32 step: that configuration is bound by the host); the raw handle is what the C ABI wants anyway
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ScatError(f"scat_amd op needs contiguous fp32 GPU tensors, got {t.dtype} {t.device} "
                            f"contiguous={t.is_contiguous()} (no CPU fallback on the product path)")


_HAVE_GPU = torch.cuda.is_available()


def workspace(nbytes: int, device, slot: str = "default") -> torch.Tensor:
    """Caller-owned scratch, grown on demand, one buffer per (device, slot, current stream): reuse is stream-ordered,
    and work issued on another stream (weight gradients, the downsample branch) gets its own buffer."""
    key = (device if isinstance(device, str) else (device.type, device.index), slot, _stream() if _HAVE_GPU else 0)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


BN_ONEPASS = os.environ.get("SCAT_BN_ONEPASS", "1") != "0"   # (the switch is read by the library; here for the byte count)

# kinds of scat_wprep_jobs (include/scat_hip.h SCAT_WPREP_*)
WPREP_CONV1X1_FWD, WPREP_CONV1X1_DGRAD, WPREP_CONV3X3_FWD, WPREP_CONV3X3_DGRAD, WPREP_FWD_SPLIT, WPREP_DGRAD_S2 = range(6)
WPREP = _sw.ab("SCAT_WPREP", True)   # 0: every convolution re-lays its weights itself (A/B runs)


class WeightPrep:
    """Prepared weights of one network (include/scat_hip.h "prepared weights"): one persistent workspace per
    (weight, kind) and a device table of their re-layout jobs, so that the 2 x 53 small per-convolution launches of a
    ResNet-50 step become ONE launch at the start of the step.

    Entries register themselves the first time a convolution is called with this object as ``wp`` (that call still
    re-lays its own weights, into the persistent workspace); ``run()`` — called by the network at the start of its
    forward — re-lays every registered weight with one launch and marks the entries ready.  Between a weight update
    and the next ``run()`` the workspaces are stale, hence the contract: the owner calls ``run()`` first thing in
    every forward (training: always; inference: when a weight's version counter moved or a training forward
    happened since)."""

    def __init__(self):
        self.entries = {}          # (data_ptr, kind) -> [dims, buf, ready, weight]
        self.table = None          # (device uint8 tensor, njobs, nblocks)
        self.dirty = True          # weights may have changed since the last run()
        self.versions = None
        self.fresh = False         # run_early() already re-laid the weights the next run() would

    def __deepcopy__(self, memo):
        return WeightPrep()     # workspaces are keyed by the original parameters' addresses

    def slot(self, w, kind, Cout, Cin, KH, KW, pad, nbytes):
        """-> (workspace, w_ready) for this weight and kind; registers the pair on first use"""
        key = (w.data_ptr(), kind)
        dims = (Cout, Cin, KH, KW, pad, int(nbytes))
        e = self.entries.get(key)
        if e is None or e[0] != dims or e[1].device != w.device:
            e = [dims, torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=w.device), False, w]
            self.entries[key] = e
            self.table = None
        return e[1], int(e[2])

    def _build(self):
        L = lib()
        jb = int(L.scat_wprep_job_bytes())
        host = ctypes.create_string_buffer(4 * jb)
        nj = ctypes.c_int(0)
        blobs, blk = [], 0
        for (ptr, kind), e in self.entries.items():
            Cout, Cin, KH, KW, pad, nbytes = e[0]
            blk = L.scat_wprep_jobs(kind, ptr, _p(e[1]), e[1].numel(), Cout, Cin, KH, KW, pad, blk, host, 4,
                                    ctypes.byref(nj))
            if blk < 0:
                raise RuntimeError(f"scat_wprep_jobs failed ({blk}): {L.scat_last_error().decode(errors='replace')}")
            blobs.append(host.raw[: nj.value * jb])
        raw = b"".join(blobs)
        dev = next(iter(self.entries.values()))[1].device
        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.table = (t, len(raw) // jb, blk)

    def run_early(self):
        """The owner has just updated the weights and nothing will read the old re-layouts any more (end of a train
        step): re-lay them now — on whatever stream is current — so that the next run() has nothing to launch."""
        if not self.entries or not WPREP or lib().scat_get_math_mode() != 1 or self.table is None:
            return
        if any(e[3].data_ptr() != k[0] for k, e in self.entries.items()):
            return
        t, nj, nblk = self.table
        lib().scat_wprep_run(_p(t), nj, nblk, _stream())
        for e in self.entries.values():
            e[2] = True
        self.versions = sum(e[3]._version for e in self.entries.values())
        self.fresh = True

    def run(self, training):
        """re-lay every registered weight (one launch) if they may have changed; no-op outside split-product mode"""
        if not self.entries or not WPREP or lib().scat_get_math_mode() != 1:
            for e in self.entries.values():
                e[2] = False
            self.fresh = False
            return
        if self.fresh:
            self.fresh = False
            if (self.table is not None and all(e[3].data_ptr() == k[0] for k, e in self.entries.items())
                    and sum(e[3]._version for e in self.entries.values()) == self.versions):
                self.dirty = bool(training)
                return                   # run_early() did this step's work (nothing touched the weights since)
        dead = [k for k, e in self.entries.items() if e[3].data_ptr() != k[0]]   # the parameter moved (.to(), .cuda())
        for k in dead:
            del self.entries[k]
            self.table = None
        if not self.entries:
            return
        ver = sum(e[3]._version for e in self.entries.values())
        stale = training or self.dirty or ver != self.versions or self.table is None
        if self.table is None:
            self._build()
        if stale:
            t, nj, nblk = self.table
            lib().scat_wprep_run(_p(t), nj, nblk, _stream())
            for e in self.entries.values():
                e[2] = True
        self.versions = ver
        self.dirty = bool(training)       # a training forward is followed by a weight update we do not see


def _wp_ws(wp, w, kind, Cout, Cin, KH, KW, pad, nbytes, device):
    if wp is not None and WPREP and lib().scat_get_math_mode() == 1:
        return wp.slot(w, kind, Cout, Cin, KH, KW, pad, nbytes)
    return workspace(nbytes, device, "wt"), 0


def set_math_mode(mode: int) -> int:
    """0: fp32 MFMA products; 1: fp32 operands as three bf16 terms, six bf16 MFMA products, fp32 accumulate
    (include/scat_hip.h).  Returns the previous mode."""
    old = lib().scat_get_math_mode()
    lib().scat_set_math_mode(int(mode))
    return old


def get_math_mode() -> int:
    return lib().scat_get_math_mode()


# ---------------------------------------------------------------- convolution

def conv_out_hw(H, W, k, stride, pad):
    return (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1


STEM_SPLIT = _sw.ab("SCAT_STEM_SPLIT", True)   # the 7x7/s2 stem on split-operand products (0: fp32 engine)
HALO = _sw.ab("SCAT_HALO", True)   # 3x3/s1/p1 through the LDS-halo kernel (0: generic gather, for A/B runs)


def _halo_ok(KH, KW, stride, pad, csrc, W):
    return HALO and KH == 3 and KW == 3 and stride == 1 and pad == 1 and W <= 63


PW_MIN_C = _sw.ab_int("SCAT_PW_MIN_C", 512)
PW = _sw.ab("SCAT_PW", True)       # 1x1/s1 through the weights-in-registers kernel (0: generic gather)


def _pw_ok(KH, KW, stride, pad, csrc, *ts):
    # measured (tools/conv_bench.py, batch 96): ahead of the generic engine from 512 contraction channels up,
    # behind it on the short, store-bound contractions of layer1/layer2 (PW_MIN_C=0 forces it, for tests)
    # (with split-operand products it is ahead everywhere)
    return (PW and KH == 1 and KW == 1 and stride == 1 and pad == 0 and csrc % 16 == 0
            and (csrc >= PW_MIN_C or lib().scat_get_math_mode() == 1)
            and all(t is None or t.data_ptr() % 16 == 0 for t in ts))


# Persistent stream-K grid for the pointwise kernels (csrc/conv1x1.hip conv1x1_sk_kernel).  OFF by default: measured at
# batch 96 (profiles/r03_streamk.txt) it is 5-15 % SLOWER than one tile per workgroup on every ResNet-50 shape — the
# hardware's dynamic dispatch already balances the tiles, and what the half-empty last round loses is less than the extra
# prologues and the partial-tile exchange cost.  SCAT_SK=1 enables it (tests do, to keep the kernel honest).
STREAMK = os.environ.get("SCAT_SK", "0") != "0"
_sk_bufs = {}


def _sk_arm(device):
    """lend the next pointwise launch this stream's stream-K scratch (include/scat_hip.h scat_streamk_arm)"""
    if not STREAMK:
        return
    key = (device.type, device.index, _stream())
    buf = _sk_bufs.get(key)
    if buf is None:
        buf = _sk_bufs[key] = torch.zeros(int(lib().scat_streamk_bytes()), dtype=torch.uint8, device=device)
    lib().scat_streamk_arm(_p(buf), buf.numel())


def streamk_check():
    """synchronise and raise if any stream-K launch of this process gave up waiting for a partial tile (tests)"""
    for (_, _, st), buf in _sk_bufs.items():
        lib().scat_streamk_error(_p(buf), buf.numel(), st)


EPI_STATS = _sw.ab("SCAT_EPI_STATS", True)   # BatchNorm sums in the convolution epilogue (0: separate pass)


def conv2d_fwd(x, w, stride, pad, in_scale=None, in_shift=None, in_relu=False, bias=None, out=None, wp=None,
               stats=False, stats_shift=None):
    """wp: the network's WeightPrep (prepared weights), or None: the call re-lays its weights itself.
    stats: a training-mode BatchNorm follows — ask the kernel to leave the per-tile channel sums of its output behind
    (include/scat_hip.h scat_epilogue_stats_arm); ``y.scat_stats`` = (partials, groups) when it did, for
    ``bn_train_stats`` to finish without reading y back."""
    if not (stats and EPI_STATS and bias is None):
        return _conv2d_fwd(x, w, stride, pad, in_scale, in_shift, in_relu, bias, out, wp)
    B, _, H, W = x.shape
    Cout, _, KH, KW = w.shape
    OH, OW = conv_out_hw(H, W, KH, stride, pad)
    nbytes = Cout * ((B * OH * OW + 31) // 32 + 4) * 8      # a column group is >= 32 pixels; + the ragged last tile
    part = workspace(nbytes, x.device, "bnpart")
    # stats_shift[Cout] (optional): a per-channel reference the sums are taken about — the previous step's batch mean
    # (resnet._BNState keeps it): fp32 partial sums of x^2 cancel in E[x^2] - mean^2 when |mean| >> sigma
    if stats_shift is not None and not (stats_shift.device == x.device and stats_shift.dtype == torch.float32
                                        and stats_shift.is_contiguous() and stats_shift.numel() == Cout):
        stats_shift = None
    if stats_shift is not None:
        lib().scat_epilogue_stats_arm_shift(_p(part), nbytes, _p(stats_shift))
    else:
        lib().scat_epilogue_stats_arm(_p(part), nbytes)
    try:
        y = _conv2d_fwd(x, w, stride, pad, in_scale, in_shift, in_relu, bias, out, wp)
    finally:
        groups = lib().scat_epilogue_stats_groups()
    # the partials live in a recycled workspace: they are only good until the next armed convolution on this stream.
    # Protocol (same host thread): arm -> convolution -> groups() -> [bn_train_stats takes them]; any other armed
    # convolution in between bumps the generation and bn_train_stats falls back to the pass over y.
    _EPI_GEN[0] += 1
    y.scat_stats = (part, groups, _EPI_GEN[0], stats_shift) if groups > 0 else None
    return y


_EPI_GEN = [0]


def _conv2d_fwd(x, w, stride, pad, in_scale=None, in_shift=None, in_relu=False, bias=None, out=None, wp=None):
    _chk(x, w, in_scale, in_shift, bias)
    B, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    OH, OW = conv_out_hw(H, W, KH, stride, pad)
    y = out if out is not None else torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device)
    if _pw_ok(KH, KW, stride, pad, Cin, x, w, in_scale, in_shift):
        ws, rdy = _wp_ws(wp, w, WPREP_CONV1X1_FWD, Cout, Cin, 1, 1, 0, lib().scat_conv1x1_s1_ws(Cout, Cin), x.device)
        _sk_arm(x.device)
        _prof(2.0 * B * OH * OW * Cout * Cin, lib().scat_conv1x1_s1, _p(x), _p(w), _p(y), B, Cin, H * W, Cout, 0,
              _p(bias), _p(in_scale), _p(in_shift), int(in_relu), 0, _p(ws), ws.numel(), rdy, _stream())
        return y
    if (stride == 2 and KH in (1, 3) and KW == KH and Cin % 16 == 0 and lib().scat_get_math_mode() == 1
            and _sw.ab("SCAT_S2_SPLIT", True)):
        ws, rdy = _wp_ws(wp, w, WPREP_FWD_SPLIT, Cout, Cin, KH, KW, pad,
                         lib().scat_conv2d_fwd_split_ws(Cout, Cin, KH, KW), x.device)
        _prof(2.0 * B * OH * OW * Cout * Cin * KH * KW, lib().scat_conv2d_fwd_split, _p(x), _p(w), _p(bias), _p(y), B,
              Cin, H, W, Cout, KH, KW, stride, pad, _p(in_scale), _p(in_shift), int(in_relu), _p(ws), ws.numel(),
              rdy, _stream())
        return y
    if (KH == 7 and KW == 7 and stride == 2 and pad == 3 and Cin == 3 and bias is None and in_scale is None
            and lib().scat_get_math_mode() == 1 and STEM_SPLIT):
        ws = workspace(lib().scat_conv7x7_s2_fwd_split_ws(Cout), x.device, "stem")
        _prof(2.0 * B * OH * OW * Cout * Cin * KH * KW, lib().scat_conv7x7_s2_fwd_split, _p(x), _p(w), _p(y), B, H, W,
              Cout, _p(ws), ws.numel(), _stream())
        return y
    if _halo_ok(KH, KW, stride, pad, Cin, W) and bias is None:
        ws, rdy = _wp_ws(wp, w, WPREP_CONV3X3_FWD, Cout, Cin, 3, 3, 1, lib().scat_conv3x3_s1_ws(Cout, Cin), x.device)
        _prof(2.0 * B * OH * OW * Cout * Cin * 9, lib().scat_conv3x3_s1, _p(x), _p(w), _p(y), B, Cin, H, W, Cout, 0,
              _p(in_scale), _p(in_shift), int(in_relu), 0, _p(ws), ws.numel(), rdy, _stream())
        return y
    _prof(2.0 * B * OH * OW * Cout * Cin * KH * KW, lib().scat_conv2d_fwd, _p(x), _p(w), _p(bias), _p(y), B, Cin, H, W,
          Cout, KH, KW, stride, pad, _p(in_scale), _p(in_shift), int(in_relu), _stream())
    return y


# ---------------------------------------------------------------- activations as pre-split bf16 planes
class Planes:
    """An activation tensor [B, C, H, W] held as its three bf16 terms in the P8 layout (include/scat_hip.h:
    planes[p][n][c / 8][pixel][c % 8]); ``buf`` is the caller-owned uint8 storage."""
    __slots__ = ("buf", "shape")

    def __init__(self, buf, shape):
        self.buf, self.shape = buf, tuple(shape)

    @property
    def device(self):
        return self.buf.device

    def record_stream(self, s):
        self.buf.record_stream(s)

    def to_f32(self):
        """hi + mid + lo as an fp32 NCHW tensor (exact) — tests and debugging only, plain torch"""
        B, C, H, W = self.shape
        v = self.buf.view(torch.bfloat16).view(3, B, C // 8, H * W, 8).float()
        x = (v[2] + v[1]) + v[0]
        return x.permute(0, 1, 3, 2).reshape(B, C, H, W).contiguous()


def planes_from(x, scale=None, shift=None, relu=False, out=None):
    """split(relu?(x * scale + shift)) -> Planes: one HBM-bound pass (4 B read, 6 B written per element)"""
    _chk(x, scale, shift)
    B, C, H, W = x.shape
    nbytes = int(lib().scat_planes_bytes(B, C, H * W))
    buf = out if out is not None else torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _prof_hbm("planes_from_f32", 10.0 * x.numel(), lib().scat_planes_from_f32, _p(x), _p(buf), B, C, H * W, _p(scale),
              _p(shift), int(relu), _stream())
    return Planes(buf, (B, C, H, W))


def _planes_ok(C, *ts):
    return (lib().scat_get_math_mode() == 1 and C % 32 == 0 and all(t is None or t.data_ptr() % 16 == 0 for t in ts))


def conv1x1_planes(xp, w, transposed=False, bias=None, out=None, accumulate=False, wp=None, lds_stages=0):
    """scat_conv1x1_s1 with the activations given as Planes: forward (w = [Cout, Cin, 1, 1], xp has Cin channels) or,
    transposed, the data gradient (xp = planes of dy with Cout channels -> dx with Cin channels)"""
    _chk(w, bias, out)
    B, C, H, W = xp.shape
    Cout, Cin = w.shape[0], w.shape[1]
    M = Cin if transposed else Cout
    assert C == (Cout if transposed else Cin), (xp.shape, tuple(w.shape), transposed)
    y = out if out is not None else torch.empty((B, M, H, W), dtype=torch.float32, device=w.device)
    ws, rdy = _wp_ws(wp, w, WPREP_CONV1X1_DGRAD if transposed else WPREP_CONV1X1_FWD, Cout, Cin, 1, 1, 0,
                     lib().scat_conv1x1_s1_ws(M, C), w.device)
    _prof(2.0 * B * H * W * Cout * Cin, lib().scat_conv1x1_planes, _p(xp.buf), _p(w), _p(y), B, C, H * W, M,
          int(transposed), _p(bias), int(accumulate), _p(ws), ws.numel(), rdy, int(lds_stages), _stream())
    return y


def conv2d_wt(w, out=None):
    _chk(w)
    Cout, Cin, KH, KW = w.shape
    wt = out if out is not None else torch.empty((Cin, Cout * KH * KW), dtype=torch.float32, device=w.device)
    lib().scat_conv2d_wt(_p(w), _p(wt), Cout, Cin, KH, KW, _stream())
    return wt


def conv2d_dgrad(dy, wt, x_shape, w_shape, stride, pad, out=None, accumulate=False):
    _chk(dy, wt, out)
    B, Cin, H, W = x_shape
    Cout, _, KH, KW = w_shape
    dx = out if out is not None else torch.empty(x_shape, dtype=torch.float32, device=dy.device)
    OH, OW = conv_out_hw(H, W, KH, stride, pad)
    _prof(2.0 * B * OH * OW * Cout * Cin * KH * KW, lib().scat_conv2d_dgrad, _p(dy), _p(wt), _p(dx), B, Cin, H, W,
          Cout, KH, KW, stride, pad, int(accumulate), _stream())
    return dx


def conv2d_dgrad_w(dy, w, x_shape, stride, pad, out=None, accumulate=False, wp=None):
    """Data gradient from the ORIGINAL weights: picks the parity-decomposed stride-2 kernels when they apply,
    else transposes the weights and runs the generic gather."""
    _chk(dy, w, out)
    B, Cin, H, W = x_shape
    Cout, _, KH, KW = w.shape
    if stride == 2 and ((KH == 1 and pad == 0) or (KH == 3 and pad == 1)) and Cout % 4 == 0:
        dx = out if out is not None else torch.empty(x_shape, dtype=torch.float32, device=dy.device)
        need = lib().scat_conv2d_dgrad_s2_ws(Cin, Cout, KH, KW)
        if Cout % 16 == 0:      # (the split-operand classes; otherwise the fp32 engine re-lays per class)
            ws, rdy = _wp_ws(wp, w, WPREP_DGRAD_S2, Cout, Cin, KH, KW, pad, need, dy.device)
        else:
            ws, rdy = workspace(need, dy.device, "wt"), 0
        OH, OW = conv_out_hw(H, W, KH, stride, pad)
        _prof(2.0 * B * OH * OW * Cout * Cin * KH * KW, lib().scat_conv2d_dgrad_s2, _p(dy), _p(w), _p(dx), B, Cin, H,
              W, Cout, KH, KW, pad, int(accumulate), _p(ws), ws.numel(), rdy, _stream())
        return dx
    if _halo_ok(KH, KW, stride, pad, Cout, W):
        dx = out if out is not None else torch.empty(x_shape, dtype=torch.float32, device=dy.device)
        ws, rdy = _wp_ws(wp, w, WPREP_CONV3X3_DGRAD, Cout, Cin, 3, 3, 1, lib().scat_conv3x3_s1_ws(Cout, Cin), dy.device)
        _prof(2.0 * B * H * W * Cout * Cin * 9, lib().scat_conv3x3_s1, _p(dy), _p(w), _p(dx), B, Cin, H, W, Cout, 1,
              0, 0, 0, int(accumulate), _p(ws), ws.numel(), rdy, _stream())
        return dx
    if _pw_ok(KH, KW, stride, pad, Cout, dy, w, out):
        dx = out if out is not None else torch.empty(x_shape, dtype=torch.float32, device=dy.device)
        ws, rdy = _wp_ws(wp, w, WPREP_CONV1X1_DGRAD, Cout, Cin, 1, 1, 0, lib().scat_conv1x1_s1_ws(Cin, Cout), dy.device)
        _sk_arm(dy.device)
        _prof(2.0 * B * H * W * Cout * Cin, lib().scat_conv1x1_s1, _p(dy), _p(w), _p(dx), B, Cout, H * W, Cin, 1, 0, 0,
              0, 0, int(accumulate), _p(ws), ws.numel(), rdy, _stream())
        return dx
    wt = conv2d_wt(w, out=workspace(4 * w.numel(), dy.device, "wt")[: 4 * w.numel()].view(torch.float32)
                   .view(Cin, Cout * KH * KW))
    return conv2d_dgrad(dy, wt, x_shape, tuple(w.shape), stride, pad, out=out, accumulate=accumulate)


# ---- deferred split-K reduces (include/scat_hip.h): the fixed-order sums of a stage's weight gradients as ONE launch
WG_DEFER = _sw.ab("SCAT_WG_DEFER", False)   # measured: 22.78 vs 22.70 ms (profiles/r04_ab_defer.txt) — the immediate reduce reads its slabs hot from the caches
_WG_ARENA = {}          # (device, stream) -> [chunks, offset]: every deferred contraction keeps its slabs until the flush
_WG_DEFERRING = [False]


def wgrad_defer_reset():
    """start of a backward: forget reduces an aborted backward may have left behind, rewind the slab arenas"""
    lib().scat_splitk_reduce_discard()
    lib().scat_splitk_defer(0)
    _WG_DEFERRING[0] = False
    for a in _WG_ARENA.values():
        a[1] = 0


def wgrad_defer(on: bool):
    """reduces of the weight-gradient calls issued while this is on are recorded, not launched (flush: wgrad_flush)"""
    on = bool(on) and WG_DEFER and _HAVE_GPU
    _WG_DEFERRING[0] = on
    lib().scat_splitk_defer(int(on))


def _wgrad_workspace(nbytes, device, slot):
    if not _WG_DEFERRING[0]:
        return workspace(nbytes, device, slot)
    key = (device.type, device.index, _stream())
    a = _WG_ARENA.setdefault(key, [[], 0])
    nbytes = (int(nbytes) + 255) // 256 * 256
    chunks, off = a
    if not chunks or off + nbytes > chunks[-1].numel():
        # (the slabs recorded so far live in the older chunks: they stay allocated until the flush)
        chunks.append(torch.empty(max(nbytes, 256 << 20), dtype=torch.uint8, device=device))
        off = 0
    a[1] = off + nbytes
    return chunks[-1][off:off + nbytes]


def wgrad_flush():
    """one grouped launch (per 48) for every recorded reduce, on the CURRENT stream — the stream the contractions ran on;
    afterwards the arena of that stream is one chunk large enough for what the stage needed"""
    if lib().scat_splitk_reduce_pending() > 0:
        lib().scat_splitk_reduce_flush(_stream())
    for key, a in _WG_ARENA.items():
        if len(a[0]) > 1:
            total = sum(c.numel() for c in a[0])
            dev = a[0][0].device
            a[0] = []                                  # (freed blocks are reused stream-ordered behind the flush)
            if key[2] == _stream():
                a[0] = [torch.empty(total, dtype=torch.uint8, device=dev)]
        a[1] = 0


def conv2d_wgrad(dy, x, w_shape, stride, pad, in_scale=None, in_shift=None, in_relu=False, out=None,
                 ws_slot="default"):
    _chk(dy, x, in_scale, in_shift, out)
    B, Cin, H, W = x.shape
    Cout, _, KH, KW = w_shape
    dw = out if out is not None else torch.empty(w_shape, dtype=torch.float32, device=x.device)
    OH, OW = conv_out_hw(H, W, KH, stride, pad)
    # (the C entry also wants dy 16-byte aligned and 2*OW + 5 input columns in its LDS row: mirrored here so that
    # anything else falls through to the general engine instead of raising — ADVICE r02)
    if (KH == 7 and KW == 7 and stride == 2 and pad == 3 and Cin == 3 and Cout == 64 and in_scale is None
            and OW % 16 == 0 and OW <= 112 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
            and lib().scat_get_math_mode() == 1 and STEM_SPLIT):
        ws = _wgrad_workspace(lib().scat_conv7x7_s2_wgrad_split_ws(B, H, W), x.device, ws_slot)
        _prof(2.0 * B * OH * OW * Cout * Cin * KH * KW, lib().scat_conv7x7_s2_wgrad_split, _p(dy), _p(x), _p(dw), B, H, W,
              Cout, _p(ws), ws.numel(), _stream())
        return dw
    need = lib().scat_conv2d_wgrad_ws(B, Cin, H, W, Cout, KH, KW, stride, pad)
    ws = _wgrad_workspace(need, x.device, ws_slot)
    _prof(2.0 * B * OH * OW * Cout * Cin * KH * KW, lib().scat_conv2d_wgrad, _p(dy), _p(x), _p(dw), B, Cin, H, W,
          Cout, KH, KW, stride, pad, _p(in_scale), _p(in_shift), int(in_relu), _p(ws), ws.numel(), _stream())
    return dw


# ---------------------------------------------------------------- GEMM

def gemm(a, a_si, a_sk, b, b_sk, b_sj, c, c_si, c_sj, M, N, K, bias=None, bias_mode=0, accumulate=False):
    need = lib().scat_gemm_ws(M, N, K)
    ws = workspace(need, c.device) if need else None
    _prof(2.0 * M * N * K, lib().scat_gemm, _p(a), a_si, a_sk, _p(b), b_sk, b_sj, _p(c), c_si, c_sj, M, N, K, _p(bias),
          bias_mode, int(accumulate), _p(ws), ws.numel() if ws is not None else 0, _stream())
    return c


GEMM_SPLIT = _sw.ab("SCAT_GEMM_SPLIT", True)        # dense projections on split-operand products
# M*N*K below which the fp32 engine stays: measured (tools/vit_gemm_bench.py, M = 2016 tokens) the split kernel wins only
# on the largest projection (qkv of layer 0, 2016 x 1536 x 784: 59 vs 72 us); the smaller ones are occupancy-bound
# (49..768 workgroups on 256 CUs) and lose to the fp32 engine's split-K + 16-channel stages
GEMM_SPLIT_MIN = _sw.ab_int("SCAT_GEMM_SPLIT_MIN", 1 << 31)


def _gemm_split_ok(M, N, K):
    return (GEMM_SPLIT and lib().scat_get_math_mode() == 1 and M >= 128 and N >= 64 and K >= 64
            and M * N * K >= GEMM_SPLIT_MIN)


def gemm_split(a, a_transposed, b, c, M, N, K, bias_n=None, accumulate=False):
    """c[M,N] (+)= op(a)[M,K] @ b[K,N] (+ bias_n) on the pointwise split kernel (scat_gemm_split)"""
    ws = workspace(lib().scat_gemm_split_ws(M, K), c.device, "gs")
    _prof(2.0 * M * N * K, lib().scat_gemm_split, _p(a), int(a_transposed), _p(b), _p(c), M, N, K, _p(bias_n),
          int(accumulate), _p(ws), ws.numel(), _stream())
    return c


def linear_fwd(x2d, w, bias=None, out=None, accumulate=False):
    """y[M,N] (+)= x[M,K] @ w[N,K]^T + bias"""
    _chk(x2d, w, bias, out)
    M, K = x2d.shape
    N = w.shape[0]
    y = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=x2d.device)
    if _gemm_split_ok(M, N, K):
        wt = workspace(4 * K * N, w.device, "wT")[: 4 * K * N].view(torch.float32)
        lib().scat_transpose2d(_p(w), _p(wt), N, K, _stream())
        return gemm_split(x2d, 0, wt, y, M, N, K, bias, accumulate)
    return gemm(x2d, K, 1, w, 1, K, y, N, 1, M, N, K, bias, 2 if bias is not None else 0, accumulate)


def linear_dgrad(dy2d, w, out=None, accumulate=False):
    """dx[M,K] = dy[M,N] @ w[N,K]"""
    _chk(dy2d, w, out)
    M, N = dy2d.shape
    K = w.shape[1]
    dx = out if out is not None else torch.empty((M, K), dtype=torch.float32, device=dy2d.device)
    if _gemm_split_ok(M, K, N):
        return gemm_split(dy2d, 0, w, dx, M, K, N, None, accumulate)
    return gemm(dy2d, N, 1, w, K, 1, dx, K, 1, M, K, N, accumulate=accumulate)


def linear_wgrad(dy2d, x2d, out=None):
    """dw[N,K] = dy[M,N]^T @ x[M,K]"""
    _chk(dy2d, x2d, out)
    M, N = dy2d.shape
    K = x2d.shape[1]
    dw = out if out is not None else torch.empty((N, K), dtype=torch.float32, device=x2d.device)
    if _gemm_split_ok(N, K, M):
        return gemm_split(dy2d, 1, x2d, dw, N, K, M)
    return gemm(dy2d, 1, N, x2d, K, 1, dw, K, 1, N, K, M)


class _GemmProblem(ctypes.Structure):
    """include/scat_hip.h ScatGemmProblem"""
    _fields_ = [("a", ctypes.c_void_p), ("a_si", ctypes.c_int64), ("a_sk", ctypes.c_int64),
                ("b", ctypes.c_void_p), ("b_sk", ctypes.c_int64), ("b_sj", ctypes.c_int64),
                ("c", ctypes.c_void_p), ("c_si", ctypes.c_int64), ("c_sj", ctypes.c_int64),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int)]


GROUP_WGRAD = _sw.ab("SCAT_GROUP_WGRAD", True)   # the token mixer's weight gradients as one launch
GROUP_MAX = 16


def linear_wgrad_group(pairs):
    """[(dy2d[M,N_q], x2d[M,K_q]), ...] -> [dw_q[N_q,K_q] = dy_q^T @ x_q] with ONE launch (scat_gemm_group): the
    independent weight gradients of a token mixer's backward (same token count M for all), <= 16 per call."""
    outs = []
    for i0 in range(0, len(pairs), GROUP_MAX):
        chunk = pairs[i0:i0 + GROUP_MAX]
        arr = (_GemmProblem * len(chunk))()
        flops = 0.0
        for q, (dy, x) in enumerate(chunk):
            _chk(dy, x)
            M, N = dy.shape
            K = x.shape[1]
            dw = torch.empty((N, K), dtype=torch.float32, device=x.device)
            outs.append(dw)
            arr[q] = _GemmProblem(_p(dy), 1, N, _p(x), K, 1, _p(dw), K, 1, N, K, M)
            flops += 2.0 * M * N * K
        need = lib().scat_gemm_group_ws(arr, len(chunk))
        ws = workspace(need, chunk[0][0].device, "gg") if need else None
        _prof(flops, lib().scat_gemm_group, arr, len(chunk), _p(ws), ws.numel() if ws is not None else 0, _stream())
    return outs


def colsum(x2d, out=None, accumulate=False):
    _chk(x2d, out)
    rows, cols = x2d.shape
    o = out if out is not None else torch.empty((cols,), dtype=torch.float32, device=x2d.device)
    nws = lib().scat_colsum_ws(rows, cols)
    if nws > 0:          # tall: row slices over ~1024 workgroups, then the slices in index order
        ws = workspace(nws, x2d.device, "colsum")
        lib().scat_colsum_sliced(_p(x2d), _p(o), rows, cols, int(accumulate), _p(ws), ws.numel(), _stream())
    else:
        lib().scat_colsum(_p(x2d), _p(o), rows, cols, int(accumulate), _stream())
    return o


class _ColsumJob(ctypes.Structure):
    """include/scat_hip.h ScatColsumJob"""
    _fields_ = [("x", ctypes.c_void_p), ("out", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                ("accumulate", ctypes.c_int)]


def colsum_group(xs):
    """[x2d, ...] -> [column sums] with one launch per 16 (scat_colsum_group; bit-identical to colsum on each)"""
    outs = []
    for i0 in range(0, len(xs), GROUP_MAX):
        chunk = xs[i0:i0 + GROUP_MAX]
        if any(lib().scat_colsum_ws(*x.shape) > 0 for x in chunk):      # tall inputs keep their sliced two-launch form
            outs += [colsum(x) for x in chunk]
            continue
        arr = (_ColsumJob * len(chunk))()
        for q, x in enumerate(chunk):
            _chk(x)
            o = torch.empty((x.shape[1],), dtype=torch.float32, device=x.device)
            outs.append(o)
            arr[q] = _ColsumJob(_p(x), _p(o), x.shape[0], x.shape[1], 0)
        lib().scat_colsum_group(arr, len(chunk), _stream())
    return outs


# ---------------------------------------------------------------- BatchNorm

def bn_train_stats(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5):
    """-> (save_mean, save_invstd, scale, shift); running stats updated in place.  When x came out of
    ``conv2d_fwd(..., stats=True)`` with its channel sums, they are finished from those (no pass over x)."""
    _chk(x, gamma, beta, running_mean, running_var)
    B, C, H, W = x.shape
    o = torch.empty((4, C), dtype=torch.float32, device=x.device)
    st = getattr(x, "scat_stats", None)
    if st is not None and st[2] != _EPI_GEN[0]:
        st = x.scat_stats = None       # another convolution has used the workspace since: take the pass over x
    if st is not None:
        part, groups, _, sshift = st
        x.scat_stats = None            # one use: the workspace behind it is recycled by the next convolution
        if sshift is not None:
            _prof_hbm("bn_train_stats_partials", 8.0 * C * groups, lib().scat_bn_train_stats_partials_shifted, _p(part),
                      groups, _p(sshift), B, C, H * W, _p(gamma), _p(beta), _p(running_mean), _p(running_var), momentum,
                      eps, _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _stream())
        else:
            _prof_hbm("bn_train_stats_partials", 8.0 * C * groups, lib().scat_bn_train_stats_partials, _p(part), groups,
                      B, C, H * W, _p(gamma), _p(beta), _p(running_mean), _p(running_var), momentum, eps, _p(o[0]),
                      _p(o[1]), _p(o[2]), _p(o[3]), _stream())
        return o[0], o[1], o[2], o[3]
    ws = workspace(lib().scat_bn_ws(B, C, H * W), x.device)
    # algorithmic traffic: one read of x
    _prof_hbm("bn_train_stats", 4.0 * x.numel(), lib().scat_bn_train_stats, _p(x), B, C, H * W, _p(gamma), _p(beta),
              _p(running_mean), _p(running_var), momentum, eps, _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _p(ws),
              ws.numel(), _stream())
    return o[0], o[1], o[2], o[3]


def bn_eval_fold(gamma, beta, running_mean, running_var, eps=1e-5):
    _chk(gamma, beta, running_mean, running_var)
    C = gamma.numel()
    o = torch.empty((2, C), dtype=torch.float32, device=gamma.device)
    lib().scat_bn_eval_fold(_p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, C, _p(o[0]), _p(o[1]),
                            _stream())
    return o[0], o[1]


def bn_apply(x, scale, shift, residual=None, relu=False, out=None, want_mask=False, res_scale=None, res_shift=None):
    """want_mask: also return the sign mask of the output (uint8, one byte per 4 elements; None when the plane is not
    a multiple of 4) — what bn_bwd needs of y, 32x smaller."""
    _chk(x, scale, shift, residual, out, res_scale, res_shift)
    B, C, H, W = x.shape
    y = out if out is not None else torch.empty_like(x)
    mask = None
    if want_mask and (H * W) % 4 == 0 and all(t is None or t.data_ptr() % 16 == 0 for t in (x, y, residual)):
        mask = torch.empty(x.numel() // 4, dtype=torch.uint8, device=x.device)
    # algorithmic traffic: read x (+ residual), write y (+ 1 bit per element of mask)
    nb = x.numel() * (8.0 + (4.0 if residual is not None else 0.0) + (0.25 if mask is not None else 0.0))
    _prof_hbm("bn_apply" + ("_res" if residual is not None else ""), nb, lib().scat_bn_apply, _p(x), _p(scale), _p(shift),
              _p(residual), _p(res_scale), _p(res_shift), int(relu), _p(y), _p(mask), B, C, H * W, _stream())
    return (y, mask) if want_mask else y


def bn_bwd(dy, x, y_out, relu, scale, shift, save_mean, save_invstd, gamma, dgamma=None, dbeta=None, dx=None,
           dres=None, dres_accumulate=False, y_mask=None):
    """-> (dx, dgamma, dbeta); if dres is given, the masked gradient is written/accumulated there.  y_mask (from
    bn_apply(want_mask=True)) replaces y_out."""
    if y_mask is not None:
        y_out = None
    _chk(dy, x, y_out, scale, shift, save_mean, save_invstd, gamma, dgamma, dbeta, dx, dres)
    B, C, H, W = x.shape
    dx = dx if dx is not None else torch.empty_like(x)
    dgamma = dgamma if dgamma is not None else torch.empty_like(gamma)
    dbeta = dbeta if dbeta is not None else torch.empty_like(gamma)
    ws = workspace(lib().scat_bn_ws(B, C, H * W), x.device)
    # algorithmic traffic of the two passes (reduce: dy, x; apply: dy, x -> dx [, masked gradient]); the block-output
    # sign comes from y_out (4 B) or its mask (1 bit) in both passes
    sign = 8.0 if y_out is not None else (0.0625 if y_mask is not None else 0.0)
    nb = x.numel() * (8.0 + 12.0 + sign + (4.0 if dres is not None and dres is not dy else 0.0))
    # small planes take the one-launch form (csrc/norm.hip bn_bwd_onepass_kernel): every tensor is read once
    if BN_ONEPASS and C >= 64 and B * H * W // (4 if (H * W) % 4 == 0 else 1) <= 8192:
        nb = x.numel() * (12.0 + sign / 2 + (4.0 if dres is not None and dres is not dy else 0.0))
    _prof_hbm("bn_bwd" + ("_res" if dres is not None else ""), nb, lib().scat_bn_bwd, _p(dy), _p(x), _p(y_out),
              _p(y_mask), int(relu), _p(scale), _p(shift), _p(save_mean), _p(save_invstd), _p(gamma), _p(dgamma),
              _p(dbeta), _p(dx), _p(dres), int(dres_accumulate), B, C, H * W, _p(ws), ws.numel(), _stream())
    return dx, dgamma, dbeta


def bn_bwd_maxpool(dy_pooled, idx, x, relu, scale, shift, save_mean, save_invstd, gamma, dgamma=None, dbeta=None):
    """BatchNorm(+ReLU) backward of the gradient a 3x3 / stride-2 max-pool's backward would scatter from ``dy_pooled`` with
    its arg-max taps ``idx`` (the stem: conv1 -> bn1 -> relu -> maxpool), without materialising that gradient.
    -> (dx[B,C,H,W], dgamma, dbeta)"""
    _chk(dy_pooled, x, scale, shift, save_mean, save_invstd, gamma, dgamma, dbeta)
    B, C, H, W = x.shape
    assert tuple(dy_pooled.shape) == (B, C, H // 2, W // 2) and idx.shape == dy_pooled.shape and idx.dtype == torch.int8
    dx = torch.empty_like(x)
    dgamma = dgamma if dgamma is not None else torch.empty_like(gamma)
    dbeta = dbeta if dbeta is not None else torch.empty_like(gamma)
    ws = workspace(lib().scat_bn_ws(B, C, H * W), x.device)
    # algorithmic traffic: the pooled gradient + taps twice, a gather of x at the arg-max pixels, x once, dx once
    nb = dy_pooled.numel() * (2 * 5.0 + 4.0) + x.numel() * 8.0
    _prof_hbm("bn_bwd_maxpool", nb, lib().scat_bn_bwd_maxpool, _p(dy_pooled), _p(idx), _p(x), int(relu), _p(scale), _p(shift),
              _p(save_mean), _p(save_invstd), _p(gamma), _p(dgamma), _p(dbeta), _p(dx), B, C, H, W, _p(ws), ws.numel(),
              _stream())
    return dx, dgamma, dbeta


def bn_bwd_pre(dy, x, relu, scale, shift, save_mean, save_invstd, gamma, dgamma=None, dbeta=None, y_out=None,
               y_mask=None, dy_add=None):
    """First half of a BatchNorm backward: dy becomes the masked gradient g IN PLACE; returns (coef3[3,C], dgamma,
    dbeta) with dx = coef3[0]*g + coef3[1]*x + coef3[2], to be applied by conv1x1_dgrad_bnb / conv1x1_wgrad_bnb.
    dy_add: a second contribution to the incoming gradient, summed on the way in (g = mask * (dy + dy_add))."""
    _chk(dy, x, y_out, scale, shift, save_mean, save_invstd, gamma, dgamma, dbeta, dy_add)
    B, C, H, W = x.shape
    dgamma = dgamma if dgamma is not None else torch.empty_like(gamma)
    dbeta = dbeta if dbeta is not None else torch.empty_like(gamma)
    coef3 = torch.empty((3, C), dtype=torch.float32, device=x.device)
    ws = workspace(lib().scat_bn_ws(B, C, H * W), x.device)
    # algorithmic traffic: read dy (+ dy_add), x, the sign; write the masked gradient
    sign = 4.0 if y_out is not None else (0.03125 if y_mask is not None else 0.0)
    nb = x.numel() * (12.0 + sign + (4.0 if dy_add is not None else 0.0))
    _prof_hbm("bn_bwd_pre", nb, lib().scat_bn_bwd_pre, _p(dy), _p(dy_add), _p(x), _p(y_out), _p(y_mask), int(relu),
              _p(scale), _p(shift), _p(save_mean), _p(save_invstd), _p(gamma), _p(dgamma), _p(dbeta), _p(coef3), B, C,
              H * W, _p(ws), ws.numel(), _stream())
    return coef3, dgamma, dbeta


def epilogue_bnb_arm(x, mask, mean):
    """Arm the next accumulating pointwise data gradient on this host thread (include/scat_hip.h scat_epilogue_bnb_arm): it
    completes the gradient of relu(bn(x) + residual); its epilogue masks it with ``mask`` and leaves the BatchNorm
    backward's two sums per channel.  -> the partials' workspace (pass it to bn_bwd_pre_partials with epilogue_bnb_groups())."""
    _chk(x, mean)
    B, C, H, W = x.shape
    nbytes = C * ((B * H * W + 31) // 32 + 4) * 8
    part = workspace(nbytes, x.device, "bnbpart")
    lib().scat_epilogue_bnb_arm(_p(x), _p(mask), _p(mean), x.numel(), _p(part), nbytes)
    return part


def epilogue_bnb_groups():
    return lib().scat_epilogue_bnb_groups()


def bn_bwd_pre_partials(part, groups, x_shape, save_mean, save_invstd, gamma, dgamma=None, dbeta=None):
    """(coef3, dgamma, dbeta) of bn_bwd_pre from the sums an armed data-gradient epilogue left (the masked gradient is in place)"""
    _chk(save_mean, save_invstd, gamma, dgamma, dbeta)
    B, C, H, W = x_shape
    dgamma = dgamma if dgamma is not None else torch.empty_like(gamma)
    dbeta = dbeta if dbeta is not None else torch.empty_like(gamma)
    coef3 = torch.empty((3, C), dtype=torch.float32, device=gamma.device)
    lib().scat_bn_bwd_pre_partials(_p(part), groups, B, C, H * W, _p(save_mean), _p(save_invstd), _p(gamma), _p(dgamma),
                                   _p(dbeta), _p(coef3), _stream())
    return coef3, dgamma, dbeta


def conv1x1_dgrad_bnb(g, z, coef3, w, x_shape, out=None, accumulate=False, wp=None):
    _chk(g, z, coef3, w, out)
    B, Cin, H, W = x_shape
    Cout = w.shape[0]
    dx = out if out is not None else torch.empty(x_shape, dtype=torch.float32, device=g.device)
    ws, rdy = _wp_ws(wp, w, WPREP_CONV1X1_DGRAD, Cout, Cin, 1, 1, 0, lib().scat_conv1x1_s1_ws(Cin, Cout), g.device)
    _sk_arm(g.device)
    _prof(2.0 * B * H * W * Cout * Cin, lib().scat_conv1x1_s1_bnb, _p(g), _p(z), _p(coef3), _p(w), _p(dx), B, Cin, H * W,
          Cout, int(accumulate), _p(ws), ws.numel(), rdy, _stream())
    return dx


def conv1x1_wgrad_bnb(g, z, coef3, x, w_shape, in_scale=None, in_shift=None, in_relu=False, out=None,
                      ws_slot="default"):
    _chk(g, z, coef3, x, in_scale, in_shift, out)
    B, Cin, H, W = x.shape
    Cout = w_shape[0]
    dw = out if out is not None else torch.empty(w_shape, dtype=torch.float32, device=x.device)
    ws = _wgrad_workspace(lib().scat_conv1x1_wgrad_bnb_ws(B, Cin, H * W, Cout), x.device, ws_slot)
    _prof(2.0 * B * H * W * Cout * Cin, lib().scat_conv1x1_wgrad_bnb, _p(g), _p(z), _p(coef3), _p(x), _p(dw), B, Cin,
          H * W, Cout, _p(in_scale), _p(in_shift), int(in_relu), _p(ws), ws.numel(), _stream())
    return dw


# ---------------------------------------------------------------- pooling

def maxpool_fwd(x, scale=None, shift=None, relu=False):
    _chk(x, scale, shift)
    B, C, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
    idx = torch.empty((B, C, OH, OW), dtype=torch.int8, device=x.device)
    lib().scat_maxpool3x3s2_fwd(_p(x), _p(scale), _p(shift), int(relu), _p(y), _p(idx), B, C, H, W, _stream())
    return y, idx


def maxpool_bwd(dy, idx, x_shape):
    _chk(dy)
    B, C, H, W = x_shape
    dx = torch.empty(x_shape, dtype=torch.float32, device=dy.device)
    lib().scat_maxpool3x3s2_bwd(_p(dy), _p(idx), _p(dx), B, C, H, W, _stream())
    return dx


def subsample2(x):
    """x[:, :, ::2, ::2] packed (the input of a 1x1/stride-2 convolution as a stride-1 one sees it)"""
    _chk(x)
    B, C, H, W = x.shape
    y = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    lib().scat_subsample2(_p(x), _p(y), B, C, H, W, _stream())
    return y


def avgpool_fwd(x, relu=True):
    _chk(x)
    B, C, H, W = x.shape
    y = torch.empty((B, C), dtype=torch.float32, device=x.device)
    lib().scat_avgpool_fwd(_p(x), _p(y), B, C, H * W, int(relu), _stream())
    return y


def avgpool_bwd(dy, y, x_shape, relu=True, out=None, accumulate=False):
    _chk(dy, y, out)
    B, C, H, W = x_shape
    dx = out if out is not None else torch.empty(x_shape, dtype=torch.float32, device=dy.device)
    lib().scat_avgpool_bwd(_p(dy), _p(y), int(relu), _p(dx), B, C, H * W, int(accumulate), _stream())
    return dx


# ---------------------------------------------------------------- LayerNorm / attention / elementwise

def layernorm_fwd(x2d, gamma, beta, eps=1e-5):
    _chk(x2d, gamma, beta)
    rows, dim = x2d.shape
    y = torch.empty_like(x2d)
    st = torch.empty((2, rows), dtype=torch.float32, device=x2d.device)
    lib().scat_layernorm_fwd(_p(x2d), _p(gamma), _p(beta), _p(y), _p(st[0]), _p(st[1]), rows, dim, eps, _stream())
    return y, st[0], st[1]


def layernorm_bwd(dy2d, x2d, gamma, mean, rstd, want_params=True):
    """-> (dx, dgamma, dbeta); want_params=False: (dx, None, None) without the parameter sums"""
    _chk(dy2d, x2d, gamma, mean, rstd)
    rows, dim = x2d.shape
    dx = torch.empty_like(x2d)
    if not want_params:
        lib().scat_layernorm_bwd(_p(dy2d), _p(x2d), _p(gamma), _p(mean), _p(rstd), _p(dx), None, None, rows, dim, None, 0,
                                 _stream())
        return dx, None, None
    dg = torch.empty((2, dim), dtype=torch.float32, device=x2d.device)
    ws = workspace(lib().scat_layernorm_bwd_ws(rows, dim), x2d.device)
    lib().scat_layernorm_bwd(_p(dy2d), _p(x2d), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dg[0]), _p(dg[1]), rows,
                             dim, _p(ws), ws.numel(), _stream())
    return dx, dg[0], dg[1]


def attention_fwd(qkv, heads, dim_head, scale):
    """qkv[B,n,3*h*d] -> (out[B,n,h*d], attn[B,h,n,n])"""
    _chk(qkv)
    B, n, _ = qkv.shape
    out = torch.empty((B, n, heads * dim_head), dtype=torch.float32, device=qkv.device)
    attn = torch.empty((B, heads, n, n), dtype=torch.float32, device=qkv.device)
    lib().scat_attention_fwd(_p(qkv), _p(out), _p(attn), B, n, heads, dim_head, scale, _stream())
    return out, attn


# qkv projection + attention of a layer in one launch (csrc/vit_fused.hip).  Off by default: measured at batch 96
# (tools/vit_fused_bench.py, profiles/r02_vit_fused.txt) the fused launch takes 77 / 51 / 38 us for dim 784 / 392 / 196
# against 74 / 51 / 36 us for the two launches it replaces — 256 workgroups of three 21-token images each stream their
# head's weight slice at the per-CU L2 fetch rate (~25 GB/s: 2.2 us per 32-feature stage for 0.6 us of MFMA), and the
# step time does not move (the token path hides under layer3/layer4).
VIT_FUSED = _sw.ab("SCAT_VIT_FUSED", False)


def vit_fused_ok(n, dim, dim_head):
    return VIT_FUSED and n <= 32 and dim % 4 == 0 and dim_head == 64 and lib().scat_get_math_mode() == 1


def qkv_attention_fwd(h2d, wqkv, B, n, heads, scale):
    """h2d[B*n,dim], wqkv[3*heads*64,dim] -> (qkv[B*n,3*heads*64], out[B,n,heads*64], attn[B,heads,n,n]) — one launch
    per layer: each workgroup projects one head of a block of 128/n images and runs their attention from LDS"""
    _chk(h2d, wqkv)
    dim = h2d.shape[1]
    inner = heads * 64
    qkv = torch.empty((B * n, 3 * inner), dtype=torch.float32, device=h2d.device)
    out = torch.empty((B, n, inner), dtype=torch.float32, device=h2d.device)
    attn = torch.empty((B, heads, n, n), dtype=torch.float32, device=h2d.device)
    ws = workspace(lib().scat_vit_qkv_attn_fwd_ws(dim, heads), h2d.device, "vit")
    _prof(2.0 * B * n * 3 * inner * dim, lib().scat_vit_qkv_attn_fwd, _p(h2d), _p(wqkv), _p(qkv), _p(attn), _p(out), B, n,
          dim, heads, scale, _p(ws), ws.numel(), _stream())
    return qkv, out, attn


def attention_bwd(dout, qkv, attn, heads, dim_head, scale):
    _chk(dout, qkv, attn)
    B, n, _ = qkv.shape
    dqkv = torch.empty_like(qkv)
    lib().scat_attention_bwd(_p(dout), _p(qkv), _p(attn), _p(dqkv), B, n, heads, dim_head, scale, _stream())
    return dqkv


def performer_fwd(kqv, w, heads):
    """kqv[B,T,heads,3e], w[m,e] -> (y[B,T,heads*e], saved tuple)"""
    _chk(kqv, w)
    B, T, H, e3 = kqv.shape
    e, m = e3 // 3, w.shape[0]
    dev = kqv.device
    y = torch.empty((B, T, H * e), dtype=torch.float32, device=dev)
    kp = torch.empty((B, H, T, m), dtype=torch.float32, device=dev)
    qp = torch.empty_like(kp)
    kptv = torch.empty((B, H, e, m), dtype=torch.float32, device=dev)
    ksum = torch.empty((B, H, m), dtype=torch.float32, device=dev)
    D = torch.empty((B, H, T), dtype=torch.float32, device=dev)
    lib().scat_performer_fwd(_p(kqv), _p(w), _p(y), _p(kp), _p(qp), _p(kptv), _p(ksum), _p(D), B, T, H, e, m, _stream())
    return y, (kp, qp, kptv, ksum, D)


def performer_bwd(dy, kqv, w, y, saved):
    _chk(dy, kqv, w, y)
    B, T, H, e3 = kqv.shape
    e, m = e3 // 3, w.shape[0]
    kp, qp, kptv, ksum, D = saved
    dkqv = torch.empty_like(kqv)
    ws = workspace(lib().scat_performer_bwd_ws(B, T, H, e, m), kqv.device)
    lib().scat_performer_bwd(_p(dy), _p(kqv), _p(w), _p(y), _p(kp), _p(qp), _p(kptv), _p(ksum), _p(D), _p(dkqv), B, T,
                             H, e, m, _p(ws), ws.numel(), _stream())
    return dkqv


def dropout(x, p, seed):
    _chk(x)
    y = torch.empty_like(x)
    lib().scat_dropout(_p(x), _p(y), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream())
    return y


def gelu_fwd(x):
    _chk(x)
    y = torch.empty_like(x)
    lib().scat_gelu_fwd(_p(x), _p(y), x.numel(), _stream())
    return y


def gelu_bwd(dy, x):
    _chk(dy, x)
    dx = torch.empty_like(x)
    lib().scat_gelu_bwd(_p(dy), _p(x), _p(dx), x.numel(), _stream())
    return dx


def relu_fwd(x):
    _chk(x)
    y = torch.empty_like(x)
    lib().scat_relu_fwd(_p(x), _p(y), x.numel(), _stream())
    return y


def relu_bwd(dy, y):
    _chk(dy, y)
    dx = torch.empty_like(y)
    lib().scat_relu_bwd(_p(dy), _p(y), _p(dx), y.numel(), _stream())
    return dx


def axpy(a, b, alpha=1.0, out=None):
    _chk(a, b, out)
    y = out if out is not None else torch.empty_like(a)
    lib().scat_axpy(_p(a), _p(b), float(alpha), _p(y), a.numel(), _stream())
    return y


def tokens_fwd(x, pe, mask_token, masked_idx):
    """x[B,T,D] + pe[T,D]; rows in masked_idx (int32 GPU tensor or None) <- mask_token[D]."""
    _chk(x, pe, mask_token)
    B, T, D = x.shape
    y = torch.empty_like(x)
    nm = 0 if masked_idx is None else masked_idx.numel()
    lib().scat_tokens_fwd(_p(x), _p(pe), _p(mask_token), _p(masked_idx), nm, _p(y), B, T, D, _stream())
    return y


def tokens_bwd(dy, masked_idx, want_dmask=True):
    _chk(dy)
    B, T, D = dy.shape
    dx = torch.empty_like(dy)
    nm = 0 if masked_idx is None else masked_idx.numel()
    dm = torch.zeros((D,), dtype=torch.float32, device=dy.device) if want_dmask else None
    lib().scat_tokens_bwd(_p(dy), _p(masked_idx), nm, _p(dx), _p(dm) if nm else 0, B, T, D, _stream())
    return dx, dm


def preprocess_u8(src, out_hw=(224, 224), hwc=False):
    """uint8 [B,3,H,W] (or [B,H,W,3] with hwc) on the GPU -> fp32 [B,3,224,224] in [-1,1], bilinear."""
    if not (src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()):
        raise ScatError("preprocess_u8 needs a contiguous uint8 GPU tensor")
    B = src.shape[0]
    SH, SW = (src.shape[1], src.shape[2]) if hwc else (src.shape[2], src.shape[3])
    dst = torch.empty((B, 3, out_hw[0], out_hw[1]), dtype=torch.float32, device=src.device)
    lib().scat_preprocess_u8(_p(src), _p(dst), B, SH, SW, out_hw[0], out_hw[1], int(hwc), _stream())
    return dst


def fuse_sum(terms, relu=True):
    """relu?(sum of terms), terms = [(tensor [B,C,H>>k,W>>k], scale|None, shift|None, k)] in the order they are added
    (one pass; include/scat_hip.h scat_fuse_sum)."""
    assert 1 <= len(terms) <= 4
    k0 = [t for t in terms if t[3] == 0]
    B, C, H, W = (k0[0][0].shape if k0 else tuple(terms[0][0].shape[:2]) + tuple(d << terms[0][3] for d in terms[0][0].shape[2:]))
    ins, scs, shs, ks = [], [], [], []
    for t, sc, sh, k in terms:
        _chk(t, sc, sh)
        assert tuple(t.shape) == (B, C, H >> k, W >> k), (tuple(t.shape), (B, C, H, W), k)
        ins.append(t); scs.append(sc); shs.append(sh); ks.append(int(k))
    pad = 4 - len(terms)
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=ins[0].device)
    lib().scat_fuse_sum(*[_p(t) for t in ins + [None] * pad], *[_p(t) for t in scs + [None] * pad],
                        *[_p(t) for t in shs + [None] * pad], *(ks + [0] * pad), len(terms), _p(out), B, C, H, W, int(relu),
                        _stream())
    return out


def upsample_nearest_fwd(x, factor):
    _chk(x)
    B, C, H, W = x.shape
    y = torch.empty((B, C, H * factor, W * factor), dtype=torch.float32, device=x.device)
    lib().scat_upsample_nearest_fwd(_p(x), _p(y), B, C, H, W, factor, _stream())
    return y


def upsample_nearest_bwd(dy, factor):
    _chk(dy)
    B, C, OH, OW = dy.shape
    dx = torch.empty((B, C, OH // factor, OW // factor), dtype=torch.float32, device=dy.device)
    lib().scat_upsample_nearest_bwd(_p(dy), _p(dx), B, C, OH // factor, OW // factor, factor, _stream())
    return dx


def token_mean_fwd(x):
    _chk(x)
    B, T, D = x.shape
    y = torch.empty((B, D), dtype=torch.float32, device=x.device)
    lib().scat_token_mean_fwd(_p(x), _p(y), B, T, D, _stream())
    return y


def token_mean_bwd(dy, T):
    _chk(dy)
    B, D = dy.shape
    dx = torch.empty((B, T, D), dtype=torch.float32, device=dy.device)
    lib().scat_token_mean_bwd(_p(dy), _p(dx), B, T, D, _stream())
    return dx


def regressor_fwd(feat, feat_out, mean, w, bias, iters, root_relative=True):
    _chk(feat, feat_out, mean, w, bias)
    B, F = feat.shape
    P = w.shape[0]
    preds = torch.empty((iters + 1, B, P), dtype=torch.float32, device=feat.device)
    out = torch.empty((B, P), dtype=torch.float32, device=feat.device)
    lib().scat_regressor_fwd(_p(feat), _p(feat_out), _p(mean), _p(w), _p(bias), _p(preds), _p(out), B, F, P, iters,
                             int(root_relative), _stream())
    return out, preds


def regressor_bwd(dout, feat, preds, w, iters, root_relative=True, want_dfeat_out=True):
    _chk(dout, feat, preds, w)
    B, F = feat.shape
    P = w.shape[0]
    dfeat = torch.empty_like(feat)
    dfeat_out = torch.empty((B, P - 3), dtype=torch.float32, device=feat.device) if want_dfeat_out else None
    dw = torch.empty_like(w)
    dbias = torch.empty((P,), dtype=torch.float32, device=feat.device)
    ws = workspace(lib().scat_regressor_bwd_ws(B, F, P, iters), feat.device)
    lib().scat_regressor_bwd(_p(dout), _p(feat), _p(preds), _p(w), _p(dfeat), _p(dfeat_out), _p(dw), _p(dbias), B, F,
                             P, iters, int(root_relative), _p(ws), ws.numel(), _stream())
    return dfeat, dfeat_out, dw, dbias


def pose_length_term(pl_term):
    """train.py:178-183 from pl_term[B,C,H,W] -> scalar tensor l_pl"""
    pl_term = pl_term if pl_term.is_contiguous() else pl_term.contiguous()
    _chk(pl_term)
    B, C, H, W = pl_term.shape
    buf = torch.empty((B + 1,), dtype=torch.float32, device=pl_term.device)
    lib().scat_pose_length_term(_p(pl_term), _p(buf), buf.data_ptr() + 4 * B, B, C, H * W, _stream())
    return buf[B]


def loss_fwd_bwd(out, labels, w3d=100000.0, w2d=10.0):
    """labels[B,105] (or [B,166]) as train.py:188-198 -> (losses[3] = loss,l3d,l2d ; dout[B,66])"""
    _chk(out, labels)
    B = out.shape[0]
    ld = labels.shape[1]
    if out.dim() != 2 or out.shape[1] != 66 or labels.dim() != 2 or labels.shape[0] != B or ld not in (105, 166):
        raise ValueError(f"loss_fwd_bwd: outputs [B,66] and labels [B,105] or [B,166] expected (train.py:188-198), got "
                         f"{tuple(out.shape)} and {tuple(labels.shape)}")
    o3, o2 = (0, 63) if ld == 105 else (61, 124)
    losses = torch.empty((3,), dtype=torch.float32, device=out.device)
    dout = torch.empty_like(out)
    base = labels.data_ptr()
    lib().scat_loss_fwd_bwd(_p(out), base + 4 * o3, base + 4 * o2, ld, w3d, w2d, _p(losses), _p(dout), B, _stream())
    return losses, dout


def adam(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    _chk(p, g, m, v)
    lib().scat_adam(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale, _stream())
