"""Shared helpers for golden generation and parity tests (test infrastructure)."""
import numpy as np
import torch


def digest(t, n=16):
    """Small, position-sensitive summary of a big tensor:
    [sum, abs-sum, cos-weighted sum, L2] followed by n strided samples (float64)."""
    f = torch.as_tensor(t).detach().cpu().double().flatten()
    w = torch.cos(torch.arange(f.numel(), dtype=torch.float64) * 0.37) + 0.5
    step = max(1, f.numel() // n)
    return np.concatenate([
        np.array([f.sum(), f.abs().sum(), (f * w).sum(), f.norm()]),
        f[::step][:n].numpy(),
    ]).astype(np.float64)


def rel_err(a, b):
    """Norm-wise relative error max|a-b| / max|b| (the parity metric; tolerance stated per test)."""
    a = np.asarray(torch.as_tensor(a).detach().cpu().double())
    b = np.asarray(torch.as_tensor(b).detach().cpu().double())
    assert a.shape == b.shape, (a.shape, b.shape)
    d = np.abs(a - b).max() if a.size else 0.0
    return float(d / max(np.abs(b).max(), 1e-30))


def digest_err(d_got, d_exp):
    """Relative error of a digest: sums are compared against the abs-sum scale, samples against max sample."""
    d_got, d_exp = np.asarray(d_got), np.asarray(d_exp)
    scale = max(abs(d_exp[1]), 1e-30)
    e = [abs(d_got[0] - d_exp[0]) / scale, abs(d_got[1] - d_exp[1]) / scale,
         abs(d_got[2] - d_exp[2]) / scale, abs(d_got[3] - d_exp[3]) / max(d_exp[3], 1e-30)]
    s = max(np.abs(d_exp[4:]).max(), 1e-30)
    e.append(np.abs(d_got[4:] - d_exp[4:]).max() / s)
    return float(max(e))
