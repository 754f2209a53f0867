"""Evaluation metrics of the reference's eval.py, restated as batched tensor code that runs on whatever
device the predictions are on (no per-frame host loop, no D2H until the final scalars).

These are the CALLERS' metrics, not kernels of the hot path: plain torch tensor
arithmetic, device-agnostic, checked on CPU against independent numpy in tests/test_metrics.py.

  mpjpe_mm            eval.py:753, :1026   mean_j ||pred_j - gt_j||_2 * 1000
  procrustes_align    eval.py:110-161      batch similarity transform (sR, t) minimising ||sR·S1 + t - S2||
  pa_mpjpe_mm         eval.py:953          MPJPE after Procrustes alignment
  pck / auc           eval.py:300-340      PCK over thresholds (mm) and its normalised area under curve
  accel_error         data_utils/eval_utils.py:23-48
"""
from __future__ import annotations

import torch


def _j(x):
    """[B,66] (cam + joints) or [B,63] or [B,21,3] -> [B,21,3]"""
    if x.dim() == 2:
        x = x[:, -63:].reshape(-1, 21, 3)
    return x


def mpjpe_mm(pred, gt):
    return (_j(pred) - _j(gt)).norm(dim=-1).mean() * 1000.0


def procrustes_align(S1, S2):
    """S1, S2: [B,N,3]. Returns S1 aligned to S2 (scale, rotation with det +1, translation)."""
    X1, X2 = S1.transpose(1, 2), S2.transpose(1, 2)                # [B,3,N] like the reference
    mu1, mu2 = X1.mean(dim=-1, keepdim=True), X2.mean(dim=-1, keepdim=True)
    A, Bm = X1 - mu1, X2 - mu2
    var1 = (A ** 2).sum(dim=(1, 2))
    K = A @ Bm.transpose(1, 2)
    U, s, Vh = torch.linalg.svd(K)
    V = Vh.transpose(1, 2)
    Z = torch.eye(3, device=S1.device, dtype=S1.dtype).repeat(S1.shape[0], 1, 1)
    Z[:, -1, -1] = torch.sign(torch.det(U @ V.transpose(1, 2)))
    R = V @ Z @ U.transpose(1, 2)
    scale = (R @ K).diagonal(dim1=1, dim2=2).sum(-1) / var1
    t = mu2 - scale[:, None, None] * (R @ mu1)
    return (scale[:, None, None] * (R @ X1) + t).transpose(1, 2)


def pa_mpjpe_mm(pred, gt):
    p, g = _j(pred), _j(gt)
    return (procrustes_align(p, g) - g).norm(dim=-1).mean() * 1000.0


def pck(pred, gt, thresholds_mm):
    """Percentage of joints (over the whole set, as eval.py:300-312 does with dist.flat) within each threshold."""
    d = (_j(pred) - _j(gt)).norm(dim=-1).reshape(-1) * 1000.0
    th = torch.as_tensor(thresholds_mm, dtype=d.dtype, device=d.device)
    return 100.0 * (d[None, :] <= th[:, None]).float().mean(dim=1)


def auc(thresholds_mm, pck_values):
    """Normalised trapezoid area (eval.py:327-338)."""
    x = torch.as_tensor(thresholds_mm, dtype=torch.float64)
    y = torch.as_tensor(pck_values, dtype=torch.float64).cpu()
    return float(torch.trapz(y, x) / torch.trapz(torch.ones_like(x), x))


def accel_error(joints_gt, joints_pred):
    """mean_j ||(X_{i-1} - 2 X_i + X_{i+1})_pred - (...)_gt|| per frame triple (all frames visible)."""
    ag = joints_gt[:-2] - 2 * joints_gt[1:-1] + joints_gt[2:]
    ap = joints_pred[:-2] - 2 * joints_pred[1:-1] + joints_pred[2:]
    return (ap - ag).norm(dim=2).mean(dim=1)
