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
