#!/usr/bin/env python
"""Would the bn3 backward survive WITHOUT re-reading c3?  (VERDICT r2 item 7 / DESIGN 7-4.)

The folded bn3 backward reads the raw conv3 output c3 three times (masking reduce, data gradient, weight gradient).
Because c3 = W3 . a2, every use of c3 can be rewritten on the 4x smaller a2 = relu(bn2(c2)):
    sum_p g c3        = sum_k W3[c,k] G[c,k],           G = g a2^T   (the raw weight-gradient product)
    dW3               = ca G + cb W3 (a2 a2^T) + cc s^T, s = sum_p a2
    da2               = W3^T (ca g) + (W3^T diag(cb) W3) a2 + W3^T cc
The catch is arithmetic: the MFMA products accumulate in fp32 (per split-K slab, fp64 only across slabs), and the
BatchNorm backward subtracts mean-sized terms from them.  This probe takes the tensors of a real batch-96 train step
(every folded block), evaluates the Gram form with that accumulation pattern (fp32 inside one image = one slab, fp64
across images) and reports its error against an all-fp64 evaluation, next to the error of the shipped kernels.
    python tools/bn3_gram_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from scat_amd.models import resnet  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    net = bench.make_net("resnet50", 1, dev)
    step = bench.Step("resnet50", net, dev)
    u8, lab = bench.build_inputs(96, 2, dev)
    for _ in range(2):
        step(u8, lab)
    torch.cuda.synchronize()
    caught = []
    orig = resnet._block_backward

    def spy(bc, rec, dcur):
        blk, c3 = rec[0], rec[6]
        if bc.use_bnb and rec[11] is not None and c3.shape[2] >= resnet.BNB_MIN_H:
            torch.cuda.synchronize()
            pend = bc.pending[0]
            # (the optimiser updates the parameters before this script looks at them: keep the values the forward used)
            caught.append((rec, (dcur + pend if pend is not None else dcur).clone(), bc, blk.conv3.weight.detach().clone(),
                           blk.bn3.weight.detach().clone()))
        return orig(bc, rec, dcur)

    resnet._block_backward = spy
    step(u8, lab)
    torch.cuda.synchronize()
    resnet._block_backward = orig
    print(f"{'block':>10s} {'C':>5s} {'HxW':>6s} | {'dgamma: shipped':>16s} {'gram':>9s} | {'dW3/max: shipped':>17s} {'gram':>9s} |"
          f" {'cond':>7s}")
    f8 = torch.float64
    for rec, dout, bc, w3, gam3 in caught:
        blk, xin, c1, s1, c2, s2, c3, s3, cd, sd, out, omask = rec
        B, C, H, W = c3.shape
        K = c2.shape[1]
        v = lambda t: t.view(1, -1, 1, 1)
        res = xin if cd is None else cd * v(sd.scale) + v(sd.shift)
        mask = (c3 * v(s3.scale) + v(s3.shift) + res) > 0
        g = dout * mask
        a2 = torch.relu(c2 * v(s2.scale) + v(s2.shift))
        W3 = w3.view(C, K)
        mu, istd, gam = s3.mean.to(f8), s3.invstd.to(f8), gam3.to(f8)
        N = B * H * W
        # ---- all-fp64 evaluation from the stored fp32 tensors
        g8 = g.to(f8)
        S1 = g8.sum(dim=(0, 2, 3))
        S2 = (g8 * (c3.to(f8) - v(mu))).sum(dim=(0, 2, 3)) * istd            # dgamma
        ca = gam * istd
        cb = -gam * istd * istd * S2 / N
        cc = -ca * S1 / N - cb * mu
        dy3 = v(ca) * g8 + v(cb) * c3.to(f8) + v(cc)
        dW_true = torch.einsum("nchw,nkhw->ck", dy3, a2.to(f8))
        del dy3, g8
        # ---- Gram form with the kernels' accumulation pattern: fp32 inside an image, fp64 across images
        gf, af = g.view(B, C, H * W), a2.view(B, K, H * W)
        G = torch.bmm(gf, af.transpose(1, 2)).to(f8).sum(0)                   # [C, K]
        A = torch.bmm(af, af.transpose(1, 2)).to(f8).sum(0)                   # [K, K]
        s = af.sum(dim=2).to(f8).sum(0)                                      # [K]
        S1g = gf.sum(dim=2).to(f8).sum(0)
        gc3 = (W3.to(f8) * G).sum(1)
        S2g = (gc3 - mu * S1g) * istd
        cbg = -gam * istd * istd * S2g / N
        ccg = -ca * S1g / N - cbg * mu
        dW_gram = ca.view(-1, 1) * G + cbg.view(-1, 1) * (W3.to(f8) @ A) + ccg.view(-1, 1) * s.view(1, -1)
        # ---- the shipped kernels' results of the same step
        dg_ship = bc.grads[blk.bn3.weight].to(f8)
        dW_ship = bc.grads[blk.conv3.weight].view(C, K).to(f8)
        rel = lambda a, b: float(((a - b).abs() / b.abs().clamp_min(1e-30)).median())
        relmax = lambda a, b: float((a - b).abs().max() / b.abs().max())
        cond = float((((W3.to(f8) * G).abs().sum(1) + (mu * S1g).abs()) / (gc3 - mu * S1g).abs().clamp_min(1e-30)).median())
        name = [n for n, m in net.named_modules() if m is blk][0].replace("main_encoder.", "")
        print(f"{name:>10s} {C:5d} {H:3d}x{W:<3d}| {rel(dg_ship, S2):16.2e} {rel(S2g, S2):9.2e} | {relmax(dW_ship, dW_true):17.2e} "
              f"{relmax(dW_gram, dW_true):9.2e} | {cond:7.1f}", flush=True)
    print("dgamma: median over channels of |x - fp64| / |fp64|;  dW3: max |x - fp64| / max |fp64|;  cond: median over "
          "channels of (sum_k |W3 G| + |mean sum_p g|) / |sum_p g (c3 - mean)|, the amplification of G's rounding in dgamma")


if __name__ == "__main__":
    main()
