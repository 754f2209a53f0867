#!/usr/bin/env python
"""Train-step throughput of the other BASELINE configurations on ONE GPU (bench.py measures config 1/2 only):
  hrnet   EncoderTransformerHRNet, HRNet-W32 + vit.Transformer(196,3,8,64,392)     (BASELINE config 4)
  coarse  EncoderTransformerCoarse (train_coarse.py's network)
  performer  EncoderPerformer: ResNet-50 tokens + 3 FAVOR+ blocks (heads 16), iteration 5     (BASELINE config 5)
Plain torch.optim.Adam + (pred*cot).sum() as the loss: these wrappers go through the per-module autograd path
(scat_amd/nn.py), not the fused ResNet executor."""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import synth  # noqa: E402

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="hrnet", choices=["hrnet", "coarse", "performer"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    from scat_amd.models import hand_net as H

    opt = SimpleNamespace(vit_heads=8, pl_reg=False, iteration=3, pos_embed=True, mask_rate=0.2, vit_depth=3,
                          hrnet_width=32)
    if a.config == "hrnet":
        net = H.EncoderTransformerHRNet(opt, T(synth.mean_params(104, 61)))
        net.load_state_dict(synth.to_torch(synth.hrnet_wrapper_state(105, net.state_dict())), strict=True)
    elif a.config == "performer":     # BASELINE config 5: heads 16, iteration 5, mask 0.2
        opt.vit_heads, opt.iteration = 16, 5
        net = H.EncoderPerformer(opt, T(synth.mean_params(104, 66)))
    else:
        net = H.EncoderTransformerCoarse(opt, T(synth.mean_params(104, 66)))
    net.cuda().train()
    optim = torch.optim.Adam(net.parameters(), lr=1e-5)
    x = T(synth.images(106, a.batch)).cuda()

    def step():
        optim.zero_grad(set_to_none=True)
        out = net(x)
        pred = out[0] if isinstance(out, tuple) else out
        (pred * 1e-3).sum().backward()
        optim.step()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"{a.config}: batch {a.batch}, {dt * 1e3:.1f} ms/step, {a.batch / dt:.1f} img/s "
          f"(peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)")


if __name__ == "__main__":
    main()
