#!/usr/bin/env python
"""SCAT_TUNE=77: where a tile of the pointwise split kernel spends its time (prologue / stage loop / epilogue)."""
import os, sys
os.environ["SCAT_TUNE"] = "77"
# the stamps exist only in the diag build (python -m scat_amd.build --diag)
os.environ.setdefault("SCAT_LIBPATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libscat_hip_diag.so"))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops
B = 96
for cin, cout, H in ((512, 256, 28), (256, 1024, 14), (64, 256, 56), (2048, 512, 7)):
    x = torch.randn(B, cin, H, H, device="cuda"); w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
    for _ in range(3):
        y = ops.conv2d_fwd(x, w, 1, 0)
    torch.cuda.synchronize()
    yc = y.cpu().numpy().reshape(B, cout, H * H)
    HW = H * H
    rows = []
    for i0 in range(0, cout, 128):
        for j0 in range(0, B * HW, 128):
            n, hw = divmod(j0, HW)
            if hw + 8 <= HW and hw % 2 == 0:
                rows.append(yc[n, i0, hw:hw + 8].view(np.uint64))
    t = np.array(rows).astype(np.float64) / 100.0
    t = t[(t[:, 3] > t[:, 0]) & (t[:, 3] - t[:, 0] < 1e4)]
    base = t[:, 0].min()
    print(f"{cin}->{cout} @{H}: {len(t)} tiles, kernel span {t[:, 3].max() - base:.1f} us, stages {cin // 32}; "
          f"{ops.lib().scat_last_kernel().decode()}")
    for name, a, b in (("prologue", 0, 1), ("stage loop", 1, 2), ("epilogue", 2, 3), ("tile", 0, 3)):
        dlt = t[:, b] - t[:, a]
        print(f"   {name:10s} us: min {dlt.min():6.1f} median {np.median(dlt):6.1f} max {dlt.max():6.1f}")
    st = np.sort(t[:, 0] - base)
    print("   tile starts (us) percentiles 0/25/50/75/100:", np.round(np.percentile(st, [0, 25, 50, 75, 100]), 1))
    # first round (tiles that start within 5 us of the kernel's first tile) against the rest
    first = (t[:, 0] - base) < 5.0
    for name, sel in (("first round", first), ("later tiles", ~first)):
        if sel.sum():
            tt = t[sel]
            print(f"   {name:11s}: {sel.sum():5d} stamped tiles; start {np.median(tt[:, 0] - base):6.1f}; prologue "
                  f"{np.median(tt[:, 1] - tt[:, 0]):5.1f} loop {np.median(tt[:, 2] - tt[:, 1]):5.1f} epilogue "
                  f"{np.median(tt[:, 3] - tt[:, 2]):5.1f}; ends at {np.median(tt[:, 3] - base):6.1f} (max {(tt[:, 3] - base).max():6.1f})")
