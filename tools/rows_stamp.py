#!/usr/bin/env python
"""SCAT_WG_ROWS_STAMP=1: where the row-walking weight gradient spends its time (prologue / row loop / epilogue)."""
import os, sys
os.environ["SCAT_WG_ROWS_STAMP"] = "1"
# the stamps exist only in the diag build (python -m scat_amd.build --diag)
os.environ.setdefault("SCAT_LIBPATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libscat_hip_diag.so"))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scat_amd import ops
B, C, H = 96, 32, 56
x = torch.randn(B, C, H, H, device="cuda"); dy = torch.randn(B, C, H, H, device="cuda")
for _ in range(3):
    ops.conv2d_wgrad(dy, x, (C, C, 3, 3), 1, 1)
torch.cuda.synchronize()
ws = ops.workspace(1, x.device).cpu().numpy()
N = C * 9
splits = 256
st = np.zeros((splits, 4), dtype=np.uint64)
for z in range(splits):
    st[z] = ws[z * C * N * 4: z * C * N * 4 + 32].view(np.uint64)
t = st.astype(np.float64) / 100.0     # us (100 MHz)
base = t[:, 0].min()
print("start spread us: min %.1f median %.1f max %.1f" % tuple(np.percentile(t[:, 0] - base, [0, 50, 100])))
for name, a, b in (("prologue", 0, 1), ("row loop", 1, 2), ("epilogue", 2, 3), ("total", 0, 3)):
    dlt = t[:, b] - t[:, a]
    print(f"{name:9s} us: min {dlt.min():7.1f} median {np.median(dlt):7.1f} max {dlt.max():7.1f}")
print("kernel span us: %.1f" % (t[:, 3].max() - base))
