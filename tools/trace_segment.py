#!/usr/bin/env python
"""Timeline of one steady-state step between two kernels (substring match) from a rocprofv3 --kernel-trace CSV:
offset, duration and the idle gap in front of every launch — for the serial head section between the backbone's
forward and backward, where nothing overlaps."""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(.*$", "", n).replace("void ", "").replace("scat::", "")
    return n.replace("false", "f").replace("true", "t")[:70]


path, first, last = sys.argv[1], sys.argv[2], sys.argv[3]
step = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
cuts = [e for s, e, n in rows if "maxpool_fwd" in n]      # one per step (the stem)
t0, t1 = cuts[step], cuts[step + 1]
win = [r for r in rows if t0 <= r[0] < t1]
i0 = next(i for i, r in enumerate(win) if first in r[2])
i1 = next(i for i, r in enumerate(win) if last in r[2] and i > i0)
seg = win[i0:i1 + 1]
prev_end = seg[0][0]
tot_k = tot_g = 0
for s, e, n in seg:
    gap = max(0, s - prev_end)
    print(f"{(s - seg[0][0]) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  {short(n)}")
    tot_k += e - s
    tot_g += gap
    prev_end = max(prev_end, e)
print(f"segment {(seg[-1][1] - seg[0][0]) / 1e3:.1f} us: {len(seg)} launches, kernel time {tot_k / 1e3:.1f} us, gaps {tot_g / 1e3:.1f} us")
