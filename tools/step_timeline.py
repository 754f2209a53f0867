#!/usr/bin/env python
"""Whole timeline of ONE steady-state train step from a rocprofv3 --kernel-trace CSV: every launch of every queue with
its start offset, duration, queue and the GPU-idle gap (no kernel running on any queue) that ends with it — what
tools/trace_gaps.py summarises, in order, so the neighbours of a gap can be read off.
usage: step_timeline.py run_kernel_trace.csv [step=8] [min_gap_us=0: print everything | >0: only rows around gaps]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(.*$", "", n).replace("void ", "").replace("scat::", "")
    return n.replace("false", "f").replace("true", "t")[:64]


path = sys.argv[1]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"))
              for r in csv.DictReader(open(path)))
cuts = [s for s, e, n, q in rows if "preprocess_kernel" in n] or [s for s, e, n, q in rows if "maxpool_fwd" in n]
t0, t1 = cuts[step], cuts[step + 1]
win = [r for r in rows if t0 <= r[0] < t1]
qs = sorted({r[3] for r in win})
busy_end = win[0][0]
out = []
for s, e, n, q in win:
    gap = max(0, s - busy_end)
    out.append((s, e, n, q, gap))
    busy_end = max(busy_end, e)
show = set()
for i, r in enumerate(out):
    if min_gap <= 0 or r[4] / 1e3 >= min_gap:
        show.update(range(max(0, i - 3), min(len(out), i + 2)))
prev = -1
for i in sorted(show):
    s, e, n, q, gap = out[i]
    if prev >= 0 and i != prev + 1:
        print("      ...")
    prev = i
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  q{qs.index(q)}  {'GAP %6.1f' % (gap / 1e3) if gap else '          '}  {short(n)}")
print(f"step {(t1 - t0) / 1e6:.3f} ms, {len(win)} launches, idle {sum(r[4] for r in out) / 1e6:.3f} ms")
