#!/usr/bin/env python
"""Summarise a rocprofv3 kernel_stats.csv: short kernel names, per-step ms."""
import csv, re, sys
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
if len(sys.argv) > 2 and sys.argv[2] == "auto":      # one loss kernel per train step
    # (the HRNet bench step has no loss kernel: one preprocess kernel per step there)
    steps = float(next((r["Calls"] for r in rows if "loss_kernel" in r["Name"]), None) or
                  next(r["Calls"] for r in rows if "preprocess_kernel" in r["Name"]))
else:
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
def short(n):
    n = re.sub(r'\(.*$', '', n).replace('void ', '').replace('scat::', '')
    n = n.replace('GatherLoader', 'G').replace('MatLoader', 'M').replace('gemm_kernel', 'gemm').replace(' ', '')
    n = n.replace('false', 'f').replace('true', 't')
    return n[:78]
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total {tot/1e6/steps:.2f} ms/step over {steps:g} steps")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{short(r['Name']):80s} calls/step={float(r['Calls'])/steps:6.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.2f}%")
