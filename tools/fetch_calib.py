#!/usr/bin/env python
"""How does rocprofv3's FETCH_SIZE count the access widths of THIS library?  MI355X_MICROARCH.md (HBM): on gfx950 it
reports exactly half the bytes of a wide coalesced streaming read (16 B per lane) and "other access widths are
uncalibrated: calibrate on a known byte count in your own access pattern".  bench.py's roofline.traffic doubles FETCH_SIZE
for every kernel; the dominant pointwise kernel reads its activations with 4-byte-per-lane buffer loads.

  run:     tools/pmc_run.sh OUT "FETCH_SIZE" -- python3 tools/fetch_calib.py          (launches the known-byte kernels)
  report:  python3 tools/fetch_calib.py --report OUT/run_counter_collection.csv

Known-byte kernels, each reading a tensor far larger than the 256 MiB Infinity Cache exactly once:
  planes_from_f32_kernel<f,t>   16-byte loads per lane  (HW % 4 == 0)
  planes_from_f32_kernel<f,f>   4-byte loads per lane   (HW % 4 != 0)
"""
import collections
import csv
import os
import sys

SHAPES = {"16B": (64, 256, 56), "4B": (64, 256, 55)}     # B, C, H: 205 MB / 198 MB of fp32 read once


def run():
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from scat_amd import ops

    for name, (B, C, H) in SHAPES.items():
        x = torch.randn(B, C, H, H, device="cuda")
        flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
        buf = None
        for _ in range(3):
            flush.fill_(1)                       # evict x from the Infinity Cache between launches
            p = ops.planes_from(x, out=buf)
            buf = p.buf
        torch.cuda.synchronize()
        print(name, "done", x.numel() * 4, "bytes read per launch", flush=True)


def report(path):
    exp = {("true" if k == "16B" else "false"): b * c * h * h * 4 for k, (b, c, h) in SHAPES.items()}
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == "FETCH_SIZE" and "planes_from_f32_kernel<false" in r["Kernel_Name"]:
            v4 = "true" if "<false, true>" in r["Kernel_Name"] else "false"
            acc[v4].append(float(r["Counter_Value"]) * 1024)
    for v4, vals in sorted(acc.items()):
        got = sum(vals[1:]) / max(len(vals) - 1, 1)
        print(f"planes_from_f32_kernel<false,{v4}> ({'16' if v4 == 'true' else '4'}-byte loads per lane): FETCH_SIZE "
              f"{got / 1e6:8.1f} MB per launch, tensor {exp[v4] / 1e6:8.1f} MB -> counter / bytes = {got / exp[v4]:.3f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2])
    else:
        run()
