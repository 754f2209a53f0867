#!/usr/bin/env python
"""HBM bytes per launch of every kernel instantiation from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE
collected separately — they do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots').

  python tools/traffic_json.py <fetch_dir>/run_counter_collection.csv <write_dir>/run_counter_collection.csv out.json

Corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide
coalesced reads (16 B per lane), so it is doubled — except for the kernels that fetch their activations with 4-byte-per-
lane buffer loads (conv1x1_split_kernel and its taps / stem forms, the narrow producer/consumer form): the guide asks for
a calibration "on a known byte count in your own access pattern", and tools/fetch_calib.py measured counter / bytes =
0.565 for that width against 0.500 for 16-byte loads (profiles/r04_fetch_calib.txt), so their FETCH_SIZE is divided by
0.565 (x 1.77), not doubled.  WRITE_SIZE is exact for 16-B/lane stores; for the 4-B/lane buffer stores of the contraction
epilogues it is reported as counted (tools/gpu_r04_b.sh checks it against a layer whose output bytes are known)."""
import collections
import csv
import json
import re
import sys


def short(name):
    n = re.sub(r"\(.*$", "", name).replace("void ", "").replace("scat::", "")
    n = n.replace("GatherLoader", "G").replace("MatLoader", "M").replace("gemm_kernel", "gemm")
    return n.replace(" ", "").replace("false", "f").replace("true", "t")


def mean_by_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = acc[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    fetch = mean_by_kernel(sys.argv[1], "FETCH_SIZE")
    write = mean_by_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) & set(write)):
        if not any(t in k for t in ("gemm<", "split_kernel", "halo_kernel", "conv1x1_kernel", "pc_kernel", "pw_kernel", "sk_kernel", "rows_kernel", "rows64_kernel", "vit_", "bn_")):
            continue
        f, n = fetch[k]
        w, _ = write[k]
        narrow = k.startswith("conv1x1_split_kernel") or k.startswith("conv1x1_sk_kernel")
        if k.startswith("conv1x1_pc_kernel"):          # <MI, TF, DIAG, W4, ...>: the W4 form loads 16 bytes per lane
            args = k[k.index("<") + 1:].split(",")
            narrow = len(args) > 3 and args[3] == "f"
        factor = 1.0 / 0.565 if narrow else 2.0
        out[k] = {"FETCH_SIZE_bytes_per_launch_raw": int(f * 1024), "WRITE_SIZE_bytes_per_launch": int(w * 1024),
                  "fetch_factor": round(factor, 3),
                  "hbm_bytes_per_launch": int(factor * f * 1024 + w * 1024), "launches_profiled": n,
                  "note": ("FETCH_SIZE / 0.565: 4-byte-per-lane activation loads, calibrated by tools/fetch_calib.py "
                           "(profiles/r04_fetch_calib.txt)" if narrow else
                           "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM)")}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(f"{len(out)} kernels -> {sys.argv[3]}")


if __name__ == "__main__":
    main()
