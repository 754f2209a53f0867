#!/usr/bin/env python
"""HBM bytes per launch of every kernel instantiation from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE
collected separately — they do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots').

  python tools/traffic_json.py <fetch_dir>/run_counter_collection.csv <write_dir>/run_counter_collection.csv out.json

Corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide
coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-B/lane stores and uncalibrated for the 4-B/lane
buffer stores of the contraction epilogues (reported as counted)."""
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
        out[k] = {"FETCH_SIZE_bytes_per_launch_raw": int(f * 1024), "WRITE_SIZE_bytes_per_launch": int(w * 1024),
                  "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024), "launches_profiled": n,
                  "note": "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM); "
                          "4-B/lane accesses uncalibrated"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(f"{len(out)} kernels -> {sys.argv[3]}")


if __name__ == "__main__":
    main()
