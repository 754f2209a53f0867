#!/usr/bin/env python
"""Per-kernel table of rocprofv3 --pmc counters normalised by SQ_WAVE_CYCLES (second launch of each kernel)."""
import collections
import csv
import re
import sys

for d in sys.argv[1:]:
    rows = list(csv.DictReader(open(f"{d}/run_counter_collection.csv")))
    kt = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
          for r in csv.DictReader(open(f"{d}/run_kernel_trace.csv"))}
    disp = collections.OrderedDict()
    for r in rows:
        disp.setdefault((r["Dispatch_Id"], r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    seen = collections.Counter()
    print("==", d)
    for (did, name), c in disp.items():
        if not any(k in name for k in ("gemm_kernel", "halo", "split_kernel", "conv1x1_kernel")) or "wt3x3" in name:
            continue
        short = re.sub(r"scat::|Loader|void ", "", name)[:70]
        seen[short] += 1
        if seen[short] != 2:
            continue
        wc = c.get("SQ_WAVE_CYCLES", 1.0)
        print(f"{short}  ns={kt.get(did, 0)}")
        print("   " + "  ".join(f"{k[3:] if k.startswith('SQ_') else k}={v / wc:.3f}" if k != "SQ_WAVE_CYCLES" else f"WAVE_CYCLES={v:.3g}"
                                for k, v in sorted(c.items())))
