#!/usr/bin/env python
"""Per-kernel table of rocprofv3 --pmc counters (second launch of each kernel).  SQ_WAVE_CYCLES, SQ_WAIT_* and
SQ_ACTIVE_INST_* count quad-cycles of wave lifetime and are printed as fractions of SQ_WAVE_CYCLES (WAIT_ANY = parked on
s_waitcnt/barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_* = issuing); SQ_INSTS_* are printed per wave quad-cycle.
SQ_VALU_MFMA_BUSY_CYCLES counts SIMD cycles (32 per v_mfma_f32_32x32x16_bf16): MfmaUtil = that / (duration x 2.4 GHz x
1024 SIMDs), i.e. the fraction of the chip's matrix-pipe cycles AT THE 2.4 GHz MAXIMUM CLOCK that executed an MFMA
(MI355X_MICROARCH.md, cycle constants).  With GRBM_GUI_ACTIVE in the same pass the clock the chip actually held is
printed too (GRBM_GUI_ACTIVE / 8 XCDs / duration: 'DVFS give-back' in the same guide — MFMA-dense loops on random data
run well below 2.4 GHz) and MfmaBusy = busy cycles / (duration x that clock x 1024): the share of the cycles that
really elapsed."""
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
        if not any(k in name for k in ("gemm_kernel", "halo", "split_kernel", "conv1x1_kernel", "pc_kernel", "pw_kernel",
                                       "vit_", "planes_kernel", "attn_mfma")) or "wt3x3" in name:
            continue
        short = re.sub(r"scat::|Loader|void ", "", name)[:70]
        seen[short] += 1
        if seen[short] != 2:
            continue
        wc = c.get("SQ_WAVE_CYCLES", 1.0)
        util = ""
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and kt.get(did, 0):
            util = f"  MfmaUtil={c['SQ_VALU_MFMA_BUSY_CYCLES'] / (kt[did] * 2.4 * 1024):.3f}"
            if c.get("GRBM_GUI_ACTIVE"):
                ghz = c["GRBM_GUI_ACTIVE"] / 8 / kt[did]
                util += f"  clock={ghz:.2f}GHz  MfmaBusy={c['SQ_VALU_MFMA_BUSY_CYCLES'] / (kt[did] * ghz * 1024):.3f}"
        print(f"{short}  ns={kt.get(did, 0)}{util}")
        print("   " + "  ".join(f"{k[3:] if k.startswith('SQ_') else k}={v / wc:.3f}" if k != "SQ_WAVE_CYCLES" else f"WAVE_CYCLES={v:.3g}"
                                for k, v in sorted(c.items())))
