#!/usr/bin/env python
"""Where does the GPU wait inside a train step?  Reads a rocprofv3 --kernel-trace CSV, cuts it into steps at the stem's
max-pool forward (one per step; a network without one: at the input pipeline's preprocess kernel), and reports per steady-state step: wall time, time with >= 1 kernel running (union of the
kernel intervals over all queues), summed kernel time, and the idle gaps grouped by the kernel that ends them."""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(.*$", "", n).replace("void ", "").replace("scat::", "")
    return n.replace("false", "f").replace("true", "t")[:60]


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8          # warm-up steps to drop
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    cuts = [e for s, e, n, q in rows if "maxpool_fwd" in n]
    if not cuts:
        cuts = [e for s, e, n, q in rows if "preprocess_kernel" in n]
    if len(cuts) < skip + 3:
        sys.exit(f"only {len(cuts)} steps in the trace")
    t0, t1 = cuts[skip], cuts[-1]
    steps = len(cuts) - 1 - skip
    win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    wall = (t1 - t0) / steps
    ksum = sum(e - s for s, e, _, _ in win) / steps
    busy, gaps, cur_end = 0, collections.defaultdict(lambda: [0, 0]), t0
    for s, e, n, q in win:
        if s > cur_end:
            g = gaps[short(n)]
            g[0] += s - cur_end
            g[1] += 1
            cur_s = s
        else:
            cur_s = cur_end
        if e > cur_end:
            busy += e - max(cur_s, s) if s > cur_end else e - cur_end
            cur_end = e
    busy /= steps
    queues = collections.Counter()
    for s, e, n, q in win:
        queues[q] += e - s
    print(f"{steps} steps: wall {wall / 1e6:.3f} ms/step, busy (>=1 kernel) {busy / 1e6:.3f} ms, idle "
          f"{(wall - busy) / 1e6:.3f} ms, summed kernel time {ksum / 1e6:.3f} ms, launches/step {len(win) / steps:.0f}")
    print("kernel time per queue (ms/step):", {q: round(v / steps / 1e6, 3) for q, v in queues.items()})
    tot_gap = sum(v[0] for v in gaps.values()) / steps
    print(f"idle gaps by the kernel that follows them (total {tot_gap / 1e6:.3f} ms/step):")
    for n, (ns, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"  {n:62s} {ns / steps / 1e3:8.1f} us/step  {c / steps:6.1f} gaps/step  avg {ns / c / 1e3:6.2f} us")
    # the same window by kernel: what a steady-state step launches (a whole-run stats file also counts the set-up:
    # parameter uploads, the stream-plan probes)
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, n, q in win:
        per[short(n)][0] += 1
        per[short(n)][1] += e - s
    ntop = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if ntop:
        print(f"kernels of the steady-state steps ({len(per)} kinds):")
        for n, (c, ns) in sorted(per.items(), key=lambda kv: -kv[1][1])[:ntop]:
            print(f"  {n:62s} calls/step={c / steps:7.1f}  ms/step={ns / steps / 1e6:7.3f}  avg_us={ns / c / 1e3:8.1f}")


if __name__ == "__main__":
    main()
