#!/usr/bin/env python
"""Which HIP calls are behind the __amd_rocclr_copyBuffer launches of a step?  Reads rocprofv3's hip-api trace (csv) and prints
the memcpy-like calls by name and, for a sample of them, the calls of the same thread around each."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
name_key = "Function" if "Function" in rows[0] else "Name"
tid_key = "Thread_Id" if "Thread_Id" in rows[0] else "Tid"
start_key = "Start_Timestamp"
rows.sort(key=lambda r: int(r[start_key]))
cnt = collections.Counter(r[name_key] for r in rows)
print("HIP calls:", {k: v for k, v in cnt.most_common(25)})
by_tid = collections.defaultdict(list)
for r in rows:
    by_tid[r[tid_key]].append(r)
shown = 0
ctx = collections.Counter()
for tid, rs in by_tid.items():
    for i, r in enumerate(rs):
        if "emcpy" in r[name_key]:
            prev = [x[name_key] for x in rs[max(0, i - 2): i]]
            nxt = [x[name_key] for x in rs[i + 1: i + 3]]
            ctx[(r[name_key], tuple(prev), tuple(nxt))] += 1
for (n, prev, nxt), c in ctx.most_common(15):
    print(f"{c:6d}  {n}   after {prev}   before {nxt}")
