#!/usr/bin/env python3
"""Which kernels do the runtime's small copy / fill kernels sit between?  (rocprofv3 --kernel-trace csv)
   scripts/trace_neighbours.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0 = rows[0][0]


def short(n):
    m = re.search(r"(lk_\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]


runs = collections.Counter()
prev = "start"
run_kind, run_len, run_start = None, 0, 0
for s, e, n in rows:
    k = "copy" if "copyBuffer" in n else "fill" if "fillBuffer" in n else None
    if k is None:
        if run_len:
            runs[(prev, run_kind, short(n))] += run_len
        prev, run_len, run_kind = short(n), 0, None
    else:
        run_kind = (run_kind + "+" + k) if run_kind and k not in run_kind else (run_kind or k)
        run_len += 1
print("small copies / fills by (kernel before, kinds, kernel after): count")
for key, c in runs.most_common(25):
    print(f"{c:8d}  {key}")
# per second of the run
per_s = collections.Counter()
for s, e, n in rows:
    if "copyBuffer" in n or "fillBuffer" in n:
        per_s[(s - t0) // 1_000_000_000] += 1
print("per second of the run:", dict(sorted(per_s.items())))
