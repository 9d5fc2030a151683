#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 --pmc passes collected by scripts/profile_round.sh."""
import csv
import glob
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for path in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
            acc[(name, row["Counter_Name"])].append(float(row["Counter_Value"]))
print(f"rocprofv3 --pmc passes (one counter set per pass) on `python3 scripts/quick_solve.py C2 5`, MI355X, {tag}")
print("units: FETCH_SIZE / WRITE_SIZE in KiB per dispatch; TCC_* in requests per dispatch; SQ_* summed over the device")
for (name, counter), v in sorted(acc.items()):
    if "lk_" in name:
        print(f"{name:44s} {counter:22s} dispatches={len(v)} mean={sum(v) / len(v):.6g}")
