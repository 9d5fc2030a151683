#!/usr/bin/env python3
"""SQ_INSTS_VALU of all lk_solve_kernel launches of ONE solve, from a rocprofv3 --pmc SQ_INSTS_VALU pass over a script
that solves the same pair n times: scripts/valu_per_solve.py <dir with *counter_collection.csv> <n solves>"""
import csv
import glob
import os
import sys

total = 0.0
per_kernel = {}
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == "SQ_INSTS_VALU" and "lk_solve_kernel" in row["Kernel_Name"]:
                total += float(row["Counter_Value"])
                k = row["Kernel_Name"].split("lk_solve_kernel")[1].split("(")[0]
                per_kernel[k] = per_kernel.get(k, 0.0) + float(row["Counter_Value"])
n = int(sys.argv[2])
print(total / n)
for k, v in sorted(per_kernel.items()):
    print(f"#  lk_solve_kernel{k}: {v / n:.4g} per solve", file=sys.stderr)
