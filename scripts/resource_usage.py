#!/usr/bin/env python3
"""Registers / occupancy of every lk_solve_kernel instance from a -Rpass-analysis=kernel-resource-usage log
(scripts/tune_build.sh NAME -Rpass-analysis=kernel-resource-usage 2> log)."""
import re
import subprocess
import sys

rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w \[\]/]*): (\w+) \[", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
dem = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("MODEL, INTERP, GROUP, THREADS, SAFE, REF[, SEQ]: VGPRs SGPRs(spill) scratch waves/SIMD LDS")
for r, d in zip(rows, dem):
    if "lk_solve_kernel" in d:
        print(re.search(r"<(.*)>", d).group(1).ljust(44), r.get("VGPRs"), r.get("TotalSGPRs"), "(%s)" % r.get("SGPRs Spill"),
              r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]"))
