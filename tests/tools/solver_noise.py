#!/usr/bin/env python3
"""CPU-only experiment: how far do alternative 6x6 solvers move the Newton-Raphson results
away from the reference's ColPivHouseholderQR, compared with the reference's own
thread-count noise?  (oracle only; config C2 geometry on a 1024^2 pair)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import correlation_amd as ca  # noqa: E402
from oracle import lk_oracle as lo  # noqa: E402

und, dfm = ca.speckle.speckle_pair(1024, 1024, seed=7)
xdim, ydim, cen = lo.rect_sector_geometry(24.0, 24.0, 999.0, 999.0, 50, 50)
pick = np.random.default_rng(1).choice(len(cen), 1500, replace=False)
lists = [lo.rect_points(cx - xdim, cy - ydim, cx + xdim, cy + ydim) for cx, cy in cen[pick]]


def run(T, solver):
    o = lo.Oracle(n_threads=T, solver=solver)
    o.set_image(0, und)
    o.set_image(1, dfm)
    return o.correlate_sectors(lists, centers=cen[pick].astype(np.float32))


def strict(a, b):
    dp = np.abs(a["p"] - b["p"])
    chi = np.abs(a["chi"] - b["chi"]) / np.abs(b["chi"])
    return float(((dp[:, :2] <= 1e-4).all(1) & (dp[:, 2:] <= 1e-6).all(1) & (chi <= 1e-5)
                  & (a["iterations"] == b["iterations"])).mean()), float(np.median(dp[:, :2].max(1))), \
        float(np.median(dp[:, 2:].max(1)))


base = run(1, 0)
for label, r in (("QR  T=8 ", run(8, 0)), ("LDLt T=1", run(1, 1)), ("f64 T=1 ", run(1, 2)), ("LDLt T=8", run(8, 1)),
                 ("f64 T=8 ", run(8, 2))):
    print(label, "strict frac %.3f  median |dp01| %.2e  median |dp25| %.2e" % strict(r, base))
