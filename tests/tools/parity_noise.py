#!/usr/bin/env python3
"""Characterise parity noise on config C2: |GPU - oracle(T=1)| beside the reference
algorithm's own |oracle(T=8) - oracle(T=1)| and |oracle(T=20) - oracle(T=1)| (T =
number_of_threads, whose only numerical effect is the summation split)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import correlation_amd as ca  # noqa: E402
from oracle import lk_oracle as lo  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000   # (LK_FORCE_SAFE / LK_FORCE_GROUP / LK_KEEP_SUMS in the environment: variants of the engine)
und, dfm = ca.speckle.speckle_pair(2048, 2048, seed=7)
e = ca.HipCorrelationEngine()
e.set_undeformed_image(und)
e.set_deformed_image(dfm)
e.set_rect_grid(24.0, 24.0, 2023.0, 2023.0, 100, 100)
e.commit_sectors()
gpu = e.correlate_all(np.zeros(6, np.float32))
xdim, ydim, cen = lo.rect_sector_geometry(24.0, 24.0, 2023.0, 2023.0, 100, 100)
pick = np.random.default_rng(1).choice(10000, N, replace=False)
lists = [lo.rect_points(cx - xdim, cy - ydim, cx + xdim, cy + ydim) for cx, cy in cen[pick]]
res = {}
for T in (1, 8, 20):
    o = lo.Oracle(n_threads=T)
    o.set_image(0, und)
    o.set_image(1, dfm)
    res[T] = o.correlate_sectors(lists, centers=cen[pick].astype(np.float32))


def dev(a, b):
    same = a["iterations"] == b["iterations"]
    d01 = np.abs(a["p"][:, :2] - b["p"][:, :2]).max(1)
    d25 = np.abs(a["p"][:, 2:] - b["p"][:, 2:]).max(1)
    dchi = np.abs(a["chi"] - b["chi"]) / np.abs(b["chi"])
    q = lambda v: {k: float(np.quantile(v, x)) for k, x in (("p50", .5), ("p95", .95), ("p99", .99), ("max", 1.0))}
    return {"iter_equal": float(same.mean()), "dchi_le_1e-5": float((dchi <= 1e-5).mean()), "dp01_same_iter": q(d01[same]), "dp25_same_iter": q(d25[same]),
            "dchi_same_iter": q(dchi[same]), "dp01_all": q(d01), "dchi_all": q(dchi)}


out = {"gpu_vs_T1": dev(gpu[pick], res[1]), "T8_vs_T1": dev(res[8], res[1]), "T20_vs_T1": dev(res[20], res[1]),
       "gpu_vs_T20": dev(gpu[pick], res[20])}
if os.environ.get("LK_BRIEF"):
    for k, v in out.items():
        print(k, "iter_equal %.4f  dchi<=1e-5 %.4f  dchi p50 %.2e p99 %.2e max %.2e  dp01 p99 %.2e max %.2e"
              % (v["iter_equal"], v["dchi_le_1e-5"], v["dchi_all"]["p50"], v["dchi_all"]["p99"], v["dchi_all"]["max"], v["dp01_all"]["p99"], v["dp01_all"]["max"]))
else:
    print(json.dumps(out, indent=1))
