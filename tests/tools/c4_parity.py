#!/usr/bin/env python3
"""Parity of BASELINE config 4's tiny sectors (7x7 samples, two starved levels) against the CPU
oracle on a subset: engine vs oracle(T=1), with the oracle's own T=8 run as the yardstick."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import correlation_amd as ca  # noqa: E402
from correlation_amd.workload import C4 as wl  # noqa: E402
from oracle import lk_oracle as lo  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7)
e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
e.set_undeformed_image(und)
e.set_deformed_image(dfm)
e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs, 20000, N)
e.commit_sectors()
got = e.correlate_all(np.zeros(6, np.float32))
st = e.stats()
xd, yd, cen = lo.rect_sector_geometry(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
cen = cen[20000:20000 + N]
lists = [lo.rect_points(cx - xd, cy - yd, cx + xd, cy + yd) for cx, cy in cen]
res = {}
for T in (1, 8):
    o = lo.Oracle(n_threads=T, py_stop=wl.py_stop)
    o.set_image(0, und)
    o.set_image(1, dfm)
    res[T] = o.correlate_sectors(lists, centers=cen.astype(np.float32))
w, w8 = res[1], res[8]
ok = w["error_code"] == 0


def summary(a, label):
    d = np.abs(a["p"] - w["p"])[ok][:, :2].max(1)
    print(f"{label}: error codes equal {np.array_equal(a['error_code'], w['error_code'])}, iterations equal "
          f"{(a['iterations'] == w['iterations'])[ok].mean():.3f}, |dp01| median {np.median(d):.2e} p90 {np.quantile(d, .9):.2e} "
          f"p99 {np.quantile(d, .99):.2e} max {d.max():.2e}, within 1e-3 px: {(d < 1e-3).mean():.3f}")


summary(got, f"engine (ill-conditioned solves {st['ill_conditioned_solves']})")
summary(w8, "oracle T=8 (yardstick)")
