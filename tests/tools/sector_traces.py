#!/usr/bin/env python3
"""CPU tool (test infrastructure: it drives the oracle): per-sector evaluation traces of a rectangular
workload - the pyramid level of every evaluation of every sector - for the launch simulator
(who are the stragglers, and at which level).  python tests/tools/sector_traces.py [C2|C4|C5] [n_sectors] -> /tmp/traces_<wl>.npz"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import correlation_amd as ca  # noqa: E402
from correlation_amd import workload  # noqa: E402
from oracle import lk_oracle as lo  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
wl = getattr(workload, name)
cache = f"/tmp/speckle_{wl.size}.npz"
if os.path.exists(cache):
    z = np.load(cache)
    und, dfm = z["und"], z["dfm"]
else:
    und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7)
    np.savez(cache, und=und, dfm=dfm)
o = lo.Oracle(model=wl.model, py_stop=wl.py_stop)
o.set_image(0, und)
o.set_image(1, dfm)
xd, yd, cen = lo.rect_sector_geometry(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
S = len(cen) if len(sys.argv) < 3 else min(int(sys.argv[2]), len(cen))
levels = np.zeros((S, 128), np.int8) - 1
n_ev = np.zeros(S, np.int32)
t0 = time.time()
for s in range(S):
    cx, cy = cen[s]
    xy = lo.rect_points(cx - xd, cy - yd, cx + xd, cy + yd)
    rec, tr = o.newton_raphson(np.zeros(6, np.float32), xy, center=(float(cx), float(cy)), trace_cap=128)
    n_ev[s] = len(tr)
    levels[s, :len(tr)] = tr["level"]
print(f"{name}: {S} sectors in {time.time() - t0:.1f} s; evaluations per sector mean {n_ev.mean():.2f} "
      f"p50/p90/p99/max {np.percentile(n_ev, [50, 90, 99, 100])}")
for L in range(wl.py_stop, -1, -1):
    c = (levels == L).sum(1)
    print(f"  level {L}: evaluations mean {c.mean():.2f} p99 {np.percentile(c, 99):.0f} max {c.max()}")
top = np.argsort(-n_ev)[:12]
for s in top:
    print(f"  sector {s}: {n_ev[s]} evaluations, per level (coarse->fine) {[int((levels[s] == L).sum()) for L in range(wl.py_stop, -1, -1)]}")
np.savez_compressed(f"/tmp/traces_{name}.npz", levels=levels, n_ev=n_ev)
