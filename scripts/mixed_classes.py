#!/usr/bin/env python3
"""A batch that mixes sector sizes (one launch chain per size class): solve time with the
classes on their own streams (default) and one after the other (LK_CLASS_STREAMS=0)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402
from correlation_amd.workload import C2 as wl  # noqa: E402

cache = f"/tmp/speckle_{wl.size}.npz"
if os.path.exists(cache):
    z = np.load(cache)
    und, dfm = z["und"], z["dfm"]
else:
    und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7)
    np.savez(cache, und=und, dfm=dfm)
rng = np.random.default_rng(3)
e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=2)
e.set_undeformed_image(und)
e.set_deformed_image(dfm)
s = 0
for half, count in ((7, 400), (20, 120), (60, 24), (150, 6)):
    for _ in range(count):
        cx, cy = rng.integers(half + 30, wl.size - half - 30, 2)
        e.resetPolygon_rect(s, int(cx - half), int(cy - half), int(cx + half), int(cy + half))
        s += 1
e.commit_sectors()
r = e.correlate_all()
ms = []
for _ in range(20):
    e.correlate_all()
    ms.append(e.stats()["solve_ms"])
print(f"{s} sectors of 15^2 / 41^2 / 121^2 / 301^2 samples: solve {np.median(ms):.3f} ms (min {min(ms):.3f}), errors {(r['error_code'] != 0).sum()}")
