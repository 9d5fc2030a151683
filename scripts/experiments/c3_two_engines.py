#!/usr/bin/env python3
"""Two engines, each with config 3's domain, solving at the same time (lk_set_pairs_in_flight(2): each takes half of the
workgroup slots for its team + persistent launches).  Checks that both finish, error-free, and prints the wall time per pair."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import correlation_amd as ca  # noqa: E402

truth = (1.1, 0.6, 0.0008, 0.0004, -0.0004, 0.0012)
und, dfm = ca.speckle.speckle_pair(4096, 4096, p=truth, seed=11, device="cuda")
rs, as_, ri, ro = 8, 32, 600.0, 1800.0
dr, da = np.float32((ro - ri) / rs), np.float32(2 * np.pi) / np.float32(as_)
t = 2 * np.pi * np.arange(64) / 64
rad = np.where(np.arange(64) % 2 == 0, 1500.0, 900.0)
engines = []
for k in range(2):
    e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY)
    e.set_pairs_in_flight(2)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    e.set_sectors_annular(0, np.float32([[np.float32(ri + i * dr), dr, np.float32(j) * da, da, 2048.0, 2048.0]
                                         for i in range(rs) for j in range(as_)]), as_)
    e.resetPolygon_blob(rs * as_, np.stack([2048 + rad * np.cos(t), 2048 + rad * np.sin(t)], 1).astype(np.float32))
    e.commit_sectors()
    engines.append(e)
g = np.zeros(6, np.float32)
out = [None, None]


def work(k, n):
    for _ in range(n):
        out[k] = engines[k].correlate_all(g)


for n in (1, 6):
    th = [threading.Thread(target=work, args=(k, n)) for k in range(2)]
    t0 = time.perf_counter()
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.perf_counter() - t0
print(f"two engines x {n} solves: {1e3 * dt / n:.3f} ms per round of two pairs; error-free {[(r['error_code'] == 0).mean() for r in out]}; "
      f"records equal {out[0].tobytes() == out[1].tobytes()}")
