#!/usr/bin/env python3
"""How much of the C2 solve time is the tail of the launch?  Solves the first N sectors of the
C2 grid for several N (the group size can be forced with LK_FORCE_GROUP) and prints
time, time per sector and the number of wavefronts of the launch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402
from correlation_amd.workload import C2 as wl  # noqa: E402

und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7, device="cuda")
counts = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 6144, 8192, 9000, 10000]
g = np.zeros(6, np.float32)
for n in counts:
    e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs, 0, n)
    e.commit_sectors()
    e.correlate_all(g)
    ms = []
    for _ in range(15):
        e.correlate_all(g)
        ms.append(e.stats()["solve_ms"])
    st = e.stats()
    m = float(np.median(ms))
    print(f"sectors {n:6d}  solve_ms {m:.4f}  ns/sector {1e6 * m / n:.1f}  evals/sector {st['evaluations'] / n:.2f}", flush=True)
    e.close()
