#!/usr/bin/env python3
"""Config 4: evaluations per sector (how long the longest chains of dependent steps are)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402
from correlation_amd.workload import C4 as wl  # noqa: E402

und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7)
for (a, b) in ((2, 2), (1, 2), (0, 2)):
    e = ca.HipCorrelationEngine(fitting_model=wl.model, py_start=a, py_stop=b)
    e.set_batch_invariant(True)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
    e.commit_sectors()
    r = e.correlate_all(np.zeros(6, np.float32))
    st = e.stats()
    ev = e.sector_stats()[:, 0]
    print(f"levels {b}..{a}: evaluations/sector mean {ev.mean():.2f}, percentiles 50/90/99/99.9/100 =",
          np.percentile(ev, [50, 90, 99, 99.9, 100]), f"sectors > 32: {(ev > 32).sum()}, > 64: {(ev > 64).sum()}, > 128: {(ev > 128).sum()};",
          f"solve {st['solve_ms']:.3f} ms")
    e.close()
