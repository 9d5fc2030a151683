#!/usr/bin/env python3
"""Config 4: does the effort a sector needed on the coarse levels predict its effort on level 0?  (No: correlation
-0.23 - ordering the lane-group launch by it would not shorten its tail.)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca
from correlation_amd.workload import C4 as wl
und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7)
ev = {}
for (a, b) in ((1, 2), (0, 2)):
    e = ca.HipCorrelationEngine(fitting_model=wl.model, py_start=a, py_stop=b)
    e.set_batch_invariant(True)
    e.set_undeformed_image(und); e.set_deformed_image(dfm)
    e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
    e.commit_sectors()
    e.correlate_all(np.zeros(6, np.float32))
    ev[a] = e.sector_stats()[:, 0].astype(np.int64)
    e.close()
upper, l0 = ev[1], ev[0] - ev[1]
print("corr(evals on levels 2..1, evals on level 0) = %.3f" % np.corrcoef(upper, l0)[0, 1])
for thr in (20, 30, 40):
    m = upper > thr
    print(f"  sectors with > {thr} evaluations above level 0: {m.sum():6d}; their level-0 evaluations mean {l0[m].mean():.1f} p99 {np.percentile(l0[m], 99):.0f} max {l0[m].max()};  the others: mean {l0[~m].mean():.1f} p99 {np.percentile(l0[~m], 99):.0f} max {l0[~m].max()}")
print("level-0 evaluations: percentiles 50/90/99/99.9/100", np.percentile(l0, [50, 90, 99, 99.9, 100]))
long0 = l0 > 30
print("sectors with > 30 level-0 evaluations:", long0.sum(), "of which > 20 above level 0:", (long0 & (upper > 20)).sum())
