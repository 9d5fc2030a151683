#!/usr/bin/env python3
"""Config 2: evaluations, LM trips and re-evaluations (rejected trips) per sector - who are the stragglers?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402
from correlation_amd.workload import C2 as wl  # noqa: E402

os.environ["LK_KEEP_SUMS"] = "0"
und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7)
e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
e.set_undeformed_image(und)
e.set_deformed_image(dfm)
e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
e.commit_sectors()
e.correlate_all(np.zeros(6, np.float32))
per = e.sector_stats().astype(np.int64)
ev, pit = per[:, 0], per[:, 2]
re = ev - pit
print("evaluations per sector: mean %.2f, percentiles 50/90/99/99.9/100 =" % ev.mean(), np.percentile(ev, [50, 90, 99, 99.9, 100]))
print("re-evaluations per sector: total %d, sectors with any %d" % (re.sum(), (re > 0).sum()))
for lo in (14, 16, 18, 20, 24):
    m = ev >= lo
    print(f"  sectors with >= {lo} evaluations: {m.sum():5d}, their re-evaluations: mean {re[m].mean() if m.any() else 0:.2f} max {re[m].max() if m.any() else 0}")
top = np.argsort(-ev)[:12]
print("top sectors (evaluations, LM trips + levels, re-evaluations):", [(int(ev[s]), int(pit[s]), int(re[s])) for s in top])
