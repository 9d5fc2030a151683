#!/usr/bin/env python3
"""consecutive default-mode solves of the C4-like property test's pair: which records change between solves?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import correlation_amd as ca  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_parity_gpu import _smooth_texture  # noqa: E402

size, hs = 2048, 224
und = _smooth_texture(size, 5)
dfm = np.roll(und, shift=(-2, 3), axis=(0, 1))
e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY, py_stop=2)
e.set_undeformed_image(und)
e.set_deformed_image(dfm)
e.set_rect_grid(24.0, 24.0, size - 25.0, size - 25.0, hs, hs)
e.commit_sectors()
recs, stats = [], []
for i in range(6):
    recs.append(e.correlate_all(np.zeros(6, np.float32)))
    stats.append(e.sector_stats())
for i in range(1, 6):
    d = np.nonzero([recs[i][s].tobytes() != recs[0][s].tobytes() for s in range(len(recs[0]))])[0]
    print(f"solve {i} vs 0: {len(d)} records differ: {d[:12]}")
    for s in d[:4]:
        print("   sector", s, "p0", recs[0][s]["p"][:2], recs[i][s]["p"][:2], "err", recs[0][s]["error_code"], recs[i][s]["error_code"], "stats", stats[0][s], stats[i][s],
              "neighbours' evals", stats[0][s - s % 4:s - s % 4 + 4, 0], stats[i][s - s % 4:s - s % 4 + 4, 0], "ill", stats[0][s - s % 4:s - s % 4 + 4, 3])
print("lk_stats", e.stats())
