#!/usr/bin/env python3
"""BASELINE config 1 on the GPU: 512x512 pair, ONE rectangular sector of 201x201 samples (the
reference's CPU-runnable case), rigid translation and affine.  Solve time of a single sector
is pure critical path; LK_FORCE_TEAM=w spreads it over w workgroups."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402

und, dfm = ca.speckle.speckle_pair(512, 512, p=(1.3, -0.7, 0.002, 0.0, 0.0, -0.001), seed=7)
for model in (ca.FM_UV, ca.FM_UVUXUYVXVY):
    e = ca.HipCorrelationEngine(fitting_model=model)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    e.resetPolygon_rect(0, 156, 156, 356, 356)
    e.commit_sectors()
    g = np.zeros(6, np.float32)
    r = e.correlate_all(g)
    ms = []
    for _ in range(10):
        e.correlate_all(g)
        ms.append(e.stats()["solve_ms"])
    st = e.stats()
    print(f"model {model}: solve_ms {np.median(ms):.4f}  evals {st['evaluations']}  p {r['p'][0][:2]}  chi {r['chi'][0]:.4f}")
    e.close()
