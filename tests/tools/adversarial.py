#!/usr/bin/env python3
"""Adversarial inputs against the oracle: flat (textureless) patches, saturated regions, sectors
touching the validity border, far-off guesses, tiny precision.  Prints where engine and oracle
disagree beyond the oracle's own 1-vs-8-thread noise."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import correlation_amd as ca  # noqa: E402
from oracle import lk_oracle as lo  # noqa: E402

rng = np.random.default_rng(11)
und, dfm = ca.speckle.speckle_pair(512, 512, p=(1.3, -0.7, 0.002, 0.0, 0.0, -0.001), seed=7)
und, dfm = und.copy(), dfm.copy()
und[100:180, 100:180] = 37            # flat patch in both
dfm[100:180, 100:180] = 37
und[300:360, 60:140] = 255            # saturated
dfm[300:360, 60:140] = 255
und[200:260, 300:380] = (rng.integers(0, 2, (60, 80)) * 255).astype(np.uint8)   # binary noise, uncorrelated
dfm[200:260, 300:380] = (rng.integers(0, 2, (60, 80)) * 255).astype(np.uint8)

cases = []
for (x0, y0, x1, y1, label) in ((110, 110, 170, 170, "flat 61x61"), (120, 120, 128, 128, "flat 9x9"), (90, 90, 130, 130, "half flat"),
                                (310, 70, 350, 130, "saturated"), (210, 310, 250, 370, "uncorrelated noise"), (3, 3, 40, 40, "at the border"),
                                (470, 470, 508, 508, "at the far border"), (240, 240, 262, 262, "normal 23x23"), (250, 100, 256, 106, "normal 7x7")):
    cases.append((lo.rect_points(x0, y0, x1, y1), ((x0 + x1) * 0.5, (y0 + y1) * 0.5), label))
guesses = np.zeros((len(cases), 6), np.float32)
for model, interp, prec, guess0 in ((3, 2, 1e-3, 0.0), (3, 2, 1e-6, 0.0), (1, 1, 1e-3, 0.0), (3, 2, 1e-3, 25.0), (2, 0, 1e-3, 0.0)):
    g = guesses.copy()
    g[:, 0] = guess0
    e = ca.HipCorrelationEngine(fitting_model=model, interpolation=interp, precision=prec)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    for s, (pts, cen, _) in enumerate(cases):
        e.set_sector_points(s, pts, center=cen)
    e.commit_sectors()
    got = e.correlate_all(g)
    st = e.stats()
    e.close()
    res = {}
    for T in (1, 8):
        o = lo.Oracle(model=model, interp=interp, precision=prec, n_threads=T)
        o.set_image(0, und)
        o.set_image(1, dfm)
        res[T] = o.correlate_sectors([c[0] for c in cases], centers=np.array([c[1] for c in cases], np.float32), guesses=g)
    w, w8 = res[1], res[8]
    print(f"--- model {model} interp {interp} precision {prec} guess u0 {guess0}  (ill-conditioned solves {st['ill_conditioned_solves']})")
    for s, (_, _, label) in enumerate(cases):
        d = float(np.nanmax(np.abs(got["p"][s] - w["p"][s])[:2])) if np.isfinite(w["p"][s][:2]).all() and np.isfinite(got["p"][s][:2]).all() else float("nan")
        d8 = float(np.nanmax(np.abs(w8["p"][s] - w["p"][s])[:2])) if np.isfinite(w["p"][s][:2]).all() else float("nan")
        flag = "" if (got["error_code"][s] == w["error_code"][s] and (np.isnan(d) or d <= max(1e-3, 3 * d8))) else "   <<<"
        print(f"  {label:20s} err {got['error_code'][s]}/{w['error_code'][s]} it {got['iterations'][s]:3d}/{w['iterations'][s]:3d} "
              f"|dp01| {d:.2e} (oracle T8 {d8:.2e}) chi {got['chi'][s]:.4g}/{w['chi'][s]:.4g} nan {np.isnan(got['p'][s]).any()}/{np.isnan(w['p'][s]).any()}{flag}")
