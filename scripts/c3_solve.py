#!/usr/bin/env python3
"""Solve time of config 3 (4096^2: 8 x 32 annular sectors of ~35 k samples + one 4.2 M-sample
blob): the two size classes solve on their own streams (LK_CLASS_STREAMS=0: one after the other)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402

truth = (1.1, 0.6, 0.0008, 0.0004, -0.0004, 0.0012)
und, dfm = ca.speckle.speckle_pair(4096, 4096, p=truth, seed=11, device="cuda")
rs, as_, ri, ro = 8, 32, 600.0, 1800.0
dr, da = np.float32((ro - ri) / rs), np.float32(2 * np.pi) / np.float32(as_)
params = np.float32([[ri + i * dr, dr, j * da, da, 2048.0, 2048.0] for i in range(rs) for j in range(as_)])
t = 2 * np.pi * np.arange(64) / 64
rad = np.where(np.arange(64) % 2 == 0, 1500.0, 900.0)
blob = np.stack([2048 + rad * np.cos(t), 2048 + rad * np.sin(t)], 1).astype(np.float32)
for what in ("annulus", "blob", "both"):
    e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    n = 0
    if what != "blob":
        e.set_sectors_annular(0, params, as_)
        n = len(params)
    if what != "annulus":
        e.resetPolygon_blob(n, blob)
    e.commit_sectors()
    r = e.correlate_all()
    ms = []
    for _ in range(10):
        r = e.correlate_all()
        ms.append(e.stats()["solve_ms"])
    u_true = truth[0] + truth[2] * (r["und_cx"] - 2048) + truth[3] * (r["und_cy"] - 2048)
    print(f"{what:8s}: solve {np.median(ms):.3f} ms (min {min(ms):.3f}), {len(r)} sectors, max |u - truth| {np.abs(r['p'][:, 0] - u_true).max():.4f} px, "
          f"errors {(r['error_code'] != 0).sum()}")
    e.close()
