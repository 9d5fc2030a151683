#!/usr/bin/env python3
"""One-off set-up cost of config 3's domains (4096^2: 8 x 32 annular sectors, 9.0 M samples; a 64-vertex
star blob, 4.2 M samples): registration + lk_commit_sectors, device masks (default) against the host
scans (LK_HOST_ROI=1).  Every measurement on a fresh engine, after one warm-up engine per path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import correlation_amd as ca  # noqa: E402

dr, da = (1800 - 600) / 8, 2 * np.pi / 32
params = np.float32([[600 + i * dr, dr, j * da, da, 2048.0, 2048.0] for i in range(8) for j in range(32)])
ang = 2 * np.pi * np.arange(64) / 64
rad = np.where(np.arange(64) % 2 == 0, 1500.0, 900.0)
contour = np.stack([2048 + rad * np.cos(ang), 2048 + rad * np.sin(ang)], 1).astype(np.float32)


def annulus(one_call):
    e = ca.HipCorrelationEngine(py_stop=2)
    t0 = time.perf_counter()
    if one_call:
        e.set_sectors_annular(0, params, 32)
    else:
        for s, q in enumerate(params):
            e.resetPolygon_annular(s, *[float(v) for v in q], 32)
    t1 = time.perf_counter()
    e.commit_sectors()
    t2 = time.perf_counter()
    n = sum(e.sector_info(s)[0] for s in range(256))
    e.close()
    return 1e3 * (t1 - t0), 1e3 * (t2 - t1), n


def blob():
    e = ca.HipCorrelationEngine(py_stop=2)
    t0 = time.perf_counter()
    e.resetPolygon_blob(0, contour)
    t1 = time.perf_counter()
    e.commit_sectors()
    t2 = time.perf_counter()
    n = e.sector_info(0)[0]
    e.close()
    return 1e3 * (t1 - t0), 1e3 * (t2 - t1), n


def both():
    e = ca.HipCorrelationEngine(py_stop=2)
    t0 = time.perf_counter()
    e.set_sectors_annular(0, params, 32)
    e.resetPolygon_blob(256, contour)
    e.commit_sectors()
    t1 = time.perf_counter()
    e.close()
    return 1e3 * (t1 - t0)


for label, host in (("device masks", "0"), ("host scans (LK_HOST_ROI=1)", "1")):
    os.environ["LK_HOST_ROI"] = host
    annulus(True), blob()   # warm-up: allocations, code objects
    r = annulus(False)
    print(f"{label}: annulus, 256 calls: registration {r[0]:.2f} ms, commit {r[1]:.2f} ms, samples {r[2]}")
    r = annulus(True)
    print(f"{label}: annulus, one call:  registration {r[0]:.2f} ms, commit {r[1]:.2f} ms")
    r = blob()
    print(f"{label}: blob: registration {r[0]:.2f} ms, commit {r[1]:.2f} ms, samples {r[2]}")
    print(f"{label}: config 3 (annulus + blob) registration + commit {both():.2f} ms")
