#!/usr/bin/env python3
"""One-off set-up cost of config 3's domains: ROI rasterisation on the host and lk_commit_sectors."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import correlation_amd as ca
e = ca.HipCorrelationEngine(py_stop=2)
dr, da = (1800 - 600) / 8, 2 * np.pi / 32
t0 = time.perf_counter()
for i in range(8):
    for j in range(32):
        e.resetPolygon_annular(i * 32 + j, 600 + i * dr, dr, j * da, da, 2048.0, 2048.0, 32)
t1 = time.perf_counter()
e.commit_sectors()
t2 = time.perf_counter()
print(f"annular rasterisation of 256 sectors {1e3*(t1-t0):.1f} ms, commit {1e3*(t2-t1):.1f} ms, samples {sum(e.sector_info(s)[0] for s in range(256))}")
e3 = ca.HipCorrelationEngine(py_stop=2)
params = np.float32([[600 + i * dr, dr, j * da, da, 2048.0, 2048.0] for i in range(8) for j in range(32)])
t0 = time.perf_counter()
e3.set_sectors_annular(0, params, 32)
t1 = time.perf_counter()
e3.commit_sectors()
t2 = time.perf_counter()
print(f"the same annulus in one call (lk_set_sectors_annular) {1e3*(t1-t0):.1f} ms, commit {1e3*(t2-t1):.1f} ms")
t0 = time.perf_counter()
e3.commit_sectors()
print(f"second commit of the same lists (buffers already allocated) {1e3*(time.perf_counter()-t0):.1f} ms")
e2 = ca.HipCorrelationEngine(py_stop=2)
ang = 2 * np.pi * np.arange(64) / 64
rad = np.where(np.arange(64) % 2 == 0, 1500.0, 900.0)
t0 = time.perf_counter()
e2.resetPolygon_blob(0, np.stack([2048 + rad * np.cos(ang), 2048 + rad * np.sin(ang)], 1).astype(np.float32))
t1 = time.perf_counter()
e2.commit_sectors()
t2 = time.perf_counter()
print(f"blob rasterisation {1e3*(t1-t0):.1f} ms, commit {1e3*(t2-t1):.1f} ms, samples {e2.sector_info(0)[0]}")
