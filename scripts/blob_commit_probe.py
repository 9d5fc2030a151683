#!/usr/bin/env python3
"""Registration + commit time of config 3's 64-vertex star blob (4.2 M samples) on fresh engines."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import correlation_amd as ca
ang = 2 * np.pi * np.arange(64) / 64
rad = np.where(np.arange(64) % 2 == 0, 1500.0, 900.0)
contour = np.stack([2048 + rad * np.cos(ang), 2048 + rad * np.sin(ang)], 1).astype(np.float32)
for it in range(3):
    e = ca.HipCorrelationEngine(py_stop=2)
    t0 = time.perf_counter()
    e.resetPolygon_blob(0, contour)
    t1 = time.perf_counter()
    e.commit_sectors()
    t2 = time.perf_counter()
    print(f"blob registration {1e3*(t1-t0):.2f} commit {1e3*(t2-t1):.2f}")
    e.close()
