#!/usr/bin/env python3
"""Config 4 as a tracked sequence (2048^2 x N frames, 224 x 224 sectors, constant-velocity guesses):
per-frame error counts and timing.  c4_sequence_probe.py [frames] [hs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import correlation_amd as ca
from correlation_amd import tracker as tk
from correlation_amd.workload import C4 as w
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hs = int(sys.argv[2]) if len(sys.argv) > 2 else w.hs
frames = ca.speckle.speckle_sequence(w.size, w.size, n, velocity=(0.8, -0.4), dilation=1e-4, seed=7, device="cuda")
e = ca.HipCorrelationEngine(fitting_model=w.model, py_stop=w.py_stop)
t = tk.SequenceTracker(w.model, tk.DOMAIN_RECT, tk.DEF_EULERIAN, tk.REF_FIRST, tk.ERRMODE_CONTINUE, lib=e.lib)
t.set_rect_domain(w.x_begin, w.x_begin, w.x_end, w.x_end, 1023.5, 1023.5, hs, hs)
t0 = time.perf_counter()
done = tk.run_sequence(e, t, frames)
dt = time.perf_counter() - t0
S = hs * hs
print(f"{done} pairs in {dt*1e3:.1f} ms = {dt*1e3/done:.2f} ms per pair, {S} sectors")
rows = t.report().split("\n")
head = rows[0].split(",")
ce, cu = head.index("error_code"), head.index("parameter_0")
for k in list(range(0, done, 8)) + [done - 1]:
    blk = rows[1 + k * S:1 + (k + 1) * S]
    codes = np.array([int(r.split(",")[ce]) for r in blk])
    u = np.array([float(r.split(",")[cu]) for r in blk])
    print(f"pair {k:2d}: error codes {np.bincount(codes, minlength=4).tolist()}  median u {np.nanmedian(u):.3f} (truth {0.8*(k+1):.1f})  nan {int(np.isnan(u).sum())}")
