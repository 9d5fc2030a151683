#!/usr/bin/env python3
"""Solve-kernel micro-benchmark on config C2 (no CPU baseline): N solve launches."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402
from correlation_amd.workload import C2, C4, C4B, C5  # noqa: E402

wl = {"C2": C2, "C4": C4, "C4B": C4B, "C5": C5}[sys.argv[1] if len(sys.argv) > 1 else "C2"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cache = f"/tmp/speckle_{wl.size}.npz"
if os.path.exists(cache):
    z = np.load(cache)
    und, dfm = z["und"], z["dfm"]
else:
    und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth if wl.size <= 2048 else (1.3, -0.7, 0.0005, 0.0, 0.0, -0.00025),
                                       seed=7 if wl.size <= 2048 else 13, device="cuda" if wl.size > 2048 else None)
    np.savez(cache, und=und, dfm=dfm)
e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop,
                            interpolation=int(os.environ.get("LK_INTERP", ca.IM_BICUBIC)))
e.set_reference_order(int(os.environ.get("LK_REF_ORDER", 0)))   # tuning: the reference-order mode's time
e.set_undeformed_image(und)
e.set_deformed_image(dfm)
grid = int(os.environ.get("LK_GRID", 0))   # tuning: another number of sectors of the same size
if grid:
    pitch = (wl.x_end - wl.x_begin + 1) / wl.hs
    e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_begin + grid * pitch - 1, wl.x_begin + grid * pitch - 1, grid, grid)
else:
    e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
e.commit_sectors()
g = np.zeros(6, np.float32)
r = e.correlate_all(g)
ms = []
for _ in range(n):
    e.correlate_all(g)
    ms.append(e.stats()["solve_ms"])
st = e.stats()
ms = np.array(ms)
print(f"{wl.name}" + (f" [grid {grid}x{grid}, {e.sector_info(0)[0]} samples per sector]" if grid else "") + f"\n solve_ms min {ms.min():.4f} median {np.median(ms):.4f}  pit/s {st['point_iterations'] / (np.median(ms) * 1e-3):.3e}"
      f"  alg GB/s {st['algorithmic_bytes'] / (np.median(ms) * 1e-3) / 1e9:.1f}  frac {st['algorithmic_bytes'] / (np.median(ms) * 1e-3) / 8e12:.4f}"
      f"  evals/sector {st['evaluations'] / st['sectors']:.2f}  ill {st['ill_conditioned_solves']}  errfree {(r['error_code'] == 0).mean():.4f}")
