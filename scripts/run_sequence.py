#!/usr/bin/env python3
"""BASELINE config 4 on one GPU: a 2048^2 sequence (constant velocity 0.8,-0.4 px/frame + 1e-4
dilation per frame), 224x224 sectors of 7x7, Eulerian / first-image reference, so the
constant-velocity initial guess is active from frame 2 on.  The next frame is uploaded on the
engine's own stream while the current pair is solved.  Prints per-frame solve times."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402
from correlation_amd.workload import C4 as wl  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
xs, ys, amps = ca.speckle.blobs(wl.size, wl.size, seed=7)


def frame(f):
    p = (0.8 * f, -0.4 * f, 1e-4 * f, 0.0, 0.0, 1e-4 * f)
    xd, yd = ca.speckle.deform(xs, ys, wl.size, wl.size, p)
    return ca.speckle._render_torch(wl.size, wl.size, xd, yd, amps, 2.5, "cuda")


frames = [frame(f) for f in range(F)]
e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
e.set_undeformed_image(frames[0])
e.set_deformed_image(frames[1])
e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
e.commit_sectors()
c = wl.size / 2.0
rows = []
t0 = time.perf_counter()
for f in range(1, F):
    if f + 1 < F:
        e.set_next_image(frames[f + 1])          # prefetch on the engine's nxt stream
    e.adjust_initial_guess(f - 1, True, np.zeros(6, np.float32), (c, c))
    r = e.correlate_all(None)
    st = e.stats()
    u_true = 0.8 * f + 1e-4 * f * (r["und_cx"] - c)
    rows.append({"frame": f, "solve_ms": st["solve_ms"], "evals_per_sector": st["evaluations"] / st["sectors"],
                 "errors": int((r["error_code"] != 0).sum()),
                 "median_abs_u_err": float(np.nanmedian(np.abs(r["p"][:, 0] - u_true)))})
    if f + 1 < F:
        e.makeDefPyramidFromNxt()
wall = time.perf_counter() - t0
print(json.dumps({"config": wl.name, "frames": F, "pairs_per_s_wall": (F - 1) / wall, "per_frame": rows}))

# the same sequence through the tracker-driven frame loop (include/lk_tracker.h):
# manager bookkeeping + guesses on the host, records back per frame, CSV report optional
from correlation_amd import tracker as tk  # noqa: E402

for with_report in (False, True):
    e2 = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
    t = tk.SequenceTracker(wl.model, tk.DOMAIN_RECT, tk.DEF_EULERIAN, tk.REF_FIRST, tk.ERRMODE_CONTINUE, lib=e2.lib)
    t.set_rect_domain(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, c, c, wl.hs, wl.vs)
    t.enable_report(with_report)
    host_frames = [np.ascontiguousarray(f) for f in frames]
    t0 = time.perf_counter()
    done = tk.run_sequence(e2, t, host_frames)
    wall2 = time.perf_counter() - t0
    res = t.results()
    u_true = 0.8 * (F - 1) + 1e-4 * (F - 1) * (res["und_center_x"] - c)
    print(json.dumps({"tracker_frame_loop": True, "report": with_report, "pairs": done,
                      "pairs_per_s_wall": done / wall2, "ms_per_pair_wall": 1e3 * wall2 / done,
                      "median_abs_u_err_last": float(np.nanmedian(np.abs(res["resulting_parameters"][:, 0] - u_true))),
                      "report_bytes": len(t.report())}))
    e2.close(), t.close()
