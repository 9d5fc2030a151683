#!/usr/bin/env python3
"""Wall time per pair of the tracker-driven frame loop (lk_sequence_run) on BASELINE config 4's
geometry for the three deformation descriptions (Eulerian / Lagrangian / strict Lagrangian)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402
from correlation_amd import tracker as tk  # noqa: E402
from correlation_amd.workload import C4 as wl  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ANNULAR = len(sys.argv) > 2 and sys.argv[2] == "annular"   # config 3's annulus: 4096^2, 8 x 32 sectors of ~35 k samples
size = 4096 if ANNULAR else wl.size
xs, ys, amps = ca.speckle.blobs(size, size, seed=7)
frames = []
for f in range(F):
    p = (0.8 * f, -0.4 * f, 1e-4 * f, 0.0, 0.0, 1e-4 * f)
    xd, yd = ca.speckle.deform(xs, ys, size, size, p)
    frames.append(np.ascontiguousarray(ca.speckle._render_torch(size, size, xd, yd, amps, 2.5, "cuda")))
c = size / 2.0
if os.environ.get("LK_PIN_FRAMES"):   # (tuning runs: page-locked frames - uploads return when enqueued)
    import ctypes as C
    lib = ca.load_library()
    for f in frames:
        assert lib.lk_pin_host_memory(C.c_void_p(f.ctypes.data), C.c_size_t(f.nbytes)) == 0
if ANNULAR:
    for name, mode, host_rebuild in (("eulerian", tk.DEF_EULERIAN, False), ("lagrangian", tk.DEF_LAGRANGIAN, False),
                                     ("lagrangian, lists moved on the host", tk.DEF_LAGRANGIAN, True),
                                     ("strict_lagrangian", tk.DEF_STRICT_LAGRANGIAN, False),
                                     ("strict_lagrangian, lists rebuilt on the host", tk.DEF_STRICT_LAGRANGIAN, True)):
        os.environ["LK_HOST_REWARP"] = "1" if host_rebuild else "0"
        e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
        t = tk.SequenceTracker(wl.model, tk.DOMAIN_ANNULAR, mode, tk.REF_PREVIOUS, tk.ERRMODE_CONTINUE, lib=e.lib)
        t.set_annular_domain(600.0, 1800.0, c, c, 8, 32)
        t.enable_report(False)
        t0 = time.perf_counter()
        done = tk.run_sequence(e, t, frames)
        wall = time.perf_counter() - t0
        r = t.results()
        print(json.dumps({"domain": "annulus 600..1800, 8 x 32 sectors, 4096^2", "mode": name, "pairs": done,
                          "ms_per_pair_wall_incl_setup": 1e3 * wall / done, "samples": int(r["number_of_points"].sum()),
                          "error_free_fraction": float((r["error_code"] == 0).mean()),
                          "median_u_last": float(np.nanmedian(r["resulting_parameters"][:, 0]))}), flush=True)
        e.close(), t.close()
    sys.exit(0)
for name, mode, ref, report, host_rebuild in (
        ("eulerian/first", tk.DEF_EULERIAN, tk.REF_FIRST, False, False),
        ("eulerian/first + CSV report", tk.DEF_EULERIAN, tk.REF_FIRST, True, False),
        ("lagrangian/previous", tk.DEF_LAGRANGIAN, tk.REF_PREVIOUS, False, False),
        ("strict_lagrangian/previous", tk.DEF_STRICT_LAGRANGIAN, tk.REF_PREVIOUS, False, False),
        ("strict_lagrangian/previous, lists rebuilt on the host", tk.DEF_STRICT_LAGRANGIAN, tk.REF_PREVIOUS, False, True)):
    if os.environ.get("LK_ONLY_FIRST") and name != "eulerian/first":   # (tuning runs: the windowed loop alone)
        continue
    os.environ["LK_HOST_REWARP"] = "1" if host_rebuild else "0"
    e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
    t = tk.SequenceTracker(wl.model, tk.DOMAIN_RECT, mode, ref, tk.ERRMODE_CONTINUE, lib=e.lib)
    t.set_rect_domain(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, c, c, wl.hs, wl.vs)
    t.enable_report(report)
    t0 = time.perf_counter()
    done = tk.run_sequence(e, t, frames)
    wall = time.perf_counter() - t0
    r = t.results()
    print(json.dumps({"mode": name, "pairs": done, "ms_per_pair_wall": 1e3 * wall / done,
                      "error_free_fraction": float((r["error_code"] == 0).mean()),
                      "median_u_last": float(np.nanmedian(r["resulting_parameters"][:, 0]))}), flush=True)
    e.close(), t.close()
