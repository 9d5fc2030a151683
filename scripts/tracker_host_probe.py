#!/usr/bin/env python3
"""Host cost of the tracker's per-frame bookkeeping on config 4's grid (50 176 sectors), without
a GPU in the loop: lk_tracker_begin_frame / lk_tracker_end_frame with synthetic records, report
off and on."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402
from correlation_amd import tracker as tk  # noqa: E402

lib = ca.load_library()
for report in (False, True):
    t = tk.SequenceTracker(ca.FM_UVUXUYVXVY, tk.DOMAIN_RECT, tk.DEF_EULERIAN, tk.REF_FIRST, tk.ERRMODE_CONTINUE, lib=lib)
    t.set_rect_domain(24.0, 24.0, 2023.0, 2023.0, 1024.0, 1024.0, 224, 224)
    t.enable_report(report)
    S = t.n_sectors
    rng = np.random.default_rng(0)
    res = np.zeros(S, ca.RESULT_DTYPE)
    res["p"] = rng.normal(0, 1, (S, 6)).astype(np.float32)
    res["chi"], res["n_points"], res["iterations"] = 3.3, 49, 3
    tb = te = 0.0
    n = 20
    for k in range(n + 1):
        t0 = time.perf_counter()
        cmds, g = t.begin_frame(k)
        t1 = time.perf_counter()
        res["und_cx"], res["und_cy"] = cmds["center_x"], cmds["center_y"]
        t2 = time.perf_counter()
        t.end_frame(k, "a", "b", res)
        t3 = time.perf_counter()
        if k:
            tb += t1 - t0
            te += t3 - t2
    print(f"{S} sectors, report {'on' if report else 'off'}: begin_frame {tb / n * 1e3:.2f} ms, end_frame {te / n * 1e3:.2f} ms per frame "
          f"(through ctypes: begin_frame includes allocating its two output arrays)")
    t.close()
