#!/usr/bin/env python3
"""Records with and without the kept sums (LK_KEEP_SUMS): which sectors differ, and how (none must)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca
und, dfm = ca.speckle.speckle_pair(640, 640, p=(1.2, -0.6, 0.0004, 0.0, 0.0, -0.0002), seed=17)
def run(keep, ro, cap=None, ill=None):
    os.environ["LK_KEEP_SUMS"] = "1" if keep else "0"
    if cap is not None: os.environ["LK_EVAL_CAP"] = str(cap)
    if ill is not None: os.environ["LK_ILL_PASS"] = str(ill)
    e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY, py_stop=2)
    e.set_batch_invariant(True); e.set_reference_order(ro)
    e.set_undeformed_image(und); e.set_deformed_image(dfm)
    e.set_rect_grid(24.0, 24.0, 615.0, 615.0, 66, 66); e.commit_sectors()
    r = e.correlate_all(np.zeros(6, np.float32)); per = e.sector_stats(); e.close()
    return r, per
for ro in (0, 1):
  for cap in (20, 0):
    a, pa = run(True, ro, cap); b, pb = run(False, ro, cap)
    d = np.nonzero([x.tobytes() != y.tobytes() for x, y in zip(a, b)])[0]
    print("ref_order", ro, "cap", cap, "differing sectors", len(d), d[:10])
    for s in d[:4]:
        print("  ", s, "keep:", a[s]["p"][:2], a[s]["iterations"], a[s]["error_code"], pa[s], " nokeep:", b[s]["p"][:2], b[s]["iterations"], b[s]["error_code"], pb[s])
