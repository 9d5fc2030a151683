import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca
from oracle import lk_oracle as lo
und, dfm = ca.speckle.speckle_pair(1024, 1024, p=(1.3, -0.7, 0.0005, 0.0, 0.0, -0.00025), seed=13)
xd, yd, cen = lo.rect_sector_geometry(32.0, 32.0, 991.0, 991.0, 53, 53)
e = ca.HipCorrelationEngine(py_start=3, py_stop=3)
e.set_undeformed_image(und); e.set_deformed_image(dfm)
e.set_rect_grid(32.0, 32.0, 991.0, 991.0, 53, 53)
e.commit_sectors()
o = lo.Oracle(py_start=3, py_stop=3); o.set_image(0, und); o.set_image(1, dfm)
for s in (21, 14, 0):
    cx, cy = cen[s]
    xy = lo.rect_points(cx - xd, cy - yd, cx + xd, cy + yd)
    r, tr = o.newton_raphson([0]*6, xy, center=(float(cx), float(cy)), trace_cap=200)
    n = e.sector_level_count(s, 3)
    print("sector", s, "n3", n, "records", len(tr))
    for i, t in enumerate(tr[:12]):
        A, b, chi, err = e.evaluate(s, 3, t["p_in"])
        Ao = t["A"].reshape(6, 6)
        sameA = np.array_equal(np.triu(A), np.triu(Ao)); sameb = np.array_equal(b, t["b"])
        chis = np.float32(chi) * np.float32(1.0 / n)
        dp = e.damped_solve(np.triu(Ao) + np.triu(Ao, 1).T, t["b"], t["lam"], np.float32(1.0 / n))
        print("  rec", i, "kind", t["kind"], "A eq", sameA, "b eq", sameb, "chi eq", chis == t["chi"], "dp eq", np.array_equal(dp, t["dp"]),
              "max|ddp|", float(np.abs(dp - t["dp"]).max()), "lam", t["lam"])
e.close()
