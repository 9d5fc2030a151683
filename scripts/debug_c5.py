import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca
from oracle import lk_oracle as lo
und, dfm = ca.speckle.speckle_pair(1024, 1024, p=(1.3, -0.7, 0.0005, 0.0, 0.0, -0.00025), seed=13)
for stop in (2, 3):
    e = ca.HipCorrelationEngine(py_stop=stop)
    e.set_undeformed_image(und); e.set_deformed_image(dfm)
    e.set_rect_grid(32.0, 32.0, 991.0, 991.0, 53, 53)
    e.commit_sectors()
    r = e.correlate_all(np.zeros(6, np.float32))
    o = lo.Oracle(py_stop=stop); o.set_image(0, und); o.set_image(1, dfm)
    xd, yd, cen = lo.rect_sector_geometry(32.0, 32.0, 991.0, 991.0, 53, 53)
    print("stop", stop, "xdim", xd, "n", r["n_points"][0], [e.sector_level_count(0, l) for l in range(stop + 1)])
    pick = np.arange(0, len(cen), 7)
    lists = [lo.rect_points(cx - xd, cy - yd, cx + xd, cy + yd) for cx, cy in cen[pick]]
    want = o.correlate_sectors(lists, centers=cen[pick].astype(np.float32))
    dp = np.abs(r["p"][pick] - want["p"])
    print("  max dp", dp.max(0), "iter eq", (r["iterations"][pick] == want["iterations"]).mean(), "err eq", (r["error_code"][pick] == want["error_code"]).mean())
    bad = np.argmax(dp[:, 0])
    print("  worst", pick[bad], r[pick[bad]], want[bad])
    # per-level evaluation check on the worst sector
    s = int(pick[bad])
    for lvl in range(stop, -1, -1):
        lxy = lists[bad] if lvl == 0 else lo.decimate(lists[bad], lvl)
        A, b, chi, err = e.evaluate(s, lvl, np.zeros(6, np.float32))
        cxl = np.float32(cen[s, 0]) * np.float32(1.0 / (1 << lvl)); cyl = np.float32(cen[s, 1]) * np.float32(1.0 / (1 << lvl))
        Ao, bo, chio, erro = lo.evaluate(2, 3, o.get_level(0, lvl), o.get_level(1, lvl), lxy, cxl, cyl, np.zeros(6, np.float32))
        print("   level", lvl, "n", len(lxy), "chi", chi, chio, "err", err, erro, "maxdA", np.abs(np.triu(A) - np.triu(Ao)).max())
    e.close()
