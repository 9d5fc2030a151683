import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca
from oracle import lk_oracle as lo
und, dfm = ca.speckle.speckle_pair(1024, 1024, p=(1.3, -0.7, 0.0005, 0.0, 0.0, -0.00025), seed=13)
xd, yd, cen = lo.rect_sector_geometry(32.0, 32.0, 991.0, 991.0, 53, 53)
pick = np.arange(0, len(cen), 7)
lists = [lo.rect_points(cx - xd, cy - yd, cx + xd, cy + yd) for cx, cy in cen[pick]]
for start in (3, 2, 1, 0):
    e = ca.HipCorrelationEngine(py_start=start, py_stop=3)
    e.set_undeformed_image(und); e.set_deformed_image(dfm)
    e.set_rect_grid(32.0, 32.0, 991.0, 991.0, 53, 53)
    e.commit_sectors()
    r = e.correlate_all(np.zeros(6, np.float32))
    o = lo.Oracle(py_start=start, py_stop=3); o.set_image(0, und); o.set_image(1, dfm)
    want = o.correlate_sectors(lists, centers=cen[pick].astype(np.float32))
    same = np.array([r[pick][i].tobytes() == want[i].tobytes() for i in range(len(pick))])
    dp = np.abs(r["p"][pick] - want["p"])
    print("levels 3..%d: bit-identical records %.3f, max dp %s, iter eq %.3f, chi eq %.3f" % (
        start, same.mean(), dp.max(0)[:2], (r["iterations"][pick] == want["iterations"]).mean(),
        (r["chi"][pick] == want["chi"]).mean()))
    if start == 3 and not same.all():
        i = int(np.argmin(same)); print("  first diff", pick[i], r[pick[i]], want[i])
    e.close()
