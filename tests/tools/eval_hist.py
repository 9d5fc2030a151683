import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import correlation_amd as ca
from oracle import lk_oracle as lo
und, dfm = ca.speckle.speckle_pair(2048, 2048, seed=7)
o = lo.Oracle(); o.set_image(0, und); o.set_image(1, dfm)
xdim, ydim, cen = lo.rect_sector_geometry(24.0, 24.0, 2023.0, 2023.0, 100, 100)
pick = np.arange(0, 10000, 5)
ev = []
for s in pick:
    cx, cy = cen[s]
    xy = lo.rect_points(cx - xdim, cy - ydim, cx + xdim, cy + ydim)
    r, tr = o.newton_raphson([0]*6, xy, center=(float(cx), float(cy)), trace_cap=256)
    lv = tr['level']
    ev.append([np.sum(lv == 2), np.sum(lv == 1), np.sum(lv == 0)])
ev = np.array(ev)
w = ev @ np.array([2, 7, 23]) + ev.sum(1) * 2.5   # passes (16 lanes) + ~solve/reduce cost in pass units
print("evals per level mean", ev.mean(0), "max", ev.max(0))
print("total evals: mean %.2f p50 %d p90 %d p99 %d max %d" % (ev.sum(1).mean(), *np.quantile(ev.sum(1), [.5,.9,.99]).astype(int), ev.sum(1).max()))
print("work units: mean %.1f p90 %.1f p99 %.1f max %.1f" % (w.mean(), *np.quantile(w, [.9,.99]), w.max()))
g = w[: len(w)//4*4].reshape(-1, 4)
print("per-wave (4 rows) max work: mean %.1f max %.1f ; sum-of-trips model" % (g.max(1).mean(), g.max(1).max()))
