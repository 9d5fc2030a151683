#!/usr/bin/env python3
"""Occupancy over time of a traced solve launch (scripts/trace_solve.py): trace_report.py trace.npz"""
import sys

import numpy as np

z = np.load(sys.argv[1])
t = z["trace"]
t0, t1, ev, steps, hw, xcc, cyc = (t[:, i].astype(np.int64) for i in range(7))
xc = xcc & 15
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
uid = (((xc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
s, e = (t0 - t0.min()) * 10e-3, (t1 - t0.min()) * 10e-3   # microseconds (100 MHz clock)
span, dur = e.max(), e - s
print(f"solve_ms {float(z['solve_ms']):.4f}; {len(t)} wavefronts on {len(np.unique(uid))} SIMDs; span {span:.1f} us")
print("wave duration us: min %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % (dur.min(), np.median(dur), np.quantile(dur, .9), np.quantile(dur, .99), dur.max()))
print("steps per wave: median %d p90 %d max %d;  shader cycles per step: median %d;  share of cycles inside evaluate %.3f;  clock %.2f GHz"
      % (np.median(steps), np.quantile(steps, .9), steps.max(), np.median(cyc / steps), ev.sum() / cyc.sum(), np.median(cyc / np.maximum(dur, 1e-3)) / 1e3))
late = s > 5.0
print("waves starting later than 5 us: %d; their start us: " % late.sum(), np.round(np.quantile(s[late], [0, .25, .5, .75, 1]), 1) if late.any() else "-")
print("   t us  live  SIMDs busy   with 1 / 2 / 3 / 4+ waves")
for x in np.arange(0, span, span / 24):
    live = (s <= x) & (e > x)
    u, c = np.unique(uid[live], return_counts=True)
    print(" %6.1f %5d %6d      %4d %4d %4d %4d" % (x, live.sum(), len(u), (c == 1).sum(), (c == 2).sum(), (c == 3).sum(), (c >= 4).sum()))
fin = np.zeros(uid.max() + 1)
np.maximum.at(fin, uid, e)
fin = fin[fin > 0]
print("per-SIMD finish us: p10 %.1f median %.1f p90 %.1f max %.1f" % tuple(np.quantile(fin, [.1, .5, .9, 1])))
w = np.zeros(uid.max() + 1)
np.add.at(w, uid, steps)
print("steps per SIMD: min %d median %d max %d;  corr(wave duration, steps) %.2f" % (w[w > 0].min(), np.median(w[w > 0]), w.max(), np.corrcoef(dur, steps)[0, 1]))
