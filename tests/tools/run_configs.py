#!/usr/bin/env python3
"""Full-size runs of BASELINE configs C3 (4096^2, annular 8x32 sectors + 64-vertex blob) and
C5 (8192^2, 447x447 sectors of 17x17, 4 levels) on ONE GPU, with oracle parity on a subset.
Prints one JSON line per config."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import correlation_amd as ca  # noqa: E402
from oracle import lk_oracle as lo  # noqa: E402

which = sys.argv[1:] or ["C3", "C5"]


def strict(a, b):
    dp = np.abs(a["p"] - b["p"])
    chi = np.abs(a["chi"] - b["chi"]) / np.abs(b["chi"])
    return float(((dp[:, :2] <= 1e-4).all(1) & (dp[:, 2:] <= 1e-6).all(1) & (chi <= 1e-5)
                  & (a["iterations"] == b["iterations"])).mean())


def timed(e, n=5):
    g = np.zeros(6, np.float32)
    r = e.correlate_all(g)
    ms = []
    for _ in range(n):
        e.correlate_all(g)
        ms.append(e.stats()["solve_ms"])
    return r, float(np.median(ms)), e.stats()


if "C3" in which:
    truth = (1.1, 0.6, 0.0008, 0.0004, -0.0004, 0.0012)
    und, dfm = ca.speckle.speckle_pair(4096, 4096, p=truth, seed=11, device="cuda")
    e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    rs, as_, ri, ro, cx, cy = 8, 32, 600.0, 1800.0, 2048.0, 2048.0
    dr, da = np.float32((ro - ri) / rs), np.float32(2 * np.pi) / np.float32(as_)
    t0 = time.perf_counter()
    s = 0
    for i in range(rs):
        for j in range(as_):
            e.resetPolygon_annular(s, np.float32(ri + i * dr), dr, np.float32(j) * da, da, cx, cy, as_)
            s += 1
    t = 2 * np.pi * np.arange(64) / 64
    rad = np.where(np.arange(64) % 2 == 0, 1500.0, 900.0)
    contour = np.stack([2048 + rad * np.cos(t), 2048 + rad * np.sin(t)], 1).astype(np.float32)
    e.resetPolygon_blob(s, contour)
    e.commit_sectors()
    t_roi = time.perf_counter() - t0
    r, ms, st = timed(e, 3)
    sizes = r["n_points"]
    # oracle on 6 annular sectors + a coarse check of the blob against ground truth
    pick = [0, 37, 100, 171, 230, 255]
    o1, o8 = lo.Oracle(n_threads=1), lo.Oracle(n_threads=8)
    for o in (o1, o8):
        o.set_image(0, und)
        o.set_image(1, dfm)
    lists = [e.getUndXY0ToCPU(k) for k in pick]
    want, self8 = o1.correlate_sectors(lists), o8.correlate_sectors(lists)
    u_true = truth[0] + truth[2] * (r["und_cx"] - 2048) + truth[3] * (r["und_cy"] - 2048)
    print(json.dumps({"config": "C3", "sectors": int(len(r)), "samples_total": int(sizes.sum()),
                      "largest_sector": int(sizes.max()), "roi_setup_s": t_roi, "solve_ms": ms,
                      "point_iterations_per_s": st["point_iterations"] / (ms * 1e-3),
                      "algorithmic_GBps": st["algorithmic_bytes"] / (ms * 1e-3) / 1e9,
                      "errors": int((r["error_code"] != 0).sum()),
                      "max_abs_u_minus_truth": float(np.abs(r["p"][:, 0] - u_true).max()),
                      "oracle_subset_max_dp01": float(np.abs(r["p"][pick][:, :2] - want["p"][:, :2]).max()),
                      "oracle_subset_max_rel_dchi": float((np.abs(r["chi"][pick] - want["chi"]) / want["chi"]).max()),
                      "oracle_self_T8_max_dp01": float(np.abs(self8["p"][:, :2] - want["p"][:, :2]).max()),
                      "iterations_equal": int((r["iterations"][pick] == want["iterations"]).sum())}))
    e.close()

if "C5" in which:
    from correlation_amd.workload import C5 as wl
    und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=(1.3, -0.7, 0.0005, 0.0, 0.0, -0.00025), seed=13,
                                       device="cuda")
    e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    t0 = time.perf_counter()
    e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
    e.commit_sectors()
    t_roi = time.perf_counter() - t0
    r, ms, st = timed(e, 3)
    o1, o8 = lo.Oracle(n_threads=1, py_stop=wl.py_stop), lo.Oracle(n_threads=8, py_stop=wl.py_stop)
    for o in (o1, o8):
        o.set_image(0, und)
        o.set_image(1, dfm)
    xd, yd, cen = lo.rect_sector_geometry(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
    pick = np.random.default_rng(3).choice(len(cen), 1500, replace=False)
    lists = [lo.rect_points(cx - xd, cy - yd, cx + xd, cy + yd) for cx, cy in cen[pick]]
    want = o1.correlate_sectors(lists, centers=cen[pick].astype(np.float32))
    self8 = o8.correlate_sectors(lists, centers=cen[pick].astype(np.float32))
    print(json.dumps({"config": "C5", "sectors": int(len(r)), "samples_per_sector": int(r["n_points"][0]),
                      "roi_setup_s": t_roi, "solve_ms": ms,
                      "point_iterations_per_s": st["point_iterations"] / (ms * 1e-3),
                      "algorithmic_GBps": st["algorithmic_bytes"] / (ms * 1e-3) / 1e9,
                      "errors": int((r["error_code"] != 0).sum()),
                      "evals_per_sector": st["evaluations"] / st["sectors"],
                      "strict_fraction_engine": strict(r[pick], want), "strict_fraction_reference_T8": strict(self8, want),
                      "max_dp01": float(np.abs(r["p"][pick][:, :2] - want["p"][:, :2]).max()),
                      "iterations_equal_fraction": float((r["iterations"][pick] == want["iterations"]).mean())}))
    e.close()
