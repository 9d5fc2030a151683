#!/usr/bin/env python3
"""Freeze the CPU oracle's own outputs on small fixed inputs -> tests/golden/oracle_regression.npz.

These are REGRESSION vectors, not reference-generated ones: they pin oracle/lk_oracle.c against
accidental change (compiler flags, refactors), in the shape of SURVEY.md section 8c's list
(pyramid, bicubic known answers, one evaluation per model x interpolator, damped solves, full
Newton_Raphson traces incl. a reject path and an out-of-image case, sample-list hashes, a sector
grid, a 3-frame constant-velocity sequence).  The reference itself cannot be run here
("parity unpinned"); vectors made from the reference's own objects are in
ref_*.npz (make_golden.py).

    python tests/golden/make_oracle_regression.py          # rewrites the .npz
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import correlation_amd as ca  # noqa: E402  (speckle generator only)
from oracle import lk_oracle as lo  # noqa: E402


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()[:8], np.uint64)[0]


def compute(stored=None):
    """stored: a previously written .npz - its input images are reused, so the check does not
    depend on the speckle generator's libm."""
    out = {}
    rng = np.random.default_rng(2024)
    # (1) pyramid
    for name, shape in (("pyr_a", (64, 48)), ("pyr_b", (130, 98))):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        o = lo.Oracle(py_stop=3)
        o.set_image(0, img)
        out[name + "_src"] = img
        for lvl in (1, 2, 3):
            out[f"{name}_l{lvl}"] = o.get_level(0, lvl)
    # (2) bicubic known answers
    win = rng.integers(0, 256, (12, 12), dtype=np.uint8)
    out["kat_img"] = win
    out["kat_coeffs"] = np.stack([lo.bicubic_coeffs(win, ix, iy) for ix, iy in ((3, 3), (5, 4), (8, 7))])
    pts = (rng.random((40, 2)) * 7 + 2).astype(np.float32)
    out["kat_pts"] = pts
    for interp in (0, 1, 2):
        out[f"kat_sample_{interp}"] = np.array(lo.interpolate_many(interp, win, pts), np.float32)
    # (3) one evaluation: 21x21 sector, every model x interpolator
    if stored is not None:
        und, dfm = stored["ev_und"], stored["ev_def"]
    else:
        und, dfm = ca.speckle.speckle_pair(128, 128, p=(0.7, -0.4, 0.003, 0.001, -0.002, 0.002), seed=9)
    out["ev_und"], out["ev_def"] = und, dfm
    xy = lo.rect_points(50, 54, 70, 74)
    for model, p in ((0, [0.6]), (1, [0.6, -0.3]), (2, [0.6, -0.3, 0.002]), (3, [0.6, -0.3, 0.002, 0.001, -0.001, 0.003])):
        for interp in (0, 1, 2):
            A, b, chi, err = lo.evaluate(interp, model, und, dfm, xy, 60.0, 64.0, np.array(p, np.float32))
            out[f"ev_{model}_{interp}"] = np.concatenate([np.ravel(A), np.ravel(b), [chi, err]]).astype(np.float32)
    # (4) damped solves
    sol = []
    for k in range(6):
        n = 6
        H = rng.standard_normal((40, n)).astype(np.float32)
        A = (H.T @ H).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32)
        lam = np.float32(10.0 ** (k - 4))
        dp = lo.damped_solve(A.copy(), b.copy(), float(lam), 1.0 / 40)
        sol.append(np.concatenate([np.ravel(A), b, [lam], np.ravel(dp)]).astype(np.float32))
    out["solve_cases"] = np.stack(sol)
    # (5) Newton_Raphson traces: config 1 rigid / affine, a far guess (reject path), out of image
    if stored is not None:
        und, dfm = stored["nr_und"], stored["nr_def"]
    else:
        und, dfm = ca.speckle.speckle_pair(256, 256, p=(1.3, -0.7, 0.002, 0.0, 0.0, -0.001), seed=7)
    out["nr_und"], out["nr_def"] = und, dfm
    big = lo.rect_points(78, 78, 178, 178)
    cases = (("rigid", 1, [0, 0], big, (128.0, 128.0)), ("affine", 3, [0] * 6, big, (128.0, 128.0)),
             ("far", 3, [4.0, 3.0, 0, 0, 0, 0], big, (128.0, 128.0)),
             ("edge", 3, [0] * 6, lo.rect_points(230, 100, 252, 130), (241.0, 115.0)))
    for name, model, guess, pts, cen in cases:
        o = lo.Oracle(model=model)
        o.set_image(0, und)
        o.set_image(1, dfm)
        res, tr = o.newton_raphson(guess, pts, center=cen, trace_cap=512)
        out[f"nr_{name}_result"] = np.array([res], lo.RESULT_DTYPE).view(np.uint8)
        out[f"nr_{name}_trace_p"] = np.array(tr["p_in"], np.float32)
        out[f"nr_{name}_trace_chi"] = np.array(tr["chi"], np.float32)
        out[f"nr_{name}_trace_lambda"] = np.array(tr["lam"], np.float32)
        out[f"nr_{name}_trace_kind"] = np.array(tr["kind"] * 100 + tr["level"], np.int32)
    # (6) sample lists
    ann = lo.annular_points(30.0, 24.0, 0.4, 0.9, 128.0, 126.0, 7)
    t = 2 * np.pi * np.arange(9) / 9
    blob = lo.blob_points(np.stack([128 + 40 * np.cos(t) * (1 + 0.3 * (np.arange(9) % 2)), 126 + 33 * np.sin(t)], 1))
    out["lists"] = np.array([len(ann), digest(ann), len(blob), digest(blob)], np.uint64)
    # (7) 6x6 sector grid
    o = lo.Oracle()
    o.set_image(0, und)
    o.set_image(1, dfm)
    xdim, ydim, cen = lo.rect_sector_geometry(24.0, 24.0, 231.0, 231.0, 6, 6)
    lists = [lo.rect_points(cx - xdim, cy - ydim, cx + xdim, cy + ydim) for cx, cy in cen]
    out["grid_results"] = o.correlate_sectors(lists, centers=cen.astype(np.float32)).view(np.uint8)
    # (8) 3-frame constant-velocity sequence through the manager oracle
    from oracle import lk_manager_oracle as mo
    if stored is not None:
        frames = list(stored["seq_frames"])
    else:
        frames = ca.speckle.speckle_sequence(192, 192, 3, velocity=(0.9, -0.5), dilation=4e-4, seed=3)
    out["seq_frames"] = np.stack(frames)
    o = lo.Oracle()
    o.set_image(0, frames[0])
    o.set_image(1, frames[1])
    m = mo.ManagerOracle(o, 3, mo.DOMAIN_RECT, mo.DEF_EULERIAN, mo.REF_FIRST)
    m.set_rect_domain(30.0, 30.0, 161.0, 161.0, 95.5, 95.5, 2, 2)
    m.run_frame(0, "f0", "f1")
    o.set_image(2, frames[2])
    o.def_from_nxt()
    m.run_frame(1, "f0", "f2")
    out["sequence_report"] = np.frombuffer(m.report_text().encode(), np.uint8)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "oracle_regression.npz"), **compute())
    print("wrote", os.path.join(HERE, "oracle_regression.npz"))
