#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE'S OWN objects (oracle/_ref).

Run in the build container only (needs /root/reference; `make -C oracle ref` first):
    python tests/golden/make_golden.py
Only the pieces of the reference that compile without third-party stand-ins are
available: ModelClass_* (model_class.cpp), polygonBlob_class (polygon_class.cpp) and the
helpers of parameters.cpp.  The vectors are inputs + the reference's outputs (data, not
source).  Everything else on the hot path has no reference-generated golden vector
("parity unpinned").
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import lk_oracle as lo  # noqa: E402


def star(cx, cy, r0, r1, n, phase=0.0):
    t = phase + 2 * np.pi * np.arange(n) / n
    r = np.where(np.arange(n) % 2 == 0, r1, r0)
    return np.stack([cx + r * np.cos(t), cy + r * np.sin(t)], 1).astype(np.float32)


def main():
    R = lo.ref_lib()
    if R is None:
        raise SystemExit("oracle/_ref/libref_pieces.so missing: run `make -C oracle ref` where "
                         "/root/reference exists")
    F = lo._fp
    rng = np.random.default_rng(123)

    # ---- warp model: 4 models x random samples / parameters ---------------------------------
    model_cases = {}
    for model, P in ((0, 1), (1, 2), (2, 3), (3, 6)):
        n = 257
        xy = np.ascontiguousarray(rng.integers(0, 2048, (n, 2)).astype(np.float32))
        xy[::7] += np.float32(0.25)  # non-integer samples too
        p = np.zeros(6, np.float32)
        p[:P] = (rng.standard_normal(P) * [3, 3, 0.01, 0.01, 0.01, 0.01][:P]).astype(np.float32)
        cx, cy = np.float32(1000.5), np.float32(987.25)
        dxy = np.zeros((n, 2), np.float32)
        dT = np.zeros((n, 2 * P), np.float32)
        got = R.ref_compute_model(model, n, F(xy), F(p), cx, cy, F(dxy), F(dT))
        assert got == P
        model_cases[f"m{model}_xy"] = xy
        model_cases[f"m{model}_p"] = p
        model_cases[f"m{model}_c"] = np.array([cx, cy], np.float32)
        model_cases[f"m{model}_def"] = dxy
        model_cases[f"m{model}_dT"] = dT
    np.savez_compressed(os.path.join(HERE, "ref_model.npz"), **model_cases)

    # ---- blob polygons ----------------------------------------------------------------------
    contours = {
        "triangle": np.array([[10.2, 11.7], [60.9, 20.1], [30.3, 70.8]], np.float32),
        "square_cw": np.array([[5, 5], [5, 40], [40, 40], [40, 5]], np.float32),
        "square_ccw": np.array([[5, 5], [40, 5], [40, 40], [5, 40]], np.float32),
        "concave_L": np.array([[10, 10], [90, 10], [90, 40], [40, 40], [40, 90], [10, 90]], np.float32),
        "star16": star(100.3, 90.6, 35.0, 80.0, 16, 0.1),
        "star64": star(300.0, 300.0, 150.0, 250.0, 64, 0.05),
        "bowtie_bad": np.array([[10, 10], [80, 80], [80, 10], [10, 80]], np.float32),
        "halfpix": np.array([[20.5, 20.5], [70.5, 25.5], [75.5, 60.5], [45.5, 80.5], [15.5, 55.5]], np.float32),
    }
    blob = {}
    for name, c in contours.items():
        c = np.ascontiguousarray(c, np.float32)
        n = R.ref_blob_points(F(c), len(c), None, 0)
        blob[f"{name}_contour"] = c
        blob[f"{name}_count"] = np.array([n], np.int64)
        if n > 0:
            out = np.zeros((n, 2), np.float32)
            R.ref_blob_points(F(c), len(c), F(out), n)
            blob[f"{name}_pts"] = out.astype(np.int16)  # integer valued; compact
        blob[f"{name}_center"] = np.array([R.ref_blob_center(F(c), len(c), 0),
                                           R.ref_blob_center(F(c), len(c), 1)], np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_blob.npz"), **blob)

    # ---- best_rotation_UVUxUyVxVy ------------------------------------------------------------
    ps = (rng.standard_normal((32, 6)) * 0.05).astype(np.float32)
    rot = np.array([R.ref_best_rotation(F(np.ascontiguousarray(p))) for p in ps], np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_rotation.npz"), p=ps, angle=rot)
    print("wrote ref_model.npz ref_blob.npz ref_rotation.npz")


if __name__ == "__main__":
    main()
