"""Pins of the CPU oracle against what the reference itself provides in this image:
 - golden vectors produced by the reference's own ModelClass_* / polygonBlob_class objects
   (tests/golden/make_golden.py, built by oracle/Makefile `ref`);
 - the reference's literal 16x16 bicubic table, read as text when /root/reference exists.
Everything else of the hot path is "parity unpinned" (DESIGN.md): the reference has no
tests or fixtures and its CPU engine needs OpenCV + Eigen, absent from this image.
"""
import os
import re

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = "/root/reference"


def test_model_matches_reference_goldens(oracle):
    g = np.load(os.path.join(GOLD, "ref_model.npz"))
    for model, P in ((0, 1), (1, 2), (2, 3), (3, 6)):
        xy, p, c = g[f"m{model}_xy"], g[f"m{model}_p"], g[f"m{model}_c"]
        want_def, want_dT = g[f"m{model}_def"], g[f"m{model}_dT"]
        for k in range(len(xy)):
            xd, yd, dTx, dTy = oracle.model_point(model, float(xy[k, 0]), float(xy[k, 1]),
                                                  float(c[0]), float(c[1]), p)
            assert np.float32(xd) == want_def[k, 0] and np.float32(yd) == want_def[k, 1]
            assert np.array_equal(dTx[:P], want_dT[k, :P])
            assert np.array_equal(dTy[:P], want_dT[k, P:])


def test_blob_matches_reference_goldens(oracle):
    g = np.load(os.path.join(GOLD, "ref_blob.npz"))
    names = sorted({k.rsplit("_", 1)[0] for k in g.files if k.endswith("_contour")})
    assert "bowtie_bad" in names and "star64" in names
    for name in names:
        pts = oracle.blob_points(g[f"{name}_contour"])
        n = int(g[f"{name}_count"][0])
        if n < 0:
            assert pts is None, name  # error_bad_domain
            continue
        assert pts is not None and len(pts) == n, name
        if n:
            assert np.array_equal(pts, g[f"{name}_pts"].astype(np.float32)), name


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present on this box")
def test_bicubic_matrix_equals_reference_table(oracle):
    """The derived (exact-inverse) matrix equals the literal table of
    interpolation_class.cpp:539-558, parsed from the reference's text."""
    src = open(os.path.join(REF, "interpolation_class.cpp")).read()
    m = re.search(r"float\s+temp1\[256\]\s*=\s*\{(.*?)\};", src, re.S)
    assert m
    ref = np.array([float(t) for t in m.group(1).replace("\n", " ").split(",") if t.strip()], np.float32)
    assert ref.size == 256
    hdr = open(os.path.join(os.path.dirname(GOLD), "..", "oracle", "lk_bicubic_matrix.h")).read()
    start = hdr.index("LKO_BICUBIC_M[256] = {")
    body = hdr[hdr.index("{", start) + 1:hdr.rindex("}")]
    mine = np.array([float(t.replace("f", "")) for t in body.replace("\n", " ").split(",") if t.strip()],
                    np.float32)
    assert np.array_equal(mine, ref)


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref",
                                                    "libref_pieces.so")),
                    reason="partial reference build not present")
def test_oracle_vs_live_reference_pieces(oracle):
    """Fresh random cases straight against oracle/_ref (beyond the committed goldens)."""
    R = oracle.ref_lib()
    rng = np.random.default_rng(99)
    F = oracle._fp
    for _ in range(20):
        nv = int(rng.integers(3, 24))
        t = np.sort(rng.uniform(0, 2 * np.pi, nv))
        r = rng.uniform(20, 90, nv)
        c = np.ascontiguousarray(np.stack([120 + r * np.cos(t), 110 + r * np.sin(t)], 1), np.float32)
        n = R.ref_blob_points(F(c), nv, None, 0)
        mine = oracle.blob_points(c)
        if n < 0:
            assert mine is None
        else:
            out = np.zeros((max(n, 1), 2), np.float32)
            R.ref_blob_points(F(c), nv, F(out), n)
            assert mine is not None and len(mine) == n and np.array_equal(mine, out[:n])


def test_oracle_regression_vectors(oracle):
    """tests/golden/oracle_regression.npz freezes the oracle's own outputs (pyramid, bicubic known
    answers, evaluations, damped solves, Newton_Raphson traces incl. reject path and out-of-image,
    sample lists, a sector grid, a 3-frame sequence report): any change of the restatement or of
    its build flags shows up here bit for bit.  (Regression vectors - not reference-generated.)"""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_oracle_regression", os.path.join(here, "make_oracle_regression.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = np.load(os.path.join(here, "oracle_regression.npz"))
    got = mod.compute(stored=want)
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        a, b = np.asarray(got[k]), want[k]
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert a.tobytes() == b.tobytes(), f"{k}: oracle output changed"
