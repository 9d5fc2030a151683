"""BASELINE configs 3 and 4 at FULL size through the C-ABI (round 1 ran them only from bench.py / scripts/):
 * config 3: 4096^2 pair, 8 x 32 annular sectors (9.0 M samples) + one 64-vertex blob (4.2 M samples,
   a team of up to 128 workgroups), lists made by the device ROI masks;
 * config 4: 2048^2 x 64 frames, 224 x 224 sectors, constant-velocity guesses, tracked by lk_sequence_run.
Checks: ground truth of the synthetic deformation, size-independent properties, and - in reference-order
mode - byte equality with the CPU oracle for all 256 annular sectors and for the 4.2 M-sample blob."""
import numpy as np
import pytest

import correlation_amd as ca
from correlation_amd.workload import C4

pytestmark = pytest.mark.gpu


def canonical(rec):
    a = np.array(rec, copy=True)
    for f in ("p", "chi", "und_cx", "und_cy"):
        v = a[f]
        v[np.isnan(v)] = np.float32(np.nan)
    return a


def test_config3_full_size(oracle):
    truth = (1.1, 0.6, 0.0008, 0.0004, -0.0004, 0.0012)
    und, dfm = ca.speckle.speckle_pair(4096, 4096, p=truth, seed=11, device="cuda")
    rs, as_, ri, ro = 8, 32, 600.0, 1800.0
    dr, da = np.float32((ro - ri) / rs), np.float32(2 * np.pi) / np.float32(as_)
    params = np.float32([[np.float32(ri + i * dr), dr, np.float32(j) * da, da, 2048.0, 2048.0]
                         for i in range(rs) for j in range(as_)])
    t = 2 * np.pi * np.arange(64) / 64
    rad = np.where(np.arange(64) % 2 == 0, 1500.0, 900.0)
    contour = np.stack([2048 + rad * np.cos(t), 2048 + rad * np.sin(t)], 1).astype(np.float32)

    def solve(reference_order):
        e = ca.HipCorrelationEngine()
        e.set_reference_order(1 if reference_order else 0)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        e.set_sectors_annular(0, params, as_)
        e.resetPolygon_blob(rs * as_, contour)
        e.commit_sectors()
        r = e.correlate_all(np.zeros(6, np.float32))
        st = e.stats()
        blob_xy = e.getUndXY0ToCPU(rs * as_) if reference_order else None
        e.close()
        return r, st, blob_xy

    fast, st, _ = solve(False)
    S = rs * as_ + 1
    assert len(fast) == S and (fast["error_code"] == 0).all()
    assert fast["n_points"][-1] == 4234328 and 9.0e6 < fast["n_points"][:-1].sum() < 9.1e6
    assert st["sectors"] == S and st["solve_ms"] < 20.0          # the blob alone took 38 ms before teams
    cx, cy = fast["und_cx"] - 2048.0, fast["und_cy"] - 2048.0
    u_true = truth[0] + truth[2] * cx + truth[3] * cy
    v_true = truth[1] + truth[4] * cx + truth[5] * cy
    assert np.abs(fast["p"][:, 0] - u_true).max() < 0.02 and np.abs(fast["p"][:, 1] - v_true).max() < 0.02
    assert np.abs(fast["p"][:, 2] - truth[2]).max() < 2e-4 and np.abs(fast["p"][:, 5] - truth[5]).max() < 2e-4
    # the star is symmetric about (2048, 2048); the reference's centre is the SEQUENTIAL float32 mean of 4.2 M
    # coordinates (pyramid_class.cpp:325-340), which drifts by a pixel or two at this size - reproduced, not fixed
    assert abs(fast["und_cx"][-1] - 2048.0) < 3.0 and abs(fast["und_cy"][-1] - 2048.0) < 3.0
    # reference-order mode: the CPU engine's summation order and solver on every level
    ref, _, blob_xy = solve(True)
    assert np.array_equal(ref["n_points"], fast["n_points"]) and np.array_equal(ref["und_cx"], fast["und_cx"])
    assert np.abs(fast["p"] - ref["p"])[:, :2].max() < 1e-3            # the fast mode deviates by summation order only
    # chi: one running float32 sum over n samples (the reference's, and the reference-order mode's) is itself off
    # by ~n * 2^-24 relative; the fast mode's tree of partial sums is the more accurate one (blob: 0.9 % apart)
    assert (np.abs(fast["chi"] - ref["chi"]) <= np.maximum(2e-3, 5e-9 * ref["n_points"]) * ref["chi"]).all()
    o = oracle.Oracle()
    o.set_image(0, und)
    o.set_image(1, dfm)
    lists = [oracle.annular_points(*[np.float32(v) for v in q], as_) for q in params]
    want = o.correlate_sectors(lists, nthreads=16)                     # centre = float mean of the samples
    assert np.array_equal(ref["n_points"][:-1], [len(x) for x in lists])
    assert canonical(ref[:-1]).tobytes() == canonical(want).tobytes()
    assert blob_xy.shape == (4234328, 2)
    want_blob = o.newton_raphson(np.zeros(6), blob_xy)                 # 4.2 M samples, ~10 s on one core
    assert canonical(np.array([ref[-1]])).tobytes() == canonical(np.array([want_blob])).tobytes()


def test_config4_default_mode_against_the_oracle_at_full_size(oracle):
    """All 50 176 sectors of config 4 (7 x 7 samples, two starved levels), default mode, against the CPU oracle with the
    1-thread summation order.  Starved levels are chaotic in the reference itself (the oracle with 8 thread chunks agrees with
    the oracle with 1 on 89 % of the iteration counts of a 3500-sector subset, tests/test_reference_order_gpu.py), so the
    bounds are the measured distance with margin - driver run of round 3 (bench.py, other_configs.C4_one_pair.parity_vs_cpu):
    17 error codes and 2 NaN flags differ, same iteration count on 94.4 %, |dp01| p50 7e-7 px, p99 0.031 px."""
    import os
    w = C4
    und, dfm = ca.speckle.speckle_pair(w.size, w.size, p=w.truth, seed=7)
    e = ca.HipCorrelationEngine(fitting_model=w.model, py_stop=w.py_stop)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    e.set_rect_grid(w.x_begin, w.x_begin, w.x_end, w.x_end, w.hs, w.vs)
    e.commit_sectors()
    r = e.correlate_all(np.zeros(6, np.float32))
    e.close()
    o = oracle.Oracle(interp=oracle.IM_BICUBIC, model=w.model, py_stop=w.py_stop)
    o.set_image(0, und)
    o.set_image(1, dfm)
    xd, yd, cen = oracle.rect_sector_geometry(w.x_begin, w.x_begin, w.x_end, w.x_end, w.hs, w.vs)
    n = (2 * xd + 1) * (2 * yd + 1)
    cat = np.concatenate([oracle.rect_points(cx - xd, cy - yd, cx + xd, cy + yd) for cx, cy in cen])
    want = o.correlate_packed(cat, np.arange(len(cen), dtype=np.int64) * n, np.full(len(cen), n, np.int32),
                              centers=cen.astype(np.float32), nthreads=os.cpu_count() or 1)
    assert len(r) == len(want) == 50176 and np.array_equal(r["n_points"], want["n_points"])
    nan_g, nan_w = np.isnan(r["p"]).any(1), np.isnan(want["p"]).any(1)
    both = ~nan_g & ~nan_w & (r["error_code"] == 0) & (want["error_code"] == 0)
    d = np.abs(r["p"][both][:, :2] - want["p"][both][:, :2]).max(1)
    assert (r["error_code"] != want["error_code"]).sum() <= 40
    assert (nan_g != nan_w).sum() <= 6
    assert (r["iterations"] == want["iterations"]).mean() >= 0.92
    assert np.percentile(d, 50) < 1e-5 and np.percentile(d, 99) < 0.06
    assert both.mean() > 0.995


def test_config4_sequence_of_64_frames(monkeypatch):
    from correlation_amd import tracker as tk
    w = C4
    frames = ca.speckle.speckle_sequence(w.size, w.size, 64, velocity=(0.8, -0.4), dilation=1e-4, seed=7, device="cuda")
    names = [f"frame{i:02d}.pgm" for i in range(64)]

    def run(sync):
        monkeypatch.setenv("LK_SEQ_SYNC", "1" if sync else "0")
        e = ca.HipCorrelationEngine(fitting_model=w.model, py_stop=w.py_stop)
        e.set_batch_invariant(True)
        t = tk.SequenceTracker(w.model, tk.DOMAIN_RECT, tk.DEF_EULERIAN, tk.REF_FIRST, tk.ERRMODE_CONTINUE, lib=e.lib)
        t.set_rect_domain(w.x_begin, w.x_begin, w.x_end, w.x_end, 1023.5, 1023.5, w.hs, w.vs)
        assert tk.run_sequence(e, t, frames, names) == 63
        text, res = t.report(), t.results()
        e.close(), t.close()
        return text, res

    text, res = run(False)            # the overlapped loop: pair k+1 launched from device guesses behind pair k's bookkeeping
    S = w.hs * w.vs
    assert len(res) == S and text.count("\n") == 1 + 63 * S
    ok = res["error_code"] == 0
    # 7 x 7-sample sectors with six parameters have singular coarse levels; a sector that diverges once is
    # handed its own bad result as the next guess (manager_class.cpp:2677-2694, "continue" policy) and stays
    # out of the image: 0.1 % of the sectors after the first pair, 23 % after 63 (scripts/c4_sequence_probe.py) -
    # the reference's semantics, mirrored by the tracker (tests/test_tracker_host.py), not an engine defect
    assert ok.mean() > 0.7
    # last pair: frame 0 against frame 63, 63 steps of (0.8, -0.4) px plus 63e-4 dilation about the centre
    cx, cy = res["und_center_x"] - 1024.0, res["und_center_y"] - 1024.0
    u_true, v_true = 0.8 * 63 + 63e-4 * cx, -0.4 * 63 + 63e-4 * cy
    du, dv = (res["resulting_parameters"][:, 0] - u_true)[ok], (res["resulting_parameters"][:, 1] - v_true)[ok]
    assert np.median(np.abs(du)) < 0.1 and np.median(np.abs(dv)) < 0.1      # (49-sample sectors: noisy, unbiased)
    assert (np.abs(du) < 1.0).mean() > 0.85
    # constant-velocity guesses (manager_class.cpp:2677-2686): 2 p(k-1) - p(k-2) lands on the new displacement,
    # a plain "previous result" guess would be one velocity step (0.8 px) short
    g = res["initial_guess"][:, 0][ok]
    assert np.median(np.abs(g - u_true[ok])) < 0.15
    # per-frame translation from the report: the first data column after the names is the frame number
    rows = text.split("\n")[1:-1]
    head = text.split("\n")[0].split(",")
    col = head.index("parameter_0") if "parameter_0" in head else None
    assert col is not None
    for k in (0, 1, 31, 62):
        u_k = np.array([float(r.split(",")[col]) for r in rows[k * S:(k + 1) * S:97]])
        assert abs(np.nanmedian(u_k) - 0.8 * (k + 1)) < 0.08 + 0.005 * (k + 1), k   # (a sparse sample of the rows; the dilation spreads u by +-1e-4 k x)
    # the one-pair-at-a-time loop gives the same report and frame_results, bit for bit
    text_sync, res_sync = run(True)
    assert text_sync == text and res_sync.tobytes() == res.tobytes()
