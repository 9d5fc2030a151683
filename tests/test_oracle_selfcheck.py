"""Self-consistency of the CPU oracle (runs anywhere, no GPU): properties that must hold
whatever the bits are - the oracle recovers a known deformation, its pieces agree with
independent float64 math, its control flow matches the reference's documented quirks."""
import numpy as np
import pytest


def catmull_rom_f64(img, x, y):
    ix, iy = int(x), int(y)
    fx, fy = x - ix, y - iy

    def w(t):
        return np.array([-0.5 * t**3 + t**2 - 0.5 * t, 1.5 * t**3 - 2.5 * t**2 + 1,
                         -1.5 * t**3 + 2 * t**2 + 0.5 * t, 0.5 * t**3 - 0.5 * t**2])
    P = img[iy - 1:iy + 3, ix - 1:ix + 3].astype(np.float64)
    return w(fy) @ P @ w(fx)


def test_bicubic_is_catmull_rom(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 48), dtype=np.uint8)
    for _ in range(200):
        x, y = rng.uniform(2, 44), rng.uniform(2, 36)
        w, wx, wy, err = oracle.interpolate(oracle.IM_BICUBIC, img, np.float32(x), np.float32(y))
        assert err == 0
        assert abs(w - catmull_rom_f64(img, float(np.float32(x)), float(np.float32(y)))) < 0.15
    # interpolates the data at integer positions (to float32 rounding of the monomial form)
    w, _, _, _ = oracle.interpolate(oracle.IM_BICUBIC, img, 10.0, 12.0)
    assert abs(w - float(img[12, 10])) < 0.1


def test_bicubic_coefficients_are_exact_quarter_integers(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (16, 16), dtype=np.uint8)
    # the separable factorisation used by the HIP kernel (lk_kernels.hip, cubic_1d)
    Cm = np.array([[2, -3, 3, -1], [-4, 9.5, -8, 2.5], [2.5, -7, 6.5, -2], [-0.5, 1.5, -1.5, 0.5]])
    for ix, iy in [(3, 3), (7, 9), (12, 5), (1, 1), (13, 13)]:
        a = oracle.bicubic_coeffs(img, ix, iy).astype(np.float64)
        assert np.all(a * 4 == np.round(a * 4))
        P = img[iy - 1:iy + 3, ix - 1:ix + 3].astype(np.float64)
        assert np.array_equal((Cm @ P @ Cm.T).reshape(-1), a)


def test_out_of_image_rule(oracle):
    img = np.full((20, 30), 7, np.uint8)
    assert oracle.interpolate(oracle.IM_BICUBIC, img, 1.0, 5.0)[3] == 1   # x > 1 strictly
    assert oracle.interpolate(oracle.IM_BICUBIC, img, 1.001, 5.0)[3] == 0
    assert oracle.interpolate(oracle.IM_BICUBIC, img, 28.0, 5.0)[3] == 1  # x < cols-2 strictly
    assert oracle.interpolate(oracle.IM_BICUBIC, img, 27.99, 17.99)[3] == 0
    assert oracle.interpolate(oracle.IM_BILINEAR, img, 0.0, 5.0)[3] == 1
    assert oracle.interpolate(oracle.IM_NEAREST, img, 28.9, 18.9)[3] == 0


def test_pyramid_level_properties(oracle):
    rng = np.random.default_rng(2)
    src = rng.integers(0, 256, (98, 131), dtype=np.uint8)
    dst = oracle.pyramid_level(src)
    assert dst.shape == (49, 65)
    assert not dst[0].any() and not dst[-1].any() and not dst[:, 0].any() and not dst[:, -1].any()
    km = np.array([0.05, 0.25, 0.4, 0.25, 0.05])
    k2 = np.outer(km, km)
    for tj, ti in [(1, 1), (10, 20), (47, 63), (25, 7)]:
        ref = (src[2 * tj - 2:2 * tj + 3, 2 * ti - 2:2 * ti + 3].astype(np.float64) * k2).sum()
        assert abs(int(dst[tj, ti]) - ref) <= 1.0
    flat = oracle.pyramid_level(np.full((64, 64), 200, np.uint8))
    assert set(np.unique(flat[1:-1, 1:-1])) <= {199, 200}


def test_decimate_and_translate(oracle):
    xy = oracle.rect_points(10, 20, 28, 38)  # 19 x 19
    assert len(xy) == 361 and tuple(xy[0]) == (10, 20) and tuple(xy[1]) == (10, 21)
    l1 = oracle.decimate(xy, 1)
    assert len(l1) == 100 and np.all(l1 * 2 % 2 == 0)
    l2 = oracle.decimate(l1, 1)
    assert len(l2) == 25
    assert np.array_equal(l2, oracle.decimate(xy, 2))  # step-1 chain == direct (Appendix B)


def test_solver_residual_and_pivoting(oracle):
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 6):
        for _ in range(50):
            J = rng.standard_normal((40, n)) * rng.uniform(0.1, 30, n)
            A = (J.T @ J).astype(np.float32)
            b = (J.T @ rng.standard_normal(40)).astype(np.float32)
            x = oracle.colpiv_qr_solve(A, b)
            ref = np.linalg.solve(A.astype(np.float64), b.astype(np.float64))
            assert np.allclose(x, ref, rtol=2e-3, atol=1e-5 * np.abs(ref).max())
    # rank deficient: zero column -> zero component, finite answer
    A = np.diag([4.0, 0.0, 9.0]).astype(np.float32)
    x = oracle.colpiv_qr_solve(A, np.array([8.0, 0.0, 18.0], np.float32))
    assert np.allclose(x, [2.0, 0.0, 2.0])


def test_damped_solve_semantics(oracle):
    A = np.array([[4.0, 1.0], [999.0, 3.0]], np.float32)  # lower triangle must be ignored
    b = np.array([1.0, 2.0], np.float32)
    dp = oracle.damped_solve(A, b, 0.5, 0.25)
    M = np.array([[4 * 1.5, 1.0], [1.0, 3 * 1.5]]) * 0.25
    assert np.allclose(dp, np.linalg.solve(M, b * 0.25), rtol=1e-5)


def test_newton_raphson_recovers_ground_truth(oracle, speckle512):
    und, dfm = speckle512
    o = oracle.Oracle(model=oracle.FM_UVUXUYVXVY)
    o.set_image(0, und)
    o.set_image(1, dfm)
    xy = oracle.rect_points(156, 156, 356, 356)
    r, tr = o.newton_raphson([0] * 6, xy, center=(256.0, 256.0), trace_cap=64)
    assert r["error_code"] == 0 and r["n_points"] == 40401 and r["iterations"] >= 1
    assert np.allclose(r["p"], [1.3, -0.7, 0.002, 0, 0, -0.001], atol=[0.02, 0.02, 2e-4, 2e-4, 2e-4, 2e-4])
    # trace: levels 2,1,0 in order, chi non-increasing over accepted steps of a level
    assert list(dict.fromkeys(tr["level"].tolist())) == [2, 1, 0]
    assert tr["kind"][0] == 0 and tr["lam"][0] == np.float32(1e-4)
    assert r["chi"] == tr["chi"][tr["level"] == 0].min()


def test_rigid_and_other_models_run(oracle, speckle512):
    und, dfm = speckle512
    xy = oracle.rect_points(200, 200, 240, 240)
    for model, P in ((oracle.FM_U, 1), (oracle.FM_UV, 2), (oracle.FM_UVQ, 3)):
        for interp in (oracle.IM_NEAREST, oracle.IM_BILINEAR, oracle.IM_BICUBIC):
            o = oracle.Oracle(model=model, interp=interp)
            o.set_image(0, und)
            o.set_image(1, dfm)
            r = o.newton_raphson([0] * P, xy, center=(220.0, 220.0))
            assert abs(r["p"][0] - 1.2) < 0.4, (model, interp, r)


def test_error_paths(oracle, speckle512):
    und, dfm = speckle512
    o = oracle.Oracle(model=oracle.FM_UV)
    o.set_image(0, und)
    o.set_image(1, dfm)
    # a sector hugging the border leaves the image at evaluation #0 of the coarsest level:
    # parameters come back unchanged, chi = FLT_MAX (correlation_class.cpp:389,413-419)
    xy = oracle.rect_points(0, 0, 20, 20)
    r = o.newton_raphson([0.5, 0.25], xy, center=(10.0, 10.0))
    assert r["error_code"] == 2 and r["chi"] == np.finfo(np.float32).max
    assert np.allclose(r["p"][:2], [0.5, 0.25])
    # max_iters = 0 -> error_correlation_max_iters_reached on every level, look-ahead returned
    o0 = oracle.Oracle(model=oracle.FM_UV, max_iters=0)
    o0.set_image(0, und)
    o0.set_image(1, dfm)
    r0 = o0.newton_raphson([0, 0], oracle.rect_points(200, 200, 240, 240), center=(220.0, 220.0))
    assert r0["error_code"] == 3


def test_cache_emulation_equals_on_the_fly_without_errors(oracle, speckle512):
    und, dfm = speckle512
    lists = [oracle.rect_points(100 + 40 * i, 120, 130 + 40 * i, 150) for i in range(6)]
    cen = [(115.0 + 40 * i, 135.0) for i in range(6)]
    res = []
    for mode in (0, 1):
        o = oracle.Oracle(model=oracle.FM_UVUXUYVXVY, cache_mode=mode)
        o.set_image(0, und)
        o.set_image(1, dfm)
        res.append(o.correlate_sectors(lists, centers=cen))
    assert res[0].tobytes() == res[1].tobytes()


def test_rect_geometry_matches_config2(oracle):
    xdim, ydim, cen = oracle.rect_sector_geometry(24.0, 24.0, 2023.0, 2023.0, 100, 100)
    assert (xdim, ydim) == (9, 9) and cen.shape == (10000, 2)
    assert cen[0, 0] - xdim >= 24 and cen[-1, 0] + xdim <= 2023
    assert cen[1, 0] == cen[0, 0] and cen[1, 1] > cen[0, 1]  # iSector = i*vs + j: j (y) runs fastest


def test_annular_points(oracle):
    pts = oracle.annular_points(20.0, 15.0, 0.0, 2 * np.pi, 100.0, 100.0, 1)
    r2 = (pts[:, 0] - 100) ** 2 + (pts[:, 1] - 100) ** 2
    assert len(pts) > 0 and np.all(r2 > 400) and np.all(r2 < 1225)
    assert abs(len(pts) - np.pi * (1225 - 400)) < 120
    # x outer, y inner
    assert np.all(np.diff(pts[:, 0]) >= 0)
    tot = sum(len(oracle.annular_points(20.0, 15.0, j * np.pi / 4, np.pi / 4, 100.0, 100.0, 8)) for j in range(8))
    assert abs(tot - len(pts)) < 0.05 * len(pts)


def test_initial_guess_policy(oracle):
    g, prev = oracle.adjust_initial_guess(3, 0, 1, [1, 2, 0.01, 0.02, 0.03, 0.04], 110.0, 95.0, 100.0, 100.0,
                                          [0] * 6, np.zeros(6, np.float32))
    assert np.allclose(g, [1 + 10 * 0.01 - 5 * 0.02, 2 + 10 * 0.03 - 5 * 0.04, 0.01, 0.02, 0.03, 0.04])
    assert np.array_equal(prev, g)
    g2, prev2 = oracle.adjust_initial_guess(3, 1, 1, [0] * 6, 0, 0, 0, 0, [2, 3, 0, 0, 0, 0], prev)
    assert np.allclose(g2[:2], [2 + (2 - g[0]), 3 + (3 - g[1])])
    assert np.allclose(prev2[:2], [2, 3])
    g3, _ = oracle.adjust_initial_guess(3, 1, 0, [0] * 6, 0, 0, 0, 0, [2, 3, 0, 0, 0, 0], prev)
    assert np.allclose(g3[:2], [2, 3])
