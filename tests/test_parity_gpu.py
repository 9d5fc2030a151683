"""GPU parity tests: the HIP engine, called through the C-ABI, against the CPU oracle on
the same seeded inputs, and against the reference-generated golden vectors.

What is bit-exact and what cannot be
  per-sample quantities (pyramid pixels, bicubic value/gradient, warp)  bit-exact (asserted)
  evaluation sums A, b, chi   rel 2e-5: float32 summation ORDER differs (the engine sums per
                              lane + a reduction tree, the reference sums sequentially)
  Newton-Raphson results      the summation order feeds b = sum(H*V), which cancels heavily
                              near convergence, so dp - and through the discontinuous
                              accept/reject and stop tests the whole trajectory - moves with
                              it.  The reference has the same sensitivity to its OWN
                              `number_of_threads` (default 20, defines.hpp:10), whose only
                              numerical effect is how the sums are split
                              (correlation_class.cpp:169-186,:253-275).  Measured on config C2
                              (tests/tools/parity_noise.py, 3000 sectors): oracle(T=8) vs
                              oracle(T=1) p50/p99/max = 3.6e-7 / 1.1e-4 / 1.3e-4 px on p0,p1
                              and 3e-7 / 2.6e-4 / 6e-4 relative on chi; the engine vs
                              oracle(T=1) shows the same distribution (3.6e-7 / 1.1e-4 /
                              1.3e-4 px; 3e-7 / 2.4e-4 / 6e-4).
  The 6x6 solve adds a second, larger, un-pinnable term: the reference calls Eigen's
  ColPivHouseholderQR in float32 (correlation_class.cpp:742-747; Eigen is not in the
  reference tree and its SIMD reduction order is build-dependent), whose own rounding error
  moves the results MORE than the thread split does: an oracle that solves the very same
  float32 systems exactly (float64 elimination) lands as far from the QR oracle (median
  4e-6 px, tests/tools/solver_noise.py) as any other backward-stable solver does.  The engine
  uses a root-free Cholesky (U^T D U) - the reference's own CUDA path uses cuSOLVER's
  Cholesky (cuda_solver.cu:120-149) - so it can be as close to the QR oracle as an exact
  solver is, not closer.
  So every Newton-Raphson comparison below checks the engine against oracle(T=1, QR) with
   (a) hard caps far below any real defect: 5e-3 px, 5e-5 on p2..p5, 5e-3 relative on chi,
       iteration counts within 1 (but for 1 sector in 200) - relaxed to 3x the yardsticks'
       own maxima where those are larger (starved configurations);
   (b) the strict tolerances of SURVEY.md 8c (1e-4 px on p0,p1; 1e-6 on p2..p5; chi rel
       1e-5; same iteration count) on at least as large a fraction of sectors as the
       yardsticks reach on the same sectors: oracle(T=8, QR) - the reference against itself
       - and oracle(T=1, exact float64 solve); the engine must match the lower of the two
       (minus a small-sample slack);
   (c) identical error codes, sample counts and centres.
"""
import os

import numpy as np
import pytest

import correlation_amd as ca

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
FLT_MAX = np.finfo(np.float32).max


def make_pair(speckle, model, interp=ca.IM_BICUBIC, oracle=None, **kw):
    """Engine + oracle(T=1) + yardstick oracles (T=8; exact solver) on the same pair."""
    und, dfm = speckle
    e = ca.HipCorrelationEngine(interpolation=interp, fitting_model=model, **kw)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    if oracle is None:
        return e, None
    os_ = []
    for T, solver in ((1, 0), (8, 0), (1, 2)):
        o = oracle.Oracle(interp=interp, model=model, n_threads=T, solver=solver, **kw)
        o.set_image(0, und)
        o.set_image(1, dfm)
        os_.append(o)
    return e, OraclePair(*os_)


class OraclePair:
    """oracle(T=1) is the parity target; oracle(T=8) measures the reference's own
    sensitivity to its thread count on the same inputs (the yardstick)."""

    def __init__(self, o1, o8, ox):
        self.o1, self.o8, self.ox = o1, o8, ox

    def correlate_sectors(self, lists, centers=None, guesses=None):
        return tuple(o.correlate_sectors(lists, centers=centers, guesses=guesses)
                     for o in (self.o1, self.o8, self.ox))

    def get_level(self, slot, level):
        return self.o1.get_level(slot, level)

    def set_image(self, slot, px):
        for o in (self.o1, self.o8, self.ox):
            o.set_image(slot, px)


def _strict_ok(a, b):
    dp = np.abs(a["p"] - b["p"])
    chi_rel = np.abs(a["chi"] - b["chi"]) / np.maximum(np.abs(b["chi"]), 1e-30)
    return ((dp[:, :2] <= 1e-4).all(1) & (dp[:, 2:] <= 1e-6).all(1) & (chi_rel <= 1e-5)
            & (a["iterations"] == b["iterations"]))


def compare_results(got, want_pair, label=""):
    """Asserts (a)-(c) of the module docstring; returns (strict fraction of the engine,
    strict fraction of the reference against itself)."""
    want, self8, exact = want_pair
    assert np.array_equal(got["error_code"], want["error_code"]), \
        f"{label}: error codes differ on {np.count_nonzero(got['error_code'] != want['error_code'])} sectors"
    assert np.array_equal(got["n_points"], want["n_points"])
    assert np.array_equal(got["und_cx"], want["und_cx"]) and np.array_equal(got["und_cy"], want["und_cy"])
    ok = want["error_code"] == 0
    bad = ~ok
    if bad.any():  # an out-of-image error at evaluation #0 returns the (rescaled) guess untouched
        e0 = bad & (want["chi"] == FLT_MAX)
        assert np.array_equal(got["chi"][e0], want["chi"][e0])
        assert np.allclose(got["p"][e0], want["p"][e0], atol=1e-6, equal_nan=True)
    g, w, s8, ex = got[ok], want[ok], self8[ok], exact[ok]
    if len(g) == 0:
        return 1.0, 1.0
    dp = np.abs(g["p"] - w["p"])
    chi_rel = np.abs(g["chi"] - w["chi"]) / np.maximum(np.abs(w["chi"]), 1e-30)
    # caps: far below any real defect, but never tighter than 3x what the yardstick runs
    # themselves show on these sectors (starved configurations are chaotic for everybody)
    def yard(field):
        out = 0.0
        for y in (s8, ex):
            if field == "chi":
                out = max(out, float((np.abs(y["chi"] - w["chi"]) / np.maximum(np.abs(w["chi"]), 1e-30)).max()))
            elif field == "p01":
                out = max(out, float(np.abs(y["p"] - w["p"])[:, :2].max()))
            elif field == "p25":
                out = max(out, float(np.abs(y["p"] - w["p"])[:, 2:].max()))
            else:
                out = max(out, float(np.count_nonzero(np.abs(y["iterations"] - w["iterations"]) > 1)))
        return out
    far = np.count_nonzero(np.abs(g["iterations"] - w["iterations"]) > 1)
    assert far <= max(1, len(g) // 200, 3 * yard("it")), f"{label}: iteration counts differ by > 1 on {far} sectors"
    assert dp[:, :2].max() <= max(5e-3, 3 * yard("p01")), f"{label}: p0/p1 off by {dp[:, :2].max()}"
    assert dp[:, 2:].max() <= max(5e-5, 3 * yard("p25")), f"{label}: p2..p5 off by {dp[:, 2:].max()}"
    assert chi_rel.max() <= max(5e-3, 3 * yard("chi")), f"{label}: chi rel {chi_rel.max()}"
    f_gpu, f_self, f_exact = _strict_ok(g, w).mean(), _strict_ok(s8, w).mean(), _strict_ok(ex, w).mean()
    slack = 0.03 + 1.5 / np.sqrt(len(g))
    print(f"{label}: strict tolerance met on {100 * f_gpu:.1f} % of {len(g)} sectors "
          f"(yardsticks: reference T=8 vs T=1 {100 * f_self:.1f} %, exact solver {100 * f_exact:.1f} %); "
          f"median |dp01| {np.median(dp[:, :2].max(1)):.2e}, median chi rel {np.median(chi_rel):.2e}")
    bar = min(f_self, f_exact)
    assert f_gpu >= bar - slack, f"{label}: strict fraction {f_gpu:.3f} < yardstick {bar:.3f} - {slack:.3f}"
    return f_gpu, bar


# ---------------------------------------------------------------------------------------------
def test_pyramid_bit_exact(oracle):
    rng = np.random.default_rng(5)
    for shape in ((64, 48), (98, 130), (131, 257), (512, 512), (768, 1024)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        if shape[0] == 512:
            img = ca.speckle.speckle_pair(512, 512, seed=11)[0]
        e = ca.HipCorrelationEngine(py_stop=3)
        e.set_undeformed_image(img)
        o = oracle.Oracle(py_stop=3)
        o.set_image(0, img)
        for lvl in range(0, 4):
            got, want = e.get_pyramid_level(ca.IMG_UND, lvl), o.get_level(0, lvl)
            assert got.shape == want.shape == (shape[0] >> lvl, shape[1] >> lvl)
            assert np.array_equal(got, want), (shape, lvl, np.count_nonzero(got != want))
        e.close()


def test_pyramid_from_device_frames_bit_exact(oracle):
    """lk_set_image_device: the fused upload + two-level kernel reads the caller's device buffer
    (any pitch, any alignment) and must give the same bytes at every level, also for 2- and
    3-level pyramids and sizes that are not multiples of the 64-pixel tile."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")  # the runtime the engine library itself is linked to
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    rng = np.random.default_rng(6)
    for shape, pad, shift, stop in (((64, 64), 0, 0, 2), ((98, 130), 6, 1, 2), ((131, 257), 3, 3, 3),
                                    ((260, 200), 56, 4, 2), ((77, 90), 0, 2, 1)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        step = shape[1] + pad
        host = np.full(shape[0] * step + 16, 255, np.uint8)  # pitch padding must never be read as pixels
        host[shift:shift + shape[0] * step].reshape(shape[0], step)[:, :shape[1]] = img
        e = ca.HipCorrelationEngine(py_stop=stop)
        dev = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(dev), host.size) == 0
        assert hip.hipMemcpy(dev, host.ctypes.data_as(ctypes.c_void_p), host.size, 1) == 0
        e.set_image_device(ca.IMG_DEF, dev.value + shift, shape[0], shape[1], step)
        e.synchronize()
        # the pair entry point (both frames in one launch) must give the same bytes in both slots
        e2 = ca.HipCorrelationEngine(py_stop=stop)
        img2 = np.ascontiguousarray(img[::-1])
        dev2 = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(dev2), img2.size) == 0
        assert hip.hipMemcpy(dev2, img2.ctypes.data_as(ctypes.c_void_p), img2.size, 1) == 0
        e2.set_image_pair_device(dev2.value, dev.value + shift, shape[0], shape[1], shape[1], step)
        e2.synchronize()
        o2 = oracle.Oracle(py_stop=stop)
        o2.set_image(0, img2)
        for lvl in range(0, stop + 1):
            assert np.array_equal(e2.get_pyramid_level(ca.IMG_DEF, lvl), e.get_pyramid_level(ca.IMG_DEF, lvl))
            assert np.array_equal(e2.get_pyramid_level(ca.IMG_UND, lvl), o2.get_level(0, lvl))
        e2.close()
        assert hip.hipFree(dev) == 0 and hip.hipFree(dev2) == 0
        o = oracle.Oracle(py_stop=stop)
        o.set_image(1, img)
        for lvl in range(0, stop + 1):
            got, want = e.get_pyramid_level(ca.IMG_DEF, lvl), o.get_level(1, lvl)
            assert np.array_equal(got, want), (shape, lvl, np.count_nonzero(got != want))
        e.close()


def test_annulus_in_one_call_and_device_built_levels(monkeypatch, speckle512):
    """lk_set_sectors_annular (all sectors of an annulus, rasterised on several host threads) and
    the device-side decimation of long explicit lists at commit must give exactly what the
    per-sector calls and the host decimation loop give: lists, centres, per-level counts, records."""
    und, dfm = speckle512
    rs, as_ = 3, 8
    dr, da = (200.0 - 60.0) / rs, 2 * np.pi / as_
    params = np.float32([[60.0 + i * dr, dr, j * da, da, 256.0, 250.0] for i in range(rs) for j in range(as_)])

    def build(batch, host_levels):
        monkeypatch.setenv("LK_HOST_REWARP", "1" if host_levels else "0")
        e = ca.HipCorrelationEngine(py_stop=3)
        e.set_batch_invariant(True)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        if batch:
            e.set_sectors_annular(0, params, as_)
        else:
            for s, q in enumerate(params):
                e.resetPolygon_annular(s, *[float(v) for v in q], as_)
        e.commit_sectors()
        S = e.n_sectors
        out = ([e.getUndXY0ToCPU(s).tobytes() for s in range(S)], [e.sector_info(s) for s in range(S)],
               [[e.sector_level_count(s, l) for l in range(4)] for s in range(S)], e.correlate_all().tobytes())
        total = sum(i[0] for i in out[1])
        e.close()
        return out, total

    want, total = build(False, True)
    assert total >= 32768, total        # long enough for the device path
    for batch, host_levels in ((True, True), (False, False), (True, False)):
        got, _ = build(batch, host_levels)
        for i, (w, g) in enumerate(zip(want, got)):
            assert w == g, (batch, host_levels, i)


def test_pairs_in_flight_give_the_records_of_a_lone_solve(speckle512):
    """Independent pairs solved side by side - one engine per pair, each on its own stream, all
    launches queued before any has finished (bench.py --inflight) - must give every pair the
    records it gets alone; the lk_set_pairs_in_flight hint only picks the lane group."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    und, dfm = speckle512
    pairs = [(und, dfm), (dfm, und), (und, np.ascontiguousarray(np.roll(dfm, 1, axis=1)))]

    def dev(a):
        d = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(d), a.size) == 0
        assert hip.hipMemcpy(d, a.ctypes.data_as(ctypes.c_void_p), a.size, 1) == 0
        return d

    frames = [(dev(a), dev(b)) for a, b in pairs]
    S = 40 * 40
    d_guess, d_res = dev(np.zeros(S * 24, np.uint8)), [dev(np.zeros(S * 48, np.uint8)) for _ in pairs]

    def engine(in_flight, stream=None):
        e = ca.HipCorrelationEngine()
        e.set_batch_invariant(True)        # the same bits whatever the neighbours on the GPU do
        e.set_pairs_in_flight(in_flight)
        if stream is not None:
            e.set_stream(stream.value)
        e.set_rect_grid(16.0, 16.0, 495.0, 495.0, 40, 40)
        e.commit_sectors()
        return e

    def fetch(d):
        out = np.zeros(S, ca.RESULT_DTYPE)
        assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), d, out.nbytes, 2) == 0
        return out

    alone = []
    e = engine(1)
    for (a, b), r in zip(frames, d_res):
        e.set_image_pair_device(a.value, b.value, 512, 512)
        e.correlate_all_device(d_guess.value, r.value)
        e.synchronize()
        alone.append(fetch(r))
    e.close()
    assert (alone[0]["error_code"] == 0).all() and not np.array_equal(alone[0]["p"], alone[1]["p"])
    streams, engines = [], []
    for _ in pairs:
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0     # hipStreamNonBlocking
        streams.append(st)
        engines.append(engine(len(pairs), st))
    for rep in range(3):                                                  # 9 launches queued back to back
        for e, (a, b), r in zip(engines, frames, d_res):
            e.set_image_pair_device(a.value, b.value, 512, 512)
            e.correlate_all_device(d_guess.value, r.value)
    for e in engines:
        e.synchronize()
    for k, r in enumerate(d_res):
        got = fetch(r)
        assert got.tobytes() == alone[k].tobytes(), f"pair {k}"
    for e, st in zip(engines, streams):
        e.close()
        assert hip.hipStreamDestroy(st) == 0
    for d in [x for f in frames for x in f] + [d_guess] + d_res:
        assert hip.hipFree(d) == 0
    with pytest.raises(ca.LkError):
        ca.HipCorrelationEngine().set_pairs_in_flight(0)


def test_team_launches_of_two_engines_take_turns(monkeypatch, speckle512):
    """Teams (several workgroups per sector that wait for each other) of different engines must
    not be on the GPU half-resident at the same time: every team launch waits for the previous
    one of the process.  Two engines queue team solves back to back on their own streams; each
    must get the records of a lone solve."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    monkeypatch.setenv("LK_FORCE_TEAM", "4")    # every sector a team of 4 workgroups (tiny: cannot fill the GPU)
    und, dfm = speckle512

    def engine(stream=None):
        e = ca.HipCorrelationEngine()
        if stream is not None:
            e.set_stream(stream.value)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        for s, (x0, y0) in enumerate(((40, 40), (260, 50), (60, 270), (250, 260))):
            e.resetPolygon_rect(s, x0, y0, x0 + 200, y0 + 190)
        e.commit_sectors()
        return e

    e = engine()
    want = e.correlate_all()
    assert (want["error_code"] == 0).all()
    e.close()
    streams, engines = [], []
    for _ in range(2):
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        streams.append(st)
        engines.append(engine(st))
    for rep in range(4):
        for e in engines:
            e.adjust_initial_guess(0, False, np.zeros(6, np.float32), (256.0, 256.0))
            e.correlate_all_async()
        for e in engines:
            got = e.wait_results()
            assert got.tobytes() == want.tobytes(), rep
    for e, st in zip(engines, streams):
        e.close()
        assert hip.hipStreamDestroy(st) == 0


@pytest.mark.parametrize("interp", [ca.IM_NEAREST, ca.IM_BILINEAR, ca.IM_BICUBIC])
def test_sampling_bit_exact(oracle, speckle512, interp):
    und, dfm = speckle512
    e = ca.HipCorrelationEngine(interpolation=interp)
    e.set_deformed_image(dfm)
    o = oracle.Oracle(interp=interp)
    o.set_image(1, dfm)
    rng = np.random.default_rng(17)
    for lvl in (0, 1, 2):
        img = o.get_level(1, lvl)
        h, w = img.shape
        pts = np.stack([rng.uniform(-2, w + 2, 1500), rng.uniform(-2, h + 2, 1500)], 1).astype(np.float32)
        pts[:50] = np.round(pts[:50])           # exact pixel centres
        pts[50:60] = [[1.0, 5.0]] * 10          # on the validity boundary
        pts[60:70] = [[w - 2.0, 5.0]] * 10
        got = e.sample(ca.IMG_DEF, lvl, pts)
        want = oracle.interpolate_many(interp, img, pts)
        assert np.array_equal(got[:, 3], want[:, 3]), "out-of-image flags differ"
        assert 0 < want[:, 3].sum() < len(pts)
        ok = want[:, 3] == 0
        assert np.array_equal(got[ok, :3].view(np.uint32), want[ok, :3].view(np.uint32)), \
            f"level {lvl}: {np.count_nonzero((got[ok, :3] != want[ok, :3]).any(1))} samples differ"
    e.close()


def test_warp_matches_reference_goldens():
    """getDefXY0 (the kModel_inPlace counterpart) against ModelClass_*::compute_model outputs
    produced by the reference's own objects (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(GOLD, "ref_model.npz"))
    img = np.zeros((64, 64), np.uint8)
    for model in (ca.FM_U, ca.FM_UV, ca.FM_UVQ, ca.FM_UVUXUYVXVY):
        e = ca.HipCorrelationEngine(fitting_model=model)
        e.set_undeformed_image(img)
        e.set_deformed_image(img)
        xy, p, c = g[f"m{model}_xy"], g[f"m{model}_p"], g[f"m{model}_c"]
        e.set_sector_points(0, xy, center=(float(c[0]), float(c[1])))
        e.commit_sectors()
        got = e.getDefXY0ToCPU(0, p)
        assert np.array_equal(got.view(np.uint32), g[f"m{model}_def"].view(np.uint32)), model
        e.close()


def test_blob_lists_match_reference_goldens():
    g = np.load(os.path.join(GOLD, "ref_blob.npz"))
    e = ca.HipCorrelationEngine()
    names = sorted({k.rsplit("_", 1)[0] for k in g.files if k.endswith("_contour")})
    for name in names:
        n = int(g[f"{name}_count"][0])
        e.clear_sectors()
        if n <= 0:
            with pytest.raises(ca.LkError) as ei:
                e.resetPolygon_blob(0, g[f"{name}_contour"])
            assert ei.value.code == ca.ERROR_BAD_DOMAIN
            continue
        e.resetPolygon_blob(0, g[f"{name}_contour"])
        assert np.array_equal(e.getUndXY0ToCPU(0), g[f"{name}_pts"].astype(np.float32)), name
    e.close()


@pytest.mark.parametrize("model", [ca.FM_U, ca.FM_UV, ca.FM_UVQ, ca.FM_UVUXUYVXVY])
@pytest.mark.parametrize("interp", [ca.IM_NEAREST, ca.IM_BILINEAR, ca.IM_BICUBIC])
def test_evaluation_sums(oracle, speckle512, model, interp):
    e, o = make_pair(speckle512, model, interp, oracle)
    xy = oracle.rect_points(240, 200, 270, 232)
    cx, cy = 255.0, 216.0
    e.set_sector_points(0, xy, center=(cx, cy))
    e.commit_sectors()
    P = ca.N_PARAMS[model]
    p = np.array([1.1, -0.6, 0.003, -0.002, 0.001, 0.002], np.float32)[:P]
    if model == ca.FM_UVQ:
        p[2] = 0.002
    for lvl in (0, 1, 2):
        lxy = xy if lvl == 0 else oracle.decimate(xy, lvl)
        assert e.sector_level_count(0, lvl) == len(lxy)
        pl = p.copy()
        pl[:min(P, 2)] /= (1 << lvl)
        A, b, chi, err = e.evaluate(0, lvl, pl)
        Ao, bo, chio, erro = oracle.evaluate(interp, model, o.get_level(0, lvl), o.get_level(1, lvl), lxy,
                                             np.float32(cx) * np.float32(1.0 / (1 << lvl)),
                                             np.float32(cy) * np.float32(1.0 / (1 << lvl)), pl)
        assert err == erro == 0
        iu = np.triu_indices(P)
        scale = np.abs(Ao[:P, :P]).max()
        assert np.allclose(A[:P, :P][iu], Ao[:P, :P][iu], rtol=2e-5, atol=2e-6 * scale)
        assert np.allclose(b[:P], bo[:P], rtol=2e-5, atol=2e-6 * np.abs(bo).max())
        assert abs(chi - chio) <= 2e-5 * chio
    e.close()


def test_damped_solve(oracle):
    """The engine factors the damped SPD matrix as U^T D U; the reference (and the oracle)
    use Eigen's pivoted Householder QR.  Both are backward stable: each must agree with the
    float64 solution to O(cond * eps), and with each other to the same bound."""
    e = ca.HipCorrelationEngine()
    rng = np.random.default_rng(23)
    eps = np.finfo(np.float32).eps
    worst = 0.0
    for n in (1, 2, 3, 6):
        for _ in range(40):
            J = rng.standard_normal((60, n)) * rng.uniform(0.5, 20, n)
            A = (J.T @ J).astype(np.float32)
            b = (J.T @ rng.standard_normal(60)).astype(np.float32)
            lam, s = np.float32(10.0 ** rng.uniform(-9, 1)), np.float32(1.0 / 361)
            got, want = e.damped_solve(A, b, lam, s), oracle.damped_solve(A, b, lam, s)
            M = np.triu(A.astype(np.float64)) + np.triu(A.astype(np.float64), 1).T
            M = M * float(s)
            M[np.diag_indices(n)] *= (1.0 + float(lam))
            ref = np.linalg.solve(M, b.astype(np.float64) * float(s))
            tol = 30 * eps * np.linalg.cond(M) * np.abs(ref).max() + 1e-30
            assert np.abs(got - ref).max() <= tol, (n, got, ref)
            assert np.abs(want - ref).max() <= tol, (n, want, ref)
            worst = max(worst, float(np.abs(got - want).max() / np.abs(ref).max()))
            # the reference path (used on starved / ill-conditioned systems) is the oracle's
            # pivoted QR operation by operation: same bits
            got_qr = e.damped_solve(A, b, lam, s, reference_solver=True)
            assert np.array_equal(got_qr, want), (n, got_qr, want)
    print(f"damped_solve: max |engine - oracle| / |dp| = {worst:.2e}")
    # semi-definite input: a textureless direction gets a zero step (like the QR's dropped pivot)
    A = np.diag([4.0, 0.0, 9.0]).astype(np.float32)
    got = e.damped_solve(A, np.array([8.0, 0.0, 18.0], np.float32), np.float32(0.0), np.float32(1.0))
    assert np.allclose(got, oracle.damped_solve(A, np.array([8.0, 0.0, 18.0], np.float32), 0.0, 1.0))
    assert np.allclose(got, [2.0, 0.0, 2.0])
    # fewer samples than parameters: a rank-4 6x6 system, damped - the engine detects the
    # small pivots and takes the reference path by itself
    J = rng.standard_normal((4, 6)) * [3, 3, 20, 20, 20, 20]
    A = (J.T @ J).astype(np.float32)
    b = (J.T @ rng.standard_normal(4)).astype(np.float32)
    for lam in (1e-4, 4e-5, 1e-9):
        assert np.array_equal(e.damped_solve(A, b, np.float32(lam), np.float32(0.25)),
                              oracle.damped_solve(A, b, np.float32(lam), np.float32(0.25)))
    e.close()


@pytest.mark.parametrize("model", [ca.FM_UV, ca.FM_UVUXUYVXVY])
def test_config1_single_big_sector(oracle, speckle512, model):
    """BASELINE config 1: 512^2 pair, one 201x201 sector (40 401 samples -> the 8-wave class)."""
    e, o = make_pair(speckle512, model, ca.IM_BICUBIC, oracle)
    e.resetPolygon_rect(0, 156, 156, 356, 356)
    e.commit_sectors()
    P = ca.N_PARAMS[model]
    got, guess = e.correlate(0, np.zeros(P, np.float32))
    xy = oracle.rect_points(156, 156, 356, 356)
    want, self8, exact = (np.array([x.newton_raphson(np.zeros(P), xy, center=(256.0, 256.0))])
                          for x in (o.o1, o.o8, o.ox))
    compare_results(np.array([got]), (want, self8, exact), f"config1 model {model}")
    # one sector: also state the numbers (40 401 samples: order noise alone is ~2e-5 on chi)
    assert got["iterations"] == want["iterations"][0]
    assert np.abs(got["p"] - want["p"][0])[:2].max() <= 1e-4 and np.abs(got["p"] - want["p"][0])[2:].max() <= 2e-6
    assert abs(got["chi"] - want["chi"][0]) <= 5e-5 * want["chi"][0]
    assert np.array_equal(guess[:P], got["p"][:P])
    assert abs(got["p"][0] - 1.3) < 0.02 and abs(got["p"][1] + 0.7) < 0.02
    e.close()


def grid_lists(oracle, x0, y0, x1, y1, hs, vs):
    xdim, ydim, cen = oracle.rect_sector_geometry(x0, y0, x1, y1, hs, vs)
    lists = [oracle.rect_points(cx - xdim, cy - ydim, cx + xdim, cy + ydim) for cx, cy in cen]
    return lists, cen.astype(np.float32)


def test_grid_10x10_affine_bicubic(oracle, speckle512):
    e, o = make_pair(speckle512, ca.FM_UVUXUYVXVY, ca.IM_BICUBIC, oracle)
    e.set_rect_grid(24.0, 24.0, 487.0, 487.0, 10, 10)
    e.commit_sectors()
    lists, cen = grid_lists(oracle, 24.0, 24.0, 487.0, 487.0, 10, 10)
    assert e.n_sectors == 100
    for s in (0, 37, 99):
        assert np.array_equal(e.getUndXY0ToCPU(s), lists[s])
    got = e.correlate_all(np.zeros(6, np.float32))
    want = o.correlate_sectors(lists, centers=cen)
    compare_results(got, want, "grid 10x10")
    assert (got["iterations"] == want[0]["iterations"]).mean() >= 0.95
    st = e.stats()
    assert st["sectors"] == 100 and st["evaluations"] > 300 and st["point_iterations"] >= 600
    assert st["algorithmic_bytes"] == 25 * st["sample_evaluations"] + 196 * st["evaluations"]
    per = e.sector_stats()            # the same counters per sector (lk_get_sector_stats)
    assert per.shape == (100, 4) and per[:, 0].sum() == st["evaluations"] and per[:, 1].sum() == st["sample_evaluations"]
    assert per[:, 2].sum() == st["point_iterations"] and (per[:, 0] >= 3).all()   # at least one evaluation per level
    assert (per[:, 1] >= per[:, 0]).all() and per[:, 3].sum() == st["ill_conditioned_solves"]
    # single-sector entry point == batch, bit for bit
    for s in (3, 58):
        one, _ = e.correlate(s, np.zeros(6, np.float32))
        assert one.tobytes() == got[s].tobytes()
    e.close()


@pytest.mark.parametrize("model", [ca.FM_U, ca.FM_UV, ca.FM_UVQ, ca.FM_UVUXUYVXVY])
@pytest.mark.parametrize("interp", [ca.IM_NEAREST, ca.IM_BILINEAR, ca.IM_BICUBIC])
def test_grid_every_model_and_interpolator(oracle, speckle512, model, interp):
    e, o = make_pair(speckle512, model, interp, oracle)
    e.set_rect_grid(100.0, 100.0, 400.0, 400.0, 6, 6)
    e.commit_sectors()
    lists, cen = grid_lists(oracle, 100.0, 100.0, 400.0, 400.0, 6, 6)
    got = e.correlate_all(np.zeros(6, np.float32))
    want = o.correlate_sectors(lists, centers=cen)
    compare_results(got, want, f"model {model} interp {interp}")
    e.close()


def test_error_paths(oracle, speckle512):
    e, o = make_pair(speckle512, ca.FM_UV, ca.IM_BICUBIC, oracle)
    xs = [oracle.rect_points(0, 0, 20, 20), oracle.rect_points(200, 200, 240, 240),
          oracle.rect_points(490, 470, 510, 500)]
    cs = [(10.0, 10.0), (220.0, 220.0), (500.0, 485.0)]
    for i, (xy, c) in enumerate(zip(xs, cs)):
        e.set_sector_points(i, xy, center=c)
    e.commit_sectors()
    g = np.tile(np.array([0.5, 0.25, 0, 0, 0, 0], np.float32), (3, 1))
    got = e.correlate_all(g)
    want = o.correlate_sectors(xs, centers=np.array(cs, np.float32), guesses=g)
    assert list(got["error_code"]) == list(want[0]["error_code"]) == [2, 0, 2]
    assert got["chi"][0] == FLT_MAX and np.array_equal(got["p"][0], want[0]["p"][0])
    assert np.allclose(got["p"][0][:2], [0.5, 0.25])
    compare_results(got, want, "error paths")
    e.close()
    # maximum_iterations = 0: every level stops with error_correlation_max_iters_reached
    e0, o0 = make_pair(speckle512, ca.FM_UV, ca.IM_BICUBIC, oracle, max_iters=0)
    e0.set_sector_points(0, xs[1], center=cs[1])
    e0.commit_sectors()
    got0 = e0.correlate_all(np.zeros(6, np.float32))
    want0 = o0.o1.newton_raphson([0, 0], xs[1], center=cs[1])
    assert got0["error_code"][0] == want0["error_code"] == 3
    assert np.allclose(got0["p"][0], want0["p"], atol=1e-4)
    e0.close()
    # per-sector counters before any sector is committed: an error, not garbage
    e1 = ca.HipCorrelationEngine(fitting_model=ca.FM_UV)
    with pytest.raises(ca.LkError) as err:
        e1.sector_stats()
    assert err.value.code == ca.ERROR_BAD_DOMAIN and "no committed sectors" in str(err.value)
    e1.close()


def test_annular_and_blob_sectors(oracle):
    und, dfm = ca.speckle.speckle_pair(768, 768, p=(0.9, 0.4, 0.001, 0.0005, -0.0005, 0.0015), seed=21)
    e, o = make_pair((und, dfm), ca.FM_UVUXUYVXVY, ca.IM_BICUBIC, oracle)
    lists = []
    rs, as_ = 2, 4
    ri, ro, cx, cy = 120.0, 330.0, 384.0, 384.0
    dr, da = (ro - ri) / rs, np.float32(2 * np.pi) / np.float32(as_)
    s = 0
    for i in range(rs):
        for j in range(as_):
            r, a = np.float32(ri + i * dr), np.float32(j) * da
            e.resetPolygon_annular(s, r, dr, a, da, cx, cy, as_)
            lists.append(oracle.annular_points(r, dr, a, da, cx, cy, as_))
            s += 1
    e.resetPolygon_annular(s, 40.0, 60.0, 0.0, 2 * np.pi, 384.0, 384.0, 1)  # full ring, as == 1
    lists.append(oracle.annular_points(40.0, 60.0, 0.0, np.float32(2 * np.pi), 384.0, 384.0, 1))
    s += 1
    t = 2 * np.pi * np.arange(24) / 24
    rad = np.where(np.arange(24) % 2 == 0, 300.0, 190.0)
    contour = np.stack([384 + rad * np.cos(t), 384 + rad * np.sin(t)], 1).astype(np.float32)
    e.resetPolygon_blob(s, contour)
    lists.append(oracle.blob_points(contour))
    e.commit_sectors()
    for k, want_xy in enumerate(lists):
        assert np.array_equal(e.getUndXY0ToCPU(k), want_xy), f"sector {k} sample list"
    sizes = np.array([len(x) for x in lists])
    assert sizes.min() > 2048 and sizes.max() > 32768  # exercises the 4- and 8-wave classes
    got = e.correlate_all(np.zeros(6, np.float32))
    want = o.correlate_sectors(lists)  # centre = float mean of the samples
    compare_results(got, want, "annular+blob")
    # ground truth displacement at each sector centre (the images were deformed about (384,384))
    u_true = 0.9 + 0.001 * (got["und_cx"] - 384.0) + 0.0005 * (got["und_cy"] - 384.0)
    assert np.abs(got["p"][:, 0] - u_true).max() < 0.05
    e.close()


def test_starved_pyramid_level_is_bit_identical(oracle):
    """BASELINE config 5's geometry in small: 17x17 sectors with a 4-level pyramid leave
    2x2 .. 3x3 samples at level 3 for 6 parameters.  The damped system is singular, the
    reference's pivoted QR decides the step and amplifies the last bit of A and b, so the
    engine sums such levels in the reference's order and uses the reference's solver:
    solving level 3 alone must reproduce the oracle BIT FOR BIT, and the full 4-level solve
    must be as close to it as the reference is to itself."""
    und, dfm = ca.speckle.speckle_pair(768, 768, p=(1.3, -0.7, 0.0005, 0.0, 0.0, -0.00025), seed=13)
    xdim, ydim, cen = oracle.rect_sector_geometry(32.0, 32.0, 735.0, 735.0, 39, 39)
    lists = [oracle.rect_points(cx - xdim, cy - ydim, cx + xdim, cy + ydim) for cx, cy in cen]
    cen = cen.astype(np.float32)
    # level 3 only
    e = ca.HipCorrelationEngine(py_start=3, py_stop=3)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    e.set_rect_grid(32.0, 32.0, 735.0, 735.0, 39, 39)
    e.commit_sectors()
    assert e.sector_info(0)[0] == 289 and max(e.sector_level_count(s, 3) for s in range(0, 1521, 97)) <= 9
    got = e.correlate_all(np.zeros(6, np.float32))
    o = oracle.Oracle(py_start=3, py_stop=3)
    o.set_image(0, und)
    o.set_image(1, dfm)
    want = o.correlate_sectors(lists, centers=cen)
    assert got.tobytes() == want.tobytes()
    e.close()
    # all four levels
    e, op = make_pair((und, dfm), ca.FM_UVUXUYVXVY, ca.IM_BICUBIC, oracle, py_stop=3)
    e.set_rect_grid(32.0, 32.0, 735.0, 735.0, 39, 39)
    e.commit_sectors()
    got = e.correlate_all(np.zeros(6, np.float32))
    compare_results(got, op.correlate_sectors(lists, centers=cen), "4-level, starved level 3")
    e.close()


def test_shards_equal_full_run(speckle512):
    e, _ = make_pair(speckle512, ca.FM_UVUXUYVXVY)
    e.set_rect_grid(24.0, 24.0, 487.0, 487.0, 12, 9)
    e.commit_sectors()
    full = e.correlate_all(np.zeros(6, np.float32))
    parts = []
    for first, count in ((0, 40), (40, 41), (81, 27)):
        e.set_rect_grid(24.0, 24.0, 487.0, 487.0, 12, 9, first, count)
        e.commit_sectors()
        parts.append(e.correlate_all(np.zeros(6, np.float32)))
    assert np.concatenate(parts).tobytes() == full.tobytes()
    # and the run is deterministic
    e.set_rect_grid(24.0, 24.0, 487.0, 487.0, 12, 9)
    e.commit_sectors()
    assert e.correlate_all(np.zeros(6, np.float32)).tobytes() == full.tobytes()
    e.close()


def test_sequence_constant_velocity(oracle):
    frames = ca.speckle.speckle_sequence(384, 384, 4, velocity=(0.8, -0.4), dilation=2e-4, seed=5)
    model = ca.FM_UVUXUYVXVY
    e, o = make_pair((frames[0], frames[1]), model, ca.IM_BICUBIC, oracle)
    e.set_rect_grid(40.0, 40.0, 343.0, 343.0, 5, 5)
    e.commit_sectors()
    lists, cen = grid_lists(oracle, 40.0, 40.0, 343.0, 343.0, 5, 5)
    gcx, gcy = 191.5, 191.5
    prev = np.zeros((25, 6), np.float32)
    res_o = np.zeros((25, 6), np.float32)
    e.set_deformed_image(frames[1])
    e.set_next_image(frames[2])
    for f in range(3):  # pairs (0,1), (0,2), (0,3): Eulerian, reference = first image
        o.set_image(1, frames[f + 1])
        e.adjust_initial_guess(f, True, np.zeros(6, np.float32), (gcx, gcy))
        g_o = np.zeros((25, 6), np.float32)
        for s in range(25):
            g_o[s], prev[s] = oracle.adjust_initial_guess(model, f, True, np.zeros(6), cen[s, 0], cen[s, 1], gcx,
                                                          gcy, res_o[s], prev[s])
        want = o.correlate_sectors(lists, centers=cen, guesses=g_o)
        g_e = e.get_guesses()
        got = e.correlate_all(None)
        if f == 0:
            assert np.array_equal(g_e, g_o)
        else:
            assert np.allclose(g_e, g_o, atol=2e-4)
            assert np.abs(g_e[:, 0]).min() > 0.5  # the extrapolated guess is in use
        compare_results(got, want, f"frame {f}")
        res_o = want[0]["p"].copy()
        if f < 2:
            e.makeDefPyramidFromNxt()
            if f == 0:
                e.set_next_image(frames[3])
    # frame 3: translation 3*(0.8,-0.4) plus dilation 3*2e-4 about the image centre (192,192)
    assert np.allclose(got["p"][:, 0], 2.4 + 6e-4 * (got["und_cx"] - 192.0), atol=0.05)
    assert np.allclose(got["p"][:, 1], -1.2 + 6e-4 * (got["und_cy"] - 192.0), atol=0.05)
    e.close()


def test_config2_full_size_properties(oracle):
    """BASELINE config 2 at full size: 2048^2, 100x100 sectors of 19x19, affine, 3 levels.
    Size-independent properties + oracle parity on a seeded subset of sectors."""
    truth = (1.3, -0.7, 0.002, 0.0, 0.0, -0.001)
    und, dfm = ca.speckle.speckle_pair(2048, 2048, p=truth, seed=7)
    e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    e.set_rect_grid(24.0, 24.0, 2023.0, 2023.0, 100, 100)
    e.commit_sectors()
    assert e.n_sectors == 10000 and e.sector_info(0)[0] == 361
    r1 = e.correlate_all(np.zeros(6, np.float32))
    r2 = e.correlate_all(np.zeros(6, np.float32))
    assert r1.tobytes() == r2.tobytes(), "the batch solve must be deterministic"
    assert (r1["error_code"] == 0).mean() > 0.999
    # ground truth: u(x) = 1.3 + 0.002 (x - 1024), v(y) = -0.7 - 0.001 (y - 1024) at each centre
    u_true = truth[0] + truth[2] * (r1["und_cx"] - 1024.0)
    v_true = truth[1] + truth[5] * (r1["und_cy"] - 1024.0)
    assert np.median(np.abs(r1["p"][:, 0] - u_true)) < 0.02
    assert np.median(np.abs(r1["p"][:, 1] - v_true)) < 0.02
    e2, o = make_pair((und, dfm), ca.FM_UVUXUYVXVY, ca.IM_BICUBIC, oracle)
    e2.close()
    lists, cen = grid_lists(oracle, 24.0, 24.0, 2023.0, 2023.0, 100, 100)
    pick = np.random.default_rng(1).choice(10000, 1000, replace=False)
    want = o.correlate_sectors([lists[i] for i in pick], centers=cen[pick])
    compare_results(r1[pick], want, "config 2 subset")
    assert (r1[pick]["iterations"] == want[0]["iterations"]).mean() >= 0.95
    e.close()


def test_mixed_size_classes_in_one_engine(oracle, speckle512):
    """Small rectangles, a mid-size explicit list and a large rectangle in ONE engine: every
    size class (16 / 32 / 64 lanes, 4- and 8-wavefront workgroups) gets its own launch through
    the `order` indirection, and records land at their sector index."""
    e, o = make_pair(speckle512, ca.FM_UVUXUYVXVY, ca.IM_BICUBIC, oracle)
    lists, cens = [], []

    def rect(s, x0, y0, x1, y1):
        e.resetPolygon_rect(s, x0, y0, x1, y1)
        lists.append(oracle.rect_points(x0, y0, x1, y1))
        cens.append(((x0 + x1) * 0.5, (y0 + y1) * 0.5))

    rect(0, 100, 100, 118, 118)            # 361 samples
    rect(1, 150, 100, 220, 170)            # 5041
    rect(2, 130, 260, 148, 278)            # 361
    rect(3, 200, 200, 420, 420)            # 48841
    xy = oracle.rect_points(60, 300, 100, 330)[::2].copy()   # every other sample: explicit list, 636
    e.set_sector_points(4, xy, center=(80.0, 315.0))
    lists.append(xy)
    cens.append((80.0, 315.0))
    rect(5, 300, 60, 330, 95)              # 1116
    e.commit_sectors()
    got = e.correlate_all(np.zeros(6, np.float32))
    want = o.correlate_sectors(lists, centers=np.array(cens, np.float32))
    assert list(got["n_points"]) == [361, 5041, 361, 48841, 636, 1116]
    compare_results(got, want, "mixed classes")
    for s in (1, 3, 4):
        one, _ = e.correlate(s, np.zeros(6, np.float32))
        assert one.tobytes() == got[s].tobytes()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("width", [2, 5])
def test_team_of_workgroups_per_sector(oracle, speckle512, width, monkeypatch):
    """Giant sectors are shared by a TEAM of 512-thread workgroups that exchange their partial
    sums through global memory once per evaluation (kernel K1, "teams").  Forced here
    on sectors small enough for the oracle (LK_FORCE_TEAM is read at commit): the records must
    pass the same parity bars as any other lane group, and agree with the single-workgroup
    engine far inside them (only the summation order differs)."""
    rects = [(40, 40, 250, 250), (260, 40, 470, 260), (40, 270, 250, 470), (300, 300, 318, 318)]
    lists = [oracle.rect_points(*r) for r in rects]
    cens = np.array([((r[0] + r[2]) * 0.5, (r[1] + r[3]) * 0.5) for r in rects], np.float32)

    def run():
        e, o = make_pair(speckle512, ca.FM_UVUXUYVXVY, ca.IM_BICUBIC, oracle)
        for s, r in enumerate(rects):
            e.resetPolygon_rect(s, *r)
        e.commit_sectors()
        got = e.correlate_all(np.zeros(6, np.float32))
        one, _ = e.correlate(1, np.zeros(6, np.float32))
        assert one.tobytes() == got[1].tobytes()
        e.close()
        return got, o

    plain, o = run()
    monkeypatch.setenv("LK_FORCE_TEAM", str(width))
    team, _ = run()
    want = o.correlate_sectors(lists, centers=cens)
    compare_results(team, want, f"team of {width}")
    assert np.array_equal(team["error_code"], plain["error_code"])
    assert np.array_equal(team["iterations"], plain["iterations"])
    assert np.abs(team["p"] - plain["p"]).max() < 2e-5
    assert (np.abs(team["chi"] - plain["chi"]) <= 1e-5 * np.abs(plain["chi"])).all()


@pytest.mark.gpu
def test_broken_team_falls_back_to_one_workgroup(speckle512, monkeypatch):
    """A team workgroup that never shows up (LK_TEAM_FAULT: rank 1 skips its arrival at step 3 - in
    production: a workgroup that is not resident because foreign kernels hold the GPU) must not
    cost the sector: the waiting workgroups give up after ~1 s, mark the team broken, and rank 0
    solves the sector again on its own.  The record is the single-workgroup solve's, bit for bit."""
    rects = [(40, 40, 250, 250), (260, 40, 470, 260)]

    def run():
        e = ca.HipCorrelationEngine()
        e.set_undeformed_image(speckle512[0])
        e.set_deformed_image(speckle512[1])
        for s, r in enumerate(rects):
            e.resetPolygon_rect(s, *r)
        e.commit_sectors()
        got = e.correlate_all(np.zeros(6, np.float32))
        e.close()
        return got

    monkeypatch.setenv("LK_FORCE_TEAM", "1")        # one 8-wavefront workgroup per sector
    alone = run()
    monkeypatch.setenv("LK_FORCE_TEAM", "4")
    healthy = run()
    monkeypatch.setenv("LK_TEAM_FAULT", "3")
    broken = run()
    assert (alone["error_code"] == 0).all() and (broken["error_code"] == 0).all()
    assert broken.tobytes() == alone.tobytes()
    assert healthy.tobytes() != alone.tobytes() and np.abs(healthy["p"] - alone["p"]).max() < 2e-5


@pytest.mark.gpu
def test_team_solve_with_foreign_kernels_in_flight(speckle512, monkeypatch):
    """Team launches assume their workgroups become resident; kernels of other streams (here: another
    engine's persistent non-team solves, queued back to back on its own stream - the shape RCCL
    kernels or a second engine have) may hold CU slots meanwhile.  They drain without waiting on
    anybody, so the team only starts later: same records as the lone team solve."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    und, dfm = speckle512
    st = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
    noisy = ca.HipCorrelationEngine()               # 57 600 small sectors: a persistent grid that fills every CU
    noisy.set_stream(st.value)
    noisy.set_undeformed_image(und)
    noisy.set_deformed_image(dfm)
    noisy.set_rect_grid(16.0, 16.0, 495.0, 495.0, 240, 240)
    noisy.commit_sectors()
    monkeypatch.setenv("LK_FORCE_TEAM", "6")
    e = ca.HipCorrelationEngine()
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    for s, (x0, y0) in enumerate(((40, 40), (260, 50), (60, 270), (250, 260))):
        e.resetPolygon_rect(s, x0, y0, x0 + 200, y0 + 190)
    e.commit_sectors()
    want = e.correlate_all(np.zeros(6, np.float32))
    assert (want["error_code"] == 0).all()
    for rep in range(3):
        for _ in range(4):
            noisy.adjust_initial_guess(0, False, np.zeros(6, np.float32), (256.0, 256.0))
            noisy.correlate_all_async()
            got = e.correlate_all(np.zeros(6, np.float32))   # runs while the other stream's solve is on the GPU
            assert got.tobytes() == want.tobytes(), rep
            noisy.wait_results()
    noisy.close()
    e.close()
    assert hip.hipStreamDestroy(st) == 0


@pytest.mark.gpu
def test_solo_half_wavefronts(oracle, speckle512, monkeypatch):
    """32-lane groups: a half-wavefront that has run out of work joins its partner's sector
    ("solo").  Same parity bars as everything else; against the
    batch-invariant mode only the summation grouping differs; and in batch-invariant mode a
    sector gets the same bits alone, in a shard with other neighbours, and in the full batch."""
    monkeypatch.setenv("LK_FORCE_GROUP", "32")
    xdim, ydim, cen = oracle.rect_sector_geometry(24.0, 24.0, 487.0, 487.0, 11, 11)
    lists = [oracle.rect_points(cx - xdim, cy - ydim, cx + xdim, cy + ydim) for cx, cy in cen]

    def run(invariant, first=0, count=-1):
        e, o = make_pair(speckle512, ca.FM_UVUXUYVXVY, ca.IM_BICUBIC, oracle)
        e.set_batch_invariant(invariant)
        e.set_rect_grid(24.0, 24.0, 487.0, 487.0, 11, 11, first, count)
        e.commit_sectors()
        got = e.correlate_all(np.zeros(6, np.float32))
        one, _ = e.correlate(3, np.zeros(6, np.float32))
        e.close()
        return got, one, o

    solo, _, o = run(False)
    inv, inv_one, _ = run(True)
    want = o.correlate_sectors(lists, centers=np.array(cen, np.float32))
    compare_results(solo, want, "solo half-wavefronts")
    compare_results(inv, want, "batch-invariant")
    assert np.array_equal(solo["error_code"], inv["error_code"])
    assert np.abs(solo["p"] - inv["p"])[:, :2].max() < 5e-3
    assert inv_one.tobytes() == inv[3].tobytes()
    shard, _, _ = run(True, 37, 51)      # odd start: every sector has a different partner
    assert shard.tobytes() == inv[37:88].tobytes()


# ---------------------------------------------------------------------------------------------
# sequences: lk_sequence_run (engine + tracker, include/lk_tracker.h) against the manager oracle
# ---------------------------------------------------------------------------------------------
def _report_table(text):
    rows = [r.split(",") for r in text.strip().split("\n")]
    head, body = rows[0], rows[1:]
    return head, body


@pytest.mark.gpu
@pytest.mark.parametrize("domain,deformation,reference,model", [
    (0, 2, 0, ca.FM_UVUXUYVXVY),    # rectangular grid, Eulerian, first image as reference (config 4's mode)
    (0, 1, 1, ca.FM_UVUXUYVXVY),    # rectangular, Lagrangian, previous image as reference
    (0, 0, 1, ca.FM_UV),            # rectangular, strict Lagrangian
    (1, 1, 1, ca.FM_UVQ),           # annular, Lagrangian
    (1, 0, 1, ca.FM_UVUXUYVXVY),    # annular, strict Lagrangian
    (2, 0, 1, ca.FM_UVQ),           # blob, strict Lagrangian
    (2, 2, 0, ca.FM_UVUXUYVXVY),    # blob, Eulerian
    (2, 1, 1, ca.FM_UVUXUYVXVY),    # blob, Lagrangian
    (3, 1, 1, ca.FM_UVUXUYVXVY),    # rectangular, Lagrangian, a finer grid (11 x 9 sectors of 15 x 17 samples)
    (3, 2, 0, ca.FM_UVQ),           # the same grid, Eulerian, rigid + rotation
])
def test_sequence_tracking_against_manager_oracle(oracle, domain, deformation, reference, model):
    """perform_multiframe_correlation on the HIP engine: frame roles (und <- def <- nxt with the
    next frame uploaded behind the solve), sector tracking, guesses, frame_results and the CSV
    report, compared with the CPU restatement of managerClass driving the CPU oracle."""
    from correlation_amd import tracker as tk
    from oracle import lk_manager_oracle as mo
    fine = domain == 3
    domain = 0 if fine else domain
    frames = ca.speckle.speckle_sequence(256, 256, 5, velocity=(0.9, -0.5), dilation=4e-4, seed=3)
    names = [f"f{i}.pgm" for i in range(len(frames))]
    guess = [0.5, -0.25, 0.0, 0.0, 0.0, 0.0]
    e = ca.HipCorrelationEngine(fitting_model=model)
    t = tk.SequenceTracker(model, domain, deformation, reference, tk.ERRMODE_CONTINUE, guess, lib=e.lib)
    o = oracle.Oracle(model=model)
    o.set_image(0, frames[0])
    o.set_image(1, frames[1])
    m = mo.ManagerOracle(o, model, domain, deformation, reference, mo.ERRMODE_CONTINUE, guess)
    if domain == 0:
        args = (40.0, 44.0, 215.0, 211.0, 127.5, 127.5, 11 if fine else 4, 9 if fine else 3)
        t.set_rect_domain(*args), m.set_rect_domain(*args)
    elif domain == 1:
        args = (30.0, 78.0, 128.0, 126.0, 2, 3)
        t.set_annular_domain(*args), m.set_annular_domain(*args)
    else:
        ang = 2 * np.pi * np.arange(7) / 7
        contour = np.stack([128 + 45 * np.cos(ang), 126 + 38 * np.sin(ang)], 1).astype(np.float32)
        t.set_blob_domain(contour, 128.0, 126.0), m.set_blob_domain(contour, 128.0, 126.0)
    assert tk.run_sequence(e, t, frames, names) == len(frames) - 1
    for k in range(len(frames) - 1):
        if k > 0:
            if reference == 1:
                o.und_from_def()
            o.set_image(2, frames[k + 1])
            o.def_from_nxt()
        m.run_frame(k, names[0] if reference == 0 else names[k], names[k + 1])
    head_g, got = _report_table(t.report())
    head_w, want = _report_table(m.report_text())
    assert head_g == head_w and len(got) == len(want) == (len(frames) - 1) * t.n_sectors
    col = {name: i for i, name in enumerate(head_g)}
    P = ca.N_PARAMS[model]
    strict_lagrangian = deformation == 0
    same_it = 0
    for rg, rw in zip(got, want):
        assert rg[:3] == rw[:3]                                   # frame number and file names
        for name in ("number_of_points", "error_status", "error_code"):
            assert rg[col[name]] == rw[col[name]], (rg[0], name)
        same_it += abs(int(rg[col["iterations"]]) - int(rw[col["iterations"]])) <= 1
        tol_c = 2e-2 if strict_lagrangian else 5e-3                 # float sample lists drift a little
        for name in ("und_center_x", "und_center_y", "def_center_x", "def_center_y", "und_global_center_x",
                     "def_global_center_x", "def_global_center_y", "parameter_0", "Initial_guess_0"):
            assert abs(float(rg[col[name]]) - float(rw[col[name]])) <= tol_c, (rg[0], name, rg[col[name]], rw[col[name]])
        for p in range(2, P):
            assert abs(float(rg[col[f"parameter_{p}"]]) - float(rw[col[f"parameter_{p}"]])) <= 2e-4
        assert abs(float(rg[col["def_angle(rad)"]]) - float(rw[col["def_angle(rad)"]])) <= 2e-4
        cg, cw = float(rg[col["chi"]]), float(rw[col["chi"]])
        assert abs(cg - cw) <= 2e-2 * abs(cw) + 1e-6
    assert same_it >= 0.9 * len(got)
    # the tracked displacement is the ground truth of the sequence: 0.9 / -0.5 px per frame
    last = got[-t.n_sectors:]
    if reference == 0 and deformation == 2:
        u = np.array([float(r[col["parameter_0"]]) for r in last])
        assert np.abs(u - 0.9 * (len(frames) - 1)).max() < 0.25
    e.close(), t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("given_centers,model,py_stop,move", [
    (True, ca.FM_UVUXUYVXVY, 2, "rewarp"), (False, ca.FM_UVQ, 2, "rewarp"), (False, ca.FM_UV, 3, "rewarp"),
    (True, ca.FM_U, 1, "rewarp"), (False, ca.FM_UVUXUYVXVY, 2, "translate"), (True, ca.FM_UVQ, 3, "translate")])
def test_device_rebuilt_lists_equal_the_host_rebuild(monkeypatch, given_centers, model, py_stop, move):
    """lk_rewarp_sectors rebuilds the moved sample lists on the device (warp, per-level
    decimation as one compaction, mean centres); the host rebuild (LK_HOST_REWARP=1) is the
    restated reference loop.  Lists, centres, per-level counts and the next solve's records must
    be the same bits - over two moves and a partial restore."""
    frames = ca.speckle.speckle_sequence(320, 288, 3, velocity=(0.7, -0.4), dilation=6e-4, seed=9)
    monkeypatch.setenv("LK_EVAL_LISTS", "0")   # (records bit for bit: the host rebuild has the lists in the reference's order only)

    def oracle_free_rect(x0, y0, x1, y1):   # the samples of a rectangle as an explicit list, x outer / y inner
        xs, ys = np.meshgrid(np.arange(x0, x1 + 1), np.arange(y0, y1 + 1), indexing="ij")
        return np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float32)

    def run(host):
        monkeypatch.setenv("LK_HOST_REWARP", "1" if host else "0")
        rng = np.random.default_rng(4)
        e = ca.HipCorrelationEngine(fitting_model=model, py_stop=py_stop)
        e.set_batch_invariant(True)
        e.set_undeformed_image(frames[0])
        e.set_deformed_image(frames[1])
        if move == "translate":     # explicit lists only: implicit rectangles stay with the host records
            e.set_sector_points(0, oracle_free_rect(40, 50, 70, 75), center=(55.0, 62.5))
        else:
            e.resetPolygon_rect(0, 40, 50, 70, 75)
        e.resetPolygon_annular(1, 40.0, 30.0, 0.3, 1.1, 150.0, 140.0, 1)
        ang = 2 * np.pi * np.arange(9) / 9
        e.resetPolygon_blob(2, np.stack([200 + 50 * np.cos(ang), 90 + 40 * np.sin(ang)], 1).astype(np.float32))
        pts = np.stack([rng.uniform(60, 250, 700), rng.uniform(60, 220, 700)], 1).astype(np.float32)
        if move == "translate":
            e.set_sector_points(3, oracle_free_rect(100, 200, 104, 203), center=(102.0, 201.5))
            e.set_sector_points(4, np.round(pts), center=(155.25, 140.5))
            e.set_sector_points(5, oracle_free_rect(180, 160, 260, 250), center=(220.0, 205.0))
            e.set_sector_points(6, oracle_free_rect(2, 2, 30, 34), center=(16.0, 18.0))   # fails: leaves the image
        else:
            e.resetPolygon_rect(3, 100, 200, 104, 203)          # 5 x 4 samples: starved upper levels
            e.set_sector_points(4, pts, center=(155.25, 140.5))
            e.resetPolygon_rect(5, 180, 160, 260, 250)
            e.resetPolygon_rect(6, 2, 2, 30, 34)                # fails: leaves the image
        e.commit_sectors()
        S = e.n_sectors
        out = []
        g = np.zeros((S, 6), np.float32)
        g[:, 0], g[:, 1] = 0.5, -0.25
        for k in range(2):
            r = e.correlate_all(g)
            # (no host-side query here: the second move must find the first one's lists on the device)
            centers = np.stack([np.trunc(r["und_cx"] + r["p"][:, 0] + 0.5), np.trunc(r["und_cy"] + 0.5)],
                               1).astype(np.float32) if given_centers else None
            if move == "translate":
                e.translate_sectors(np.ascontiguousarray(r["p"][:, :2] + np.float32([0.3, 0.0])), centers)
            else:
                e.rewarp_sectors(centers)
            counts = [[e.sector_level_count(s, l) for l in range(py_stop + 1)] for s in range(S)]
            if k == 0:      # the lists stay on the device between the frames of a sequence
                e.makeUndPyramidFromDef()
                e.set_deformed_image(frames[2])
                out.append((r.tobytes(), counts))
                continue
            lists = [e.getUndXY0ToCPU(s).tobytes() for s in range(S)]
            out.append((r.tobytes(), counts, lists, [e.sector_info(s) for s in range(S)]))
        r = e.correlate_all(g)
        out.append(r.tobytes())
        e.restore_sectors(2)        # sectors 2.. go back to the lists of the second frame
        out.append([e.getUndXY0ToCPU(s).tobytes() for s in range(S)])
        out.append([[e.sector_level_count(s, l) for l in range(py_stop + 1)] for s in range(S)])
        out.append(e.correlate_all(g).tobytes())
        e.close()
        return out

    want, got = run(True), run(False)
    assert len(want) == len(got)
    for i, (w, g) in enumerate(zip(want, got)):
        assert w == g, f"stage {i}"


@pytest.mark.gpu
@pytest.mark.parametrize("deformation", [1, 0])
def test_stop_frame_policy_through_the_frame_loop(oracle, deformation):
    """stopFrame with sectors that walk out of the image: the sector loop of a frame ends at the
    first failing sector, the sectors behind it keep the state AND the sample lists they had
    (lk_restore_sectors after lk_translate_sectors / lk_rewarp_sectors), and the sequence goes on
    (manager_class.cpp:520-546).  Compared with the manager oracle, which applies the same policy."""
    from correlation_amd import tracker as tk
    from oracle import lk_manager_oracle as mo
    model = ca.FM_UV
    frames = ca.speckle.speckle_sequence(256, 256, 6, velocity=(3.6, -2.2), dilation=0.0, seed=3)
    names = [f"f{i}" for i in range(len(frames))]
    guess = [3.0, -2.0, 0.0, 0.0, 0.0, 0.0]
    e = ca.HipCorrelationEngine(fitting_model=model)
    t = tk.SequenceTracker(model, tk.DOMAIN_RECT, deformation, tk.REF_PREVIOUS, tk.ERRMODE_STOP_FRAME, guess, lib=e.lib)
    o = oracle.Oracle(model=model)
    o.set_image(0, frames[0])
    o.set_image(1, frames[1])
    m = mo.ManagerOracle(o, model, tk.DOMAIN_RECT, deformation, tk.REF_PREVIOUS, mo.ERRMODE_STOP_FRAME, guess)
    args = (150.0, 20.0, 243.0, 120.0, 196.0, 70.0, 3, 3)      # close to the right and upper borders
    t.set_rect_domain(*args), m.set_rect_domain(*args)
    assert tk.run_sequence(e, t, frames, names) == len(frames) - 1
    for k in range(len(frames) - 1):
        if k > 0:
            o.und_from_def()
            o.set_image(2, frames[k + 1])
            o.def_from_nxt()
        m.run_frame(k, names[k], names[k + 1])
    head_g, got = _report_table(t.report())
    head_w, want = _report_table(m.report_text())
    assert head_g == head_w and len(got) == len(want)
    col = {name: i for i, name in enumerate(head_g)}
    errors = 0
    for rg, rw in zip(got, want):
        for name in ("Frame#", "number_of_points", "error_status", "error_code"):
            assert rg[col[name]] == rw[col[name]], (rg[0], name, rg[col[name]], rw[col[name]])
        errors += rg[col["error_status"]] == "1"
        if rg[col["error_status"]] == "0":   # (a failed sector reports the reference's stale count)
            assert abs(int(rg[col["iterations"]]) - int(rw[col["iterations"]])) <= 1
        for name in ("und_center_x", "und_center_y", "def_center_x", "def_center_y", "parameter_0", "parameter_1",
                     "Initial_guess_0", "Initial_guess_1"):
            a, b = float(rg[col[name]]), float(rw[col[name]])
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 2e-2, (rg[0], name, a, b)
    assert errors > 0, "the domain was meant to lose sectors at the border"
    e.close(), t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model,reference", [(ca.FM_UVUXUYVXVY, 0), (ca.FM_UVQ, 0), (ca.FM_UV, 1), (ca.FM_U, 0),
                                             (ca.FM_UVUXUYVXVY, 1)])
def test_overlapped_frame_loop_equals_the_synchronous_one(monkeypatch, model, reference):
    """Eulerian sequences on a rectangular grid launch pair k+1 from device-computed guesses before
    the tracker has digested pair k (lk_sequence_run).  The report must be the text of the
    one-pair-at-a-time loop (LK_SEQ_SYNC=1), and the guesses the device used must be the tracker's
    own, bit for bit (LK_SEQ_CHECK=1 makes the loop fail otherwise)."""
    from correlation_amd import tracker as tk
    frames = ca.speckle.speckle_sequence(320, 288, 6, velocity=(0.8, -0.45), dilation=5e-4, seed=21)
    names = [f"f{i}" for i in range(len(frames))]
    guess = [0.4, -0.2, 1e-3, 5e-4, -5e-4, 2e-3]

    def run(sync):
        monkeypatch.setenv("LK_SEQ_SYNC", "1" if sync else "0")
        monkeypatch.setenv("LK_SEQ_CHECK", "1")
        e = ca.HipCorrelationEngine(fitting_model=model)
        e.set_batch_invariant(True)
        t = tk.SequenceTracker(model, tk.DOMAIN_RECT, tk.DEF_EULERIAN, reference, tk.ERRMODE_CONTINUE, guess, lib=e.lib)
        t.set_rect_domain(30.0, 34.0, 289.0, 251.0, 160.0, 144.0, 9, 7)
        assert tk.run_sequence(e, t, frames, names) == len(frames) - 1
        text, res = t.report(), t.results().tobytes()
        e.close(), t.close()
        return text, res

    want, got = run(True), run(False)
    assert got[0] == want[0] and got[1] == want[1]
    assert want[0].count("\n") == 1 + 5 * 63
    all_frames = frames
    for n in (2, 3):        # one pair (nothing to prefetch) and two
        frames = all_frames[:n]
        assert run(True) == run(False)


@pytest.mark.gpu
def test_windows_under_the_stop_all_policy_end_where_the_synchronous_loop_ends(monkeypatch):
    """stopAll (manager_class.cpp:1485-1486): the sequence ends with the first frame in which a sector fails.  The frame
    loop computes its windows ahead and discards what lies behind that frame; no frame before it had an error, so the
    report must be the synchronous loop's text, and the same number of pairs must have been correlated.  Frame 6 of ten is
    blank: no gradient, every sector of pair 5 fails."""
    from correlation_amd import tracker as tk
    frames = ca.speckle.speckle_sequence(320, 288, 10, velocity=(0.8, -0.45), dilation=5e-4, seed=21)
    frames[6] = np.full_like(frames[6], 90)
    names = [f"f{i}" for i in range(len(frames))]

    def run(sync, window):
        monkeypatch.setenv("LK_SEQ_SYNC", "1" if sync else "0")
        monkeypatch.setenv("LK_SEQ_WINDOW", str(window))
        e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY)
        e.set_batch_invariant(True)
        t = tk.SequenceTracker(ca.FM_UVUXUYVXVY, tk.DOMAIN_RECT, tk.DEF_EULERIAN, tk.REF_FIRST, tk.ERRMODE_STOP_ALL,
                               [0.4, -0.2, 1e-3, 5e-4, -5e-4, 2e-3], lib=e.lib)
        t.set_rect_domain(60.0, 60.0, 259.0, 227.0, 160.0, 144.0, 6, 5)    # (well inside the image: no sector leaves it)
        done = tk.run_sequence(e, t, frames, names)
        text, res = t.report(), t.results().tobytes()
        e.close(), t.close()
        return done, text, res

    want = run(True, 1)
    assert 2 <= want[0] < len(frames) - 1, "the sequence is meant to stop in the middle"
    for window in (3, 16):
        assert run(False, window) == want, window


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_update_sector_equals_the_frame_loop(mode):
    """CudaClass::updatePolygon's call shape (one sector, the engine's own last record) must move
    a sector exactly like the tracker-driven frame loop does (strict Lagrangian 0, Lagrangian 1)."""
    from correlation_amd import tracker as tk
    frames = ca.speckle.speckle_sequence(256, 256, 3, velocity=(1.1, -0.6), dilation=3e-4, seed=5)
    model = ca.FM_UVUXUYVXVY

    def setup():
        e = ca.HipCorrelationEngine(fitting_model=model)
        e.set_batch_invariant(True)   # (bit for bit: the frame loop moves its lists on the device, with their row-major
        #                                evaluation copy; the per-sector path re-commits host lists, which have none)
        e.set_undeformed_image(frames[0])
        e.set_deformed_image(frames[1])
        return e

    # A: the frame loop
    ea = setup()
    t = tk.SequenceTracker(model, tk.DOMAIN_RECT, mode, tk.REF_PREVIOUS, lib=ea.lib)
    t.set_rect_domain(40.0, 44.0, 215.0, 211.0, 127.5, 127.5, 3, 3)
    tk.sequence_frame(ea, t, 0)
    first = t.results()
    ea.makeUndPyramidFromDef()
    ea.set_deformed_image(frames[2])
    tk.sequence_frame(ea, t, 1)
    want = t.results()
    # B: per-sector calls
    eb = setup()
    cmds, guesses = tk.SequenceTracker(model, tk.DOMAIN_RECT, mode, tk.REF_PREVIOUS, lib=eb.lib), None
    cmds.set_rect_domain(40.0, 44.0, 215.0, 211.0, 127.5, 127.5, 3, 3)
    c0, g0 = cmds.begin_frame(0)
    for s, c in enumerate(c0):
        eb.resetPolygon_rect(s, int(c["x0"]), int(c["y0"]), int(c["x1"]), int(c["y1"]))
    eb.commit_sectors()
    r0 = eb.correlate_all(g0)
    assert np.array_equal(r0["p"], first["resulting_parameters"])
    eb.makeUndPyramidFromDef()
    eb.set_deformed_image(frames[2])
    for s in range(len(c0)):
        eb.update_sector(s, mode)
    r1 = eb.correlate_all(r0["p"])
    assert np.array_equal(r1["und_cx"], want["und_center_x"]) and np.array_equal(r1["und_cy"], want["und_center_y"])
    assert np.array_equal(r1["n_points"], want["number_of_points"])
    assert r1["p"].tobytes() == want["resulting_parameters"].tobytes()
    assert r1["chi"].tobytes() == want["chi"].tobytes()
    ea.close(), eb.close(), t.close(), cmds.close()


@pytest.mark.gpu
def test_finisher_of_parked_sectors_is_bit_identical(oracle, monkeypatch):
    """The lanes of the one-lane-per-sector kernel park a sector after LK_EVAL_CAP evaluations
    and a 16-lane finisher resumes it in the middle of its level with the products summed in
    the reference's sample order (row_newbcast) and the same QR: every record must keep its
    bits whatever the cap (0 = no finisher, 1 = everything goes through it), including explicit
    lists, all models, and against the CPU oracle on the starved level alone."""
    und, dfm = ca.speckle.speckle_pair(768, 768, p=(1.3, -0.7, 0.0005, 0.0, 0.0, -0.00025), seed=13)
    xdim, ydim, cen = oracle.rect_sector_geometry(32.0, 32.0, 735.0, 735.0, 39, 39)
    lists = [oracle.rect_points(cx - xdim, cy - ydim, cx + xdim, cy + ydim) for cx, cy in cen]

    def run(cap, model, py_start, explicit):
        monkeypatch.setenv("LK_EVAL_CAP", str(cap))
        e = ca.HipCorrelationEngine(fitting_model=model, py_start=py_start, py_stop=3)
        e.set_batch_invariant(True)   # the finer levels run in two passes (parked / not parked):
        e.set_undeformed_image(und)    # only batch-invariant records are comparable bit for bit
        e.set_deformed_image(dfm)
        if explicit:   # every third sample of each sector: explicit lists, centre = their mean
            for s, pts in enumerate(lists[:400]):
                e.set_sector_points(s, pts[::3].copy())
        else:
            e.set_rect_grid(32.0, 32.0, 735.0, 735.0, 39, 39)
        e.commit_sectors()
        got = e.correlate_all(np.zeros(6, np.float32))
        e.close()
        return got

    o = oracle.Oracle(py_start=3, py_stop=3)
    o.set_image(0, und)
    o.set_image(1, dfm)
    want3 = o.correlate_sectors(lists, centers=cen.astype(np.float32))
    for cap in (1, 3, 16):
        assert run(cap, ca.FM_UVUXUYVXVY, 3, False).tobytes() == want3.tobytes(), cap
    for model, py_start, explicit in ((ca.FM_UVUXUYVXVY, 0, False), (ca.FM_UVQ, 0, False), (ca.FM_UV, 2, False),
                                      (ca.FM_UVUXUYVXVY, 0, True)):
        ref = run(0, model, py_start, explicit)
        for cap in (1, 5):
            assert run(cap, model, py_start, explicit).tobytes() == ref.tobytes(), (model, py_start, explicit, cap)


@pytest.mark.gpu
def test_sharded_sequence_on_the_engine():
    """correlation_amd.distributed.ShardedSequence (the multi-GPU shape of config 4: tracker
    everywhere, sectors sharded) on one rank drives the HIP engine through the same calls as
    lk_sequence_run and must write the same report."""
    from correlation_amd import tracker as tk
    from correlation_amd.distributed import ShardedSequence
    frames = ca.speckle.speckle_sequence(256, 256, 4, velocity=(0.9, -0.5), dilation=4e-4, seed=3)
    names = [f"f{i}" for i in range(4)]
    reports = []
    for sharded in (False, True):
        e = ca.HipCorrelationEngine()
        t = tk.SequenceTracker(ca.FM_UVUXUYVXVY, tk.DOMAIN_RECT, tk.DEF_LAGRANGIAN, tk.REF_PREVIOUS, lib=e.lib)
        t.set_rect_domain(40.0, 44.0, 215.0, 211.0, 127.5, 127.5, 4, 3)
        if sharded:
            assert ShardedSequence(e, t).run(frames, names) == 3
        else:
            assert tk.run_sequence(e, t, frames, names) == 3
        reports.append(t.report())
        e.close(), t.close()
    assert reports[0] == reports[1]


@pytest.mark.gpu
def test_tiny_and_ragged_sectors(oracle, speckle512):
    """Sectors of 1, 2, 3, 7 and 13 samples, some at odd coordinates only so that coarser pyramid
    levels are EMPTY (n_L = 0: the reference's 1/n scaling becomes inf and its sums NaN), next to
    a normal sector.  Starved levels are solved in the reference's order with its QR, so even
    these degenerate records must match the oracle bit for bit wherever every level is starved,
    and the normal sector must not be disturbed by its neighbours."""
    e, o = make_pair(speckle512, ca.FM_UVUXUYVXVY, ca.IM_BICUBIC, oracle)
    lists = [np.array([[201, 203]], np.float32),
             np.array([[211, 203], [213, 205]], np.float32),
             np.array([[220, 220], [221, 221], [222, 224]], np.float32),
             oracle.rect_points(240, 240, 246, 240),
             np.array([[260 + (i % 4), 260 + i // 4] for i in range(13)], np.float32),
             oracle.rect_points(300, 300, 330, 330)]
    for s, xy in enumerate(lists):
        e.set_sector_points(s, xy)
    e.commit_sectors()
    got = e.correlate_all(np.zeros(6, np.float32))
    want = o.o1.correlate_sectors(lists)
    assert list(got["n_points"]) == [1, 2, 3, 7, 13, 961]
    assert np.array_equal(got["error_code"], want["error_code"])
    for s in range(4):   # every level of these sectors has at most 12 samples: bit-identical
        same = got[s].tobytes() == want[s].tobytes()
        same = same or (np.array_equal(got["p"][s], want["p"][s], equal_nan=True) and
                        got["iterations"][s] == want["iterations"][s])
        assert same, (s, got[s], want[s])
    assert np.abs(got["p"][5] - want["p"][5])[:2].max() < 1e-3
    alone, _ = e.correlate(5, np.zeros(6, np.float32))
    assert np.abs(alone["p"] - got["p"][5]).max() < 1e-4
    e.close()


@pytest.mark.gpu
def test_separable_bicubic_extension(oracle, speckle512):
    """LK_IM_BICUBIC_SEPARABLE (an extension, never the default): the reference's bicubic patch is
    the Catmull-Rom spline in exact arithmetic; evaluated in separable form its value and
    gradient equal the reference-order evaluation to float rounding, the validity rule is the
    same, and whole solves stay within 5e-3 px / 1e-4 / 0.5 % chi of the reference."""
    und, dfm = speckle512
    e = ca.HipCorrelationEngine(interpolation=ca.IM_BICUBIC_SEPARABLE)
    e.set_deformed_image(dfm)
    rng = np.random.default_rng(23)
    h, w = dfm.shape
    pts = np.stack([rng.uniform(-2, w + 2, 4000), rng.uniform(-2, h + 2, 4000)], 1).astype(np.float32)
    pts[:50] = np.round(pts[:50])
    pts[50:60] = [[1.0, 5.0]] * 10
    got = e.sample(ca.IMG_DEF, 0, pts)
    want = oracle.interpolate_many(ca.IM_BICUBIC, dfm, pts)
    assert np.array_equal(got[:, 3], want[:, 3])
    ok = want[:, 3] == 0
    # float64 Catmull-Rom as the yardstick: the separable float32 form is CLOSER to it than the
    # reference-order monomial evaluation (whose dx in [1,2) monomials cancel at ~1e-3 grey levels)
    x, y = pts[ok, 0].astype(np.float64), pts[ok, 1].astype(np.float64)
    ix, iy = np.floor(x).astype(int), np.floor(y).astype(int)
    tx, ty = x - ix, y - iy

    def cr(t):
        return (np.stack([t * ((2 - t) * t - 1), t * t * (3 * t - 5) + 2, t * ((4 - 3 * t) * t + 1), t * t * (t - 1)]) / 2,
                np.stack([(4 - 3 * t) * t - 1, t * (9 * t - 10), (8 - 9 * t) * t + 1, t * (3 * t - 2)]) / 2)

    (wx, gx), (wy, gy) = cr(tx), cr(ty)
    win = np.stack([[dfm[iy + j - 1, ix + i - 1].astype(np.float64) for i in range(4)] for j in range(4)])  # [j][i][n]
    exact = np.stack([np.einsum("jn,in,jin->n", wy, wx, win), np.einsum("jn,in,jin->n", wy, gx, win),
                      np.einsum("jn,in,jin->n", gy, wx, win)], 1)
    assert np.abs(got[ok, :3] - exact).max() < 2e-4               # grey levels, values up to 255
    assert np.abs(want[ok, :3] - exact).max() < 1e-2
    assert np.abs(got[ok, :3] - exact).max() < np.abs(want[ok, :3] - exact).max()
    e.close()
    e, o = make_pair(speckle512, ca.FM_UVUXUYVXVY, ca.IM_BICUBIC, oracle)
    e.close()
    e = ca.HipCorrelationEngine(interpolation=ca.IM_BICUBIC_SEPARABLE, fitting_model=ca.FM_UVUXUYVXVY)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    e.set_rect_grid(24.0, 24.0, 487.0, 487.0, 20, 20)
    e.commit_sectors()
    xdim, ydim, cen = oracle.rect_sector_geometry(24.0, 24.0, 487.0, 487.0, 20, 20)
    lists = [oracle.rect_points(cx - xdim, cy - ydim, cx + xdim, cy + ydim) for cx, cy in cen]
    got = e.correlate_all(np.zeros(6, np.float32))
    want = o.o1.correlate_sectors(lists, centers=cen.astype(np.float32))
    # Looser bars than compare_results on purpose: this mode is closer to the exact spline than
    # the reference is, so it cannot also sit inside the reference's own rounding (its chi
    # agrees to 1e-5 on about half of the sectors, the reference with itself on two thirds) -
    # which is why it is an opt-in extension and never the default.
    assert np.array_equal(got["error_code"], want["error_code"])
    assert (np.abs(got["iterations"] - want["iterations"]) <= 1).mean() >= 0.95
    assert np.abs(got["p"] - want["p"])[:, :2].max() < 5e-3
    assert np.abs(got["p"] - want["p"])[:, 2:].max() < 1e-4
    assert (np.abs(got["chi"] - want["chi"]) / want["chi"]).max() < 5e-3
    e.close()


@pytest.mark.gpu
def test_batch_invariant_records_across_batch_sizes(speckle512):
    """lk_set_batch_invariant(1): no solo / adaptive width and the lane group chosen from the
    sector's own size - so a sector keeps its bits in the full grid, in shards of any size and
    alignment, and alone (the default mode picks wider groups for small batches)."""
    def run(first=0, count=-1):
        e, _ = make_pair(speckle512, ca.FM_UVUXUYVXVY)
        e.set_batch_invariant(True)
        e.set_rect_grid(24.0, 24.0, 487.0, 487.0, 23, 23, first, count)
        e.commit_sectors()
        got = e.correlate_all(np.zeros(6, np.float32))
        one, _ = e.correlate(min(5, len(got) - 1), np.zeros(6, np.float32))
        assert one.tobytes() == got[min(5, len(got) - 1)].tobytes()
        e.close()
        return got

    full = run()
    for first, count in ((0, 7), (101, 64), (300, 229)):
        assert run(first, count).tobytes() == full[first:first + count].tobytes(), (first, count)


def _smooth_texture(n, seed):
    """A random texture with a few pixels of correlation length, cheap at 8192^2 (box blurs by
    cumulative sums; the speckle renderer would take minutes on the CPU at that size)."""
    rng = np.random.default_rng(seed)
    a = rng.random((n, n), dtype=np.float32)
    for _ in range(3):
        for axis in (0, 1):
            c = np.cumsum(a, axis=axis, dtype=np.float64)
            k = 5
            c = np.take(c, np.arange(k, n), axis=axis) - np.take(c, np.arange(0, n - k), axis=axis)
            pad = [(0, 0), (0, 0)]
            pad[axis] = (k // 2, k - k // 2)
            a = np.pad(c.astype(np.float32) / k, pad, mode="edge")
    a -= a.min()
    return (a * (255.0 / a.max())).astype(np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("size,hs,n0,stop,sectors", [(2048, 224, 49, 2, 50176), (8192, 447, 289, 3, 199809)])
def test_configs_4_and_5_full_size_properties(size, hs, n0, stop, sectors):
    """BASELINE configs 4 (one pair of its 50 176-sector grid) and 5 (199 809 sectors, 4 levels,
    8192^2) at full size, through properties that need no oracle: the deformed frame is the
    undeformed one moved by whole pixels, so every sector's translation is known exactly; the
    batch is deterministic; in batch-invariant mode a shard of the grid reproduces the full run
    bit for bit; the counters add up."""
    und = _smooth_texture(size, 5)
    dfm = np.roll(und, shift=(-2, 3), axis=(0, 1))          # content moves by u = +3, v = -2
    lo_, hi_ = (24.0, size - 25.0) if size == 2048 else (32.0, size - 33.0)

    def engine(invariant, first=0, count=-1):
        e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY, py_stop=stop)
        e.set_batch_invariant(invariant)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        e.set_rect_grid(lo_, lo_, hi_, hi_, hs, hs, first, count)
        e.commit_sectors()
        return e

    e = engine(False)
    assert e.n_sectors == sectors and e.sector_info(0)[0] == n0
    r1 = e.correlate_all(np.zeros(6, np.float32))
    r2 = e.correlate_all(np.zeros(6, np.float32))
    assert r1.tobytes() == r2.tobytes(), "the batch solve must be deterministic"
    st = e.stats()
    assert st["sectors"] == sectors and st["point_iterations"] >= (stop + 1) * sectors
    assert st["algorithmic_bytes"] == 25 * st["sample_evaluations"] + 196 * st["evaluations"]
    ok = r1["error_code"] == 0
    assert ok.mean() > 0.97
    # 7x7 samples cannot pin six parameters on every patch of texture (config 4's sectors are
    # that small by the reference's own grid rule): the bulk must sit on the known shift
    du, dv = np.abs(r1["p"][ok][:, 0] - 3.0), np.abs(r1["p"][ok][:, 1] + 2.0)
    assert np.median(du) < 5e-3 and np.median(dv) < 5e-3
    assert ((du < 0.2) & (dv < 0.2)).mean() > (0.5 if n0 < 100 else 0.7)   # (a starved level 3 scatters some)
    assert np.median(np.abs(r1["p"][ok][:, 2:])) < 1e-3
    assert (r1["n_points"] == n0).all()
    e.close()
    full = engine(True)
    rf = full.correlate_all(np.zeros(6, np.float32))
    full.close()
    assert np.median(np.abs(rf["p"][ok] - r1["p"][ok])[:, :2]) < 1e-4    # same solutions, other summation grouping
    first, count = sectors // 3 + 1, 4097
    part = engine(True, first, count)
    rp = part.correlate_all(np.zeros(6, np.float32))
    part.close()
    assert rp.tobytes() == rf[first:first + count].tobytes()


@pytest.mark.gpu
def test_cpp_example_tracks_a_pgm_sequence(tmp_path):
    """examples/track_sequence.cpp - a plain C++ program on the C ABI (lk_create, lk_tracker_*,
    lk_sequence_run with its own PGM frame provider, lk_tracker_report) - must write the report
    the Python-driven frame loop writes for the same frames."""
    import subprocess
    from correlation_amd import tracker as tk
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "track_sequence"
    cmd = ["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "track_sequence.cpp"),
           "-L" + os.path.join(root, "correlation_amd"), "-llk_engine", "-Wl,-rpath," + os.path.join(root, "correlation_amd"),
           "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    frames = ca.speckle.speckle_sequence(256, 256, 4, velocity=(0.9, -0.5), dilation=4e-4, seed=3)
    paths = []
    for i, f in enumerate(frames):
        p = tmp_path / f"f{i}.pgm"
        p.write_bytes(b"P5\n256 256\n255\n" + f.tobytes())
        paths.append(str(p))
    for mode, deformation, ref in (("eulerian", tk.DEF_EULERIAN, tk.REF_FIRST), ("lagrangian", tk.DEF_LAGRANGIAN, tk.REF_PREVIOUS)):
        out = tmp_path / f"report_{mode}.csv"
        r = subprocess.run([str(exe), str(out), mode, "4", "3"] + paths, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "3 pairs, 12 sectors" in r.stdout
        e = ca.HipCorrelationEngine()
        t = tk.SequenceTracker(ca.FM_UVUXUYVXVY, tk.DOMAIN_RECT, deformation, ref, lib=e.lib)
        t.set_rect_domain(24.0, 24.0, 231.0, 231.0, 127.5, 127.5, 4, 3)
        assert tk.run_sequence(e, t, frames, paths) == 3
        assert out.read_text() == t.report()
        e.close(), t.close()


@pytest.mark.gpu
def test_row_distributed_qr_is_bit_identical(oracle):
    """The finisher of starved levels runs the restated Eigen QR spread over the 16 lanes of a
    row (one column per lane, positions instead of column swaps, ds_bpermute for what the
    pivot owns).  lk_damped_solve(reference_solver=2) drives it; every step must equal the
    one-lane restatement (reference_solver=1) and the oracle bit for bit - on healthy systems,
    on rank-deficient ones (fewer samples than parameters: the pivot order decides), with
    exact zeros, ties between column norms, and non-finite entries."""
    e = ca.HipCorrelationEngine()
    rng = np.random.default_rng(31)

    def check(A, b, lam, s, label):
        one = e.damped_solve(A, b, np.float32(lam), np.float32(s), reference_solver=1)
        row = e.damped_solve(A, b, np.float32(lam), np.float32(s), reference_solver=2)
        assert one.tobytes() == row.tobytes(), (label, one, row)
        if np.isfinite(A).all():   # (x86 and gfx950 give the default NaN different sign bits)
            want = oracle.damped_solve(A, b, np.float32(lam), np.float32(s))
            assert np.array_equal(np.isnan(one), np.isnan(want)), label
            assert one[~np.isnan(one)].tobytes() == want[~np.isnan(want)].tobytes(), label

    for n in (1, 2, 3, 6):
        for m in (1, 2, 3, 4, 5, 9, 60):                      # samples: rank min(m, n)
            for _ in range(12):
                J = rng.standard_normal((m, n)) * rng.uniform(0.5, 20, n)
                A = (J.T @ J).astype(np.float32)
                b = (J.T @ rng.standard_normal(m)).astype(np.float32)
                check(A, b, 10.0 ** rng.uniform(-9, 1), 1.0 / m, (n, m))
    # ties and exact zeros
    check(np.eye(6, dtype=np.float32), np.arange(6, dtype=np.float32), 0.0, 1.0, "identity")
    check(np.zeros((6, 6), np.float32), np.ones(6, np.float32), 1e-4, 1.0, "zero matrix")
    A = np.diag([4.0, 0.0, 9.0, 0.0, 4.0, 9.0]).astype(np.float32)
    check(A, np.array([8, 1, 18, 1, 8, 18], np.float32), 0.0, 1.0, "semi-definite with ties")
    A = np.ones((6, 6), np.float32)
    check(A, np.ones(6, np.float32), 1e-9, 1.0, "rank one")
    # non-finite input (an empty pyramid level scales by 1/0): same NaN pattern, same finite bits
    A = (rng.standard_normal((6, 6)) ** 2).astype(np.float32)
    A = np.triu(A) + np.triu(A, 1).T
    for bad in (np.inf, np.nan):
        B = A.copy()
        B[2, 2] = bad
        one = e.damped_solve(B, np.ones(6, np.float32), np.float32(1e-4), np.float32(1.0), reference_solver=1)
        row = e.damped_solve(B, np.ones(6, np.float32), np.float32(1e-4), np.float32(1.0), reference_solver=2)
        assert np.array_equal(np.isnan(one), np.isnan(row)) and np.array_equal(one[~np.isnan(one)], row[~np.isnan(row)])
    check(A, np.ones(6, np.float32), 1e-4, np.inf, "infinite scaling")
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,model,interp", [(1, ca.FM_UVUXUYVXVY, ca.IM_BICUBIC), (2, ca.FM_UVUXUYVXVY, ca.IM_BICUBIC),
                                               (3, ca.FM_UVUXUYVXVY, ca.IM_BICUBIC), (4, ca.FM_UVQ, ca.IM_BILINEAR),
                                               (5, ca.FM_UV, ca.IM_NEAREST), (6, ca.FM_U, ca.IM_BICUBIC),
                                               (7, ca.FM_UVUXUYVXVY, ca.IM_BILINEAR)])
def test_random_mix_of_sectors(oracle, seed, model, interp):
    """Sixty sectors of every size class at once - 7x7 ... 301x301 rectangles, decimated explicit
    lists - on a 4-level pyramid, so that one engine runs the one-lane kernel, the finisher, 16-
    and 32-lane groups with alignment / adaptive width / solo, workgroup groups and teams in
    one solve.  Same bars as everything else; and in batch-invariant mode single-sector calls
    reproduce the batch bit for bit."""
    rng = np.random.default_rng(seed)
    und, dfm = ca.speckle.speckle_pair(768, 768, p=(1.1, -0.6, 0.0007, 0.0003, -0.0002, 0.0009), seed=30 + seed)
    lists, cens, specs = [], [], []
    for k in range(60):
        half = int(rng.choice([3, 3, 4, 9, 9, 9, 22, 22, 60, 150]))
        cx, cy = int(rng.integers(half + 12, 768 - half - 12)), int(rng.integers(half + 12, 768 - half - 12))
        pts = oracle.rect_points(cx - half, cy - half, cx + half, cy + half)
        explicit = half >= 3 and rng.random() < 0.3
        if explicit:
            pts = pts[rng.random(len(pts)) < 0.6].copy()
        lists.append(pts)
        cens.append((float(cx), float(cy)))
        specs.append((explicit, cx - half, cy - half, cx + half, cy + half))

    def engine(invariant):
        e = ca.HipCorrelationEngine(fitting_model=model, interpolation=interp, py_stop=3)
        e.set_batch_invariant(invariant)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        for s, (explicit, x0, y0, x1, y1) in enumerate(specs):
            if explicit:
                e.set_sector_points(s, lists[s], center=cens[s])
            else:
                e.resetPolygon_rect(s, x0, y0, x1, y1)
        e.commit_sectors()
        return e

    os_ = []
    for T, solver in ((1, 0), (8, 0), (1, 2)):
        o = oracle.Oracle(n_threads=T, solver=solver, py_stop=3, model=model, interp=interp)
        o.set_image(0, und)
        o.set_image(1, dfm)
        os_.append(o)
    want = OraclePair(*os_).correlate_sectors(lists, centers=np.array(cens, np.float32))
    n0 = np.array([len(x) for x in lists])
    big = n0 >= 100

    def check(got, label):
        # sectors of a few dozen samples are chaotic for the reference itself (its own thread
        # count moves some of them by 0.015 px): strict bars for the others, loose ones there
        compare_results(got[big], tuple(w[big] for w in want), label)
        g, w, w8 = got[~big], want[0][~big], want[1][~big]
        assert np.array_equal(g["error_code"], w["error_code"]) and np.array_equal(g["n_points"], w["n_points"])
        ok = w["error_code"] == 0
        d, d8 = np.abs(g["p"] - w["p"])[ok][:, :2], np.abs(w8["p"] - w["p"])[ok][:, :2]
        assert d.max() <= max(0.05, 3 * d8.max()), (label, d.max(), d8.max())
        assert np.median(d) < 1e-4

    e = engine(False)
    got = e.correlate_all(np.zeros(6, np.float32))
    e.close()
    check(got, f"random mix, seed {seed}")
    e = engine(True)
    inv = e.correlate_all(np.zeros(6, np.float32))
    check(inv, f"random mix, seed {seed}, batch-invariant")
    for s in rng.choice(60, 8, replace=False):
        one, _ = e.correlate(int(s), np.zeros(6, np.float32))
        assert one.tobytes() == inv[s].tobytes(), s
    e.close()


@pytest.mark.gpu
def test_device_roi_masks_equal_the_host_scans(oracle, monkeypatch):
    """Annular and blob sectors are rasterised by the device mask at commit (lk_roi_tile_kernel: the CPU
    engine's predicates on the bounding box walked x outer / y inner, the blob's scan lines from the
    host ear clipper, order-preserving compaction) - the counterpart of cudaPolygon's thrust
    rasterise + remove_if (cuda_polygon.cuh:180-292).  Lists, centres, per-level counts and records
    must equal the host scans' (LK_HOST_ROI=1) and the oracle's, bit for bit - also for the reference's
    own polygonBlob_class outputs and for sectors mixed in one domain."""
    und, dfm = ca.speckle.speckle_pair(768, 768, p=(0.9, 0.4, 0.001, 0.0005, -0.0005, 0.0015), seed=21)
    g = np.load(os.path.join(GOLD, "ref_blob.npz"))
    blobs = [n[:-len("_contour")] for n in g.files if n.endswith("_contour") and int(g[n[:-len("_contour")] + "_count"][0]) > 0]
    assert len(blobs) >= 5
    rs, as_ = 3, 5
    dr, da = np.float32((330.0 - 90.0) / rs), np.float32(2 * np.pi) / np.float32(as_)
    ann = [(np.float32(90.0 + i * dr), dr, np.float32(j) * da, da, 384.0, 380.0, as_) for i in range(rs) for j in range(as_)]
    ann.append((30.0, 55.0, 0.0, np.float32(2 * np.pi), 384.0, 384.0, 1))     # a full ring

    monkeypatch.setenv("LK_EVAL_LISTS", "0")   # (records bit for bit: both paths walk the lists in the reference's order;
    #                                               the row-major evaluation copy has its own test below)

    def build(host):
        monkeypatch.setenv("LK_HOST_ROI", "1" if host else "0")
        e = ca.HipCorrelationEngine(py_stop=3)
        e.set_batch_invariant(True)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        s = 0
        for q in ann[:7]:
            e.resetPolygon_annular(s, *q)
            s += 1
        e.set_sectors_annular(s, np.float32([q[:6] for q in ann[7:-1]]), as_)
        s += len(ann) - 8
        e.resetPolygon_annular(s, *ann[-1])
        s += 1
        for name in blobs:
            c = g[f"{name}_contour"].astype(np.float32)
            c = (c - c.mean(0)) * np.float32(200.0 / np.abs(c - c.mean(0)).max()) + np.float32([384.0, 384.0])   # inside the image
            e.resetPolygon_blob(s, c)
            s += 1
        before = e.getUndXY0ToCPU(2).tobytes()           # a list asked for BEFORE the commit: the host scan on demand
        e.commit_sectors()
        S = e.n_sectors
        out = ([e.getUndXY0ToCPU(k).tobytes() for k in range(S)], [e.sector_info(k) for k in range(S)],
               [[e.sector_level_count(k, l) for l in range(4)] for k in range(S)], e.correlate_all(np.zeros(6, np.float32)).tobytes())
        assert out[0][2] == before
        e.close()
        return out

    dev, host = build(False), build(True)
    for i, (a, b) in enumerate(zip(dev, host)):
        assert a == b, i
    # ... and the oracle's lists for the annular sectors
    for k, q in enumerate(ann):
        want = oracle.annular_points(*q)
        assert np.frombuffer(dev[0][k], np.float32).reshape(-1, 2).tobytes() == want.tobytes(), k
    assert min(n for n, _, _ in dev[1]) > 500
    # the reference's own scan-fill outputs, through the device mask
    e = ca.HipCorrelationEngine()
    for name in blobs:
        e.clear_sectors()
        e.resetPolygon_blob(0, g[f"{name}_contour"])
        e.commit_sectors()
        assert np.array_equal(e.getUndXY0ToCPU(0), g[f"{name}_pts"].astype(np.float32)), name
    e.close()


@pytest.mark.gpu
def test_evaluation_copy_of_annular_lists_is_row_major_and_changes_only_the_summation_order(monkeypatch):
    """Annular lists come in the reference's order, x outer / y inner (manager_class.cpp:907-918): 64 consecutive
    samples lie on 64 image rows.  The lane groups of the default mode walk a second copy of every level's lists with
    each sector's samples y outer / x inner (LkLevelView::xy_eval, built by the same device masks walked row by row and
    the same decimation).  The copy must hold exactly the canonical list's samples, sorted by (y, x), at every level;
    a blob's scan lines are rows already; centres, counts and the reference-order records do not depend on it; the
    default mode's records move by summation-order noise only."""
    und, dfm = ca.speckle.speckle_pair(768, 768, p=(0.9, 0.4, 0.001, 0.0005, -0.0005, 0.0015), seed=21)
    rs, as_ = 3, 6
    dr, da = np.float32((330.0 - 90.0) / rs), np.float32(2 * np.pi) / np.float32(as_)
    t = 2 * np.pi * np.arange(24) / 24
    blob = np.stack([384 + np.where(np.arange(24) % 2 == 0, 300.0, 180.0) * np.cos(t),
                     384 + np.where(np.arange(24) % 2 == 0, 300.0, 180.0) * np.sin(t)], 1).astype(np.float32)

    def run(flag, ref_order):
        monkeypatch.setenv("LK_EVAL_LISTS", flag)
        e = ca.HipCorrelationEngine(py_stop=2)
        e.set_reference_order(ref_order)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        e.set_sectors_annular(0, np.float32([[np.float32(90.0 + i * dr), dr, np.float32(j) * da, da, 384.0, 380.0]
                                             for i in range(rs) for j in range(as_)]), as_)
        e.resetPolygon_blob(rs * as_, blob)
        e.commit_sectors()
        lists = None
        if flag == "1" and not ref_order:
            lists = [[(e.level_xy(l, k), e.level_xy(l, k, evaluation_copy=True)) for l in range(3)] for k in range(e.n_sectors)]
        out = [e.correlate_all(np.zeros(6, np.float32)), [e.sector_info(k) for k in range(e.n_sectors)], lists]
        if not ref_order:   # the lists move with the last result (Lagrangian descriptions): the copy moves with them
            e.rewarp_sectors()
            if flag == "1":
                out.append([[(e.level_xy(l, k), e.level_xy(l, k, evaluation_copy=True)) for l in range(3)] for k in range(e.n_sectors)])
            out.append(e.correlate_all(np.zeros(6, np.float32)))
        e.close()
        return out

    r1, info1, lists, moved_lists, m1 = run("1", 0)
    r0, info0, _, m0 = run("0", 0)
    assert info1 == info0                                   # counts and centres
    moved = 0
    for k, per_level in enumerate(lists):
        for l, (canon, ev) in enumerate(per_level):
            assert len(canon) == len(ev) and len(canon) > 0, (k, l)
            if k == rs * as_:   # the blob: scan lines are rows already (flat triangle by flat triangle, polygon_class.cpp:389-391)
                assert np.array_equal(canon, ev), l
                continue
            key = lambda a: a[:, 1].astype(np.int64) * 65536 + a[:, 0].astype(np.int64)
            assert np.array_equal(np.sort(key(canon)), key(ev)), (k, l)      # the same samples, sorted by (y, x)
            moved += int(not np.array_equal(canon, ev))
    assert moved >= 3 * rs * as_ - 3                        # the annular sectors really come in another order
    assert np.array_equal(r1["error_code"], r0["error_code"]) and (r1["error_code"] == 0).all()
    assert np.abs(r1["p"][:, :2] - r0["p"][:, :2]).max() < 5e-4
    assert (np.abs(r1["chi"] - r0["chi"]) / np.abs(r0["chi"])).max() < 3e-3
    assert (r1["iterations"] == r0["iterations"]).mean() >= 0.9
    # moved lists: the copy is still the canonical list's samples (float coordinates now), annular sectors in another order
    for k, per_level in enumerate(moved_lists):
        for l, (canon, ev) in enumerate(per_level):
            assert len(canon) == len(ev) and len(canon) > 0, (k, l)
            assert np.array_equal(canon[np.lexsort((canon[:, 0], canon[:, 1]))], ev[np.lexsort((ev[:, 0], ev[:, 1]))]), (k, l)
            assert k == rs * as_ or not np.array_equal(canon, ev), (k, l)
    assert np.array_equal(m1["error_code"], m0["error_code"]) and (m1["error_code"] == 0).all()
    assert np.abs(m1["p"][:, :2] - m0["p"][:, :2]).max() < 5e-4
    assert (np.abs(m1["chi"] - m0["chi"]) / np.abs(m0["chi"])).max() < 3e-3
    # the reference-order mode walks the canonical lists: byte-identical records with and without the copy
    assert run("1", 1)[0].tobytes() == run("0", 1)[0].tobytes()


@pytest.mark.gpu
def test_rectangles_that_become_lists_get_a_row_major_evaluation_copy(monkeypatch):
    """lk_rewarp_sectors turns implicit rectangles into explicit lists in the reference's order (x outer / y inner,
    manager_class.cpp:1607-1611); the lane groups would then walk columns.  The move writes a second copy row by row:
    the same samples at every level, and the next solve differs from the list-order walk by summation order only."""
    frames = ca.speckle.speckle_sequence(320, 288, 3, velocity=(0.7, -0.4), dilation=6e-4, seed=9)

    def run(flag):
        monkeypatch.setenv("LK_EVAL_LISTS", flag)
        e = ca.HipCorrelationEngine(py_stop=2)
        e.set_undeformed_image(frames[0])
        e.set_deformed_image(frames[1])
        e.set_rect_grid(30.0, 30.0, 289.0, 257.0, 5, 4)
        e.commit_sectors()
        g = np.zeros(6, np.float32)
        e.correlate_all(g)
        e.rewarp_sectors()
        e.makeUndPyramidFromDef()
        e.set_deformed_image(frames[2])
        lists = [[(e.level_xy(l, k), e.level_xy(l, k, evaluation_copy=True)) for l in range(3)] for k in range(e.n_sectors)] if flag == "1" else None
        r = e.correlate_all(g)
        e.close()
        return r, lists

    r1, lists = run("1")
    r0, _ = run("0")
    for k, per_level in enumerate(lists):
        for l, (canon, ev) in enumerate(per_level):
            assert len(canon) == len(ev) and len(canon) > 0, (k, l)
            assert np.array_equal(canon[np.lexsort((canon[:, 0], canon[:, 1]))], ev[np.lexsort((ev[:, 0], ev[:, 1]))]), (k, l)
            assert not np.array_equal(canon, ev), (k, l)
    # level 0 of the copy: rows of the rectangle, x fastest (the move is smooth: y still grows row by row)
    ev0 = lists[0][0][1]
    w = int(np.argmax(np.diff(ev0[:, 0]) < 0)) + 1          # the first row's length
    assert w > 8 and np.all(np.diff(ev0[:w, 0]) > 0) and abs(ev0[w, 1] - ev0[0, 1] - 1.0) < 0.1
    assert np.array_equal(r1["error_code"], r0["error_code"])
    ok = r1["error_code"] == 0
    assert ok.sum() >= 15
    assert np.abs(r1["p"][ok][:, :2] - r0["p"][ok][:, :2]).max() < 5e-4
    assert (np.abs(r1["chi"][ok] - r0["chi"][ok]) / np.abs(r0["chi"][ok])).max() < 3e-3


@pytest.mark.gpu
def test_mean_centre_of_long_integer_lists_is_the_sequential_float_mean():
    """The centre of a device-masked sector is the reference's SEQUENTIAL float32 mean of its list
    (pyramid_class.cpp:325-340), evaluated in parallel: exact chunk sums, parity maps for the predicted binade,
    one verifying walk (lk_mean_center_int_kernel).  Lists built to stress it - many binade crossings (large
    coordinates), none (small ones), crossings on one axis only, x sweeping the whole 15-bit range, lengths
    around the 8192-sample chunk and the 32768-sample pass - against numpy's sequential float32 accumulate on
    the very list the engine holds."""
    def rect(x0, y0, x1, y1):
        return np.float32([[x0, y0], [x1, y0], [x1, y1], [x0, y1]])

    shapes = [rect(30000.2, 10.3, 32001.7, 700.6),      # 1.4 M samples, sums up to 4e10 / 5e8
              rect(1.5, 1.5, 600.4, 500.2),             # small coordinates: long exact stretch, few crossings
              rect(0.3, 5.2, 32700.8, 7.9),             # x sweeps the range, y is tiny
              rect(20000.4, 30000.1, 20090.9, 30090.7), # both large, 8281 samples: just over one chunk
              rect(16384.5, 100.5, 16511.4, 356.6),     # 127 x 256 = 32512: just under one pass
              rect(5.5, 32000.5, 2500.4, 32700.4)]      # x small, y large
    e = ca.HipCorrelationEngine(py_stop=1)
    for s, c in enumerate(shapes):
        e.resetPolygon_blob(s, c)
    e.commit_sectors()
    total = 0
    for s in range(len(shapes)):
        xy = e.getUndXY0ToCPU(s)
        n, cx, cy = e.sector_info(s)
        assert n == len(xy) and n > 8000
        total += n
        assert (xy == np.floor(xy)).all() and xy.min() >= 0 and xy.max() < 32768
        want_x = np.add.accumulate(xy[:, 0], dtype=np.float32)[-1] / np.float32(n)
        want_y = np.add.accumulate(xy[:, 1], dtype=np.float32)[-1] / np.float32(n)
        assert np.float32(cx) == want_x and np.float32(cy) == want_y, (s, n, cx, cy, want_x, want_y)
        # (and the sequential float mean is visibly NOT the exact mean on the long lists)
        if n > 1000000:
            assert abs(float(want_x) - float(xy[:, 0].astype(np.float64).mean())) > 1e-3
    assert total > 1500000
    e.close()



@pytest.mark.gpu
def test_kept_sums_save_evaluations_and_change_no_record(monkeypatch):
    """A rejected LM trip goes back to the last good parameters with a larger lambda.  The reference evaluates
    there again (correlation_class.cpp:441-499) and gets the sums it had; the one-lane kernel and the 16-lane
    groups keep those sums and solve from them at once.  With the cache switched off (LK_KEEP_SUMS=0) the engine
    takes the reference's extra evaluation: every record must keep its bits, in the default flavour (batch-invariant,
    so that records are comparable) and in reference-order mode, on sectors with singular coarse levels
    (7 x 7 samples, 3 levels: config 4's geometry) where rejections are frequent."""
    und, dfm = ca.speckle.speckle_pair(640, 640, p=(1.2, -0.6, 0.0004, 0.0, 0.0, -0.0002), seed=17)

    def run(keep, reference_order):
        monkeypatch.setenv("LK_KEEP_SUMS", "1" if keep else "0")
        e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY, py_stop=2)
        e.set_batch_invariant(True)
        e.set_reference_order(1 if reference_order else 0)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        e.set_rect_grid(24.0, 24.0, 615.0, 615.0, 66, 66)          # 4356 sectors of 7 x 7 samples
        e.commit_sectors()
        assert e.sector_info(0)[0] == 49
        rec = e.correlate_all(np.zeros(6, np.float32))
        st, per = e.stats(), e.sector_stats()
        e.close()
        return rec, st, per

    for reference_order in (False, True):
        with_cache, st1, per1 = run(True, reference_order)
        without, st0, per0 = run(False, reference_order)
        assert with_cache.tobytes() == without.tobytes(), reference_order
        assert st1["point_iterations"] == st0["point_iterations"]                 # the LM trips are the same trips
        assert (per1[:, 0] <= per0[:, 0]).all()                                   # never more evaluations with the cache
        saved = st0["evaluations"] - st1["evaluations"]
        assert saved > 0.08 * st0["evaluations"], (saved, st0["evaluations"])     # (config 4: 16 %)
        assert st1["evaluations"] >= st1["point_iterations"]                     # at least one evaluation per LM trip and level



@pytest.mark.gpu
def test_sectors_appended_one_by_one_equal_the_full_commit(speckle512):
    """lk_commit_sectors' append path (the reference's first-frame loop registers and solves sector after sector,
    manager_class.cpp:304-460): rectangles of different sizes - incl. one with a starved coarsest level and one that
    gets a whole wavefront - registered and committed one at a time, each solved on its own in between, must leave the
    engine in the state a single commit of all of them gives: same batch records, same single-sector records; a
    rectangle moved by lk_update_sector is patched in place; re-registering a committed sector falls back to the full
    rebuild and keeps the others' sequence state."""
    und, dfm = speckle512
    rects = [(40, 40, 58, 58), (100, 60, 140, 90), (200, 200, 206, 206), (300, 100, 330, 160), (60, 300, 78, 318),
             (150, 350, 168, 368), (400, 400, 440, 440), (260, 260, 278, 278)]

    def engine():
        e = ca.HipCorrelationEngine()
        e.set_batch_invariant(True)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        return e
    n = len(rects)
    ref = engine()                                      # the same rectangles in ONE commit, plus an explicit list behind them
    for s, r in enumerate(rects):
        ref.resetPolygon_rect(s, *r)
    ref.set_sector_points(n, np.float32([[10, 10], [11, 10], [10, 11], [11, 11], [12, 10], [12, 11]]))
    ref.commit_sectors()
    want_all = ref.correlate_all(np.zeros(6, np.float32))
    want = want_all[:n]
    e = engine()
    singles = []
    for s, r in enumerate(rects):
        e.resetPolygon_rect(s, *r)
        e.commit_sectors()                              # appends behind the committed ones (from the second on)
        singles.append(e.correlate(s, np.zeros(6, np.float32))[0])
    assert np.array(singles, dtype=want.dtype).tobytes() == want.tobytes()
    got = e.correlate_all(np.zeros(6, np.float32))      # (the batch's size classes are analysed now)
    assert got.tobytes() == want.tobytes()
    assert [e.sector_info(s)[0] for s in range(n)] == [ref.sector_info(s)[0] for s in range(n)]
    # Lagrangian move of every sector by its own record: patched in place here; on the reference engine the explicit list
    # moves first, which sends every later update through the host records and a full rebuild before the next solve
    for s in range(n):
        e.update_sector(s, 1)
    for s in [n] + list(range(n)):
        ref.update_sector(s, 1)
    a, b = e.correlate_all(want["p"]), ref.correlate_all(want_all["p"])
    assert a.tobytes() == b[:n].tobytes()
    assert np.abs(a["und_cx"] - want["und_cx"] - np.round(want["p"][:, 0])).max() <= 1.0
    # a committed sector registered anew: full rebuild, its own state starts from zero, the others keep theirs
    e.resetPolygon_rect(2, 210, 210, 216, 216)
    e.commit_sectors()
    c = e.correlate_all(want["p"])
    keep = [s for s in range(n) if s != 2]
    assert c[keep].tobytes() == a[keep].tobytes() and c[2]["und_cx"] == 213.0
    e.close()
    ref.close()


@pytest.mark.gpu
def test_appends_across_growth_steps_without_a_solve_or_a_synchronisation_in_between(speckle512):
    """The append path grows the per-sector buffers geometrically (hipMalloc + copy + hipFree).  Kernels still queued on
    the engine's non-blocking stream - the append kernels of the commits before, an unsynchronised solve - must not be
    lost to such a step: 300 sectors committed one by one with nothing in between, then 100 more behind a solve that
    nobody waited for, against ONE commit of the same rectangles."""
    und, dfm = speckle512

    def engine():
        e = ca.HipCorrelationEngine()
        e.set_batch_invariant(True)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        return e

    rects = [(30 + 23 * (s % 19), 30 + 21 * (s // 19), 30 + 23 * (s % 19) + 12 + s % 5, 30 + 21 * (s // 19) + 14 - s % 3) for s in range(400)]
    ref = engine()
    for s, r in enumerate(rects):
        ref.resetPolygon_rect(s, *r)
    ref.commit_sectors()
    want = ref.correlate_all(np.zeros(6, np.float32))
    e = engine()
    for s, r in enumerate(rects[:300]):                 # crosses several growth steps (64, 192, 448 ...), no solve, no sync
        e.resetPolygon_rect(s, *r)
        e.commit_sectors()
    got300 = e.correlate_all(np.zeros(6, np.float32))
    assert got300.tobytes() == want[:300].tobytes()
    assert [e.sector_info(s) for s in (0, 63, 64, 191, 192, 299)] == [ref.sector_info(s) for s in (0, 63, 64, 191, 192, 299)]
    e.adjust_initial_guess(0, False, np.zeros(6, np.float32), (255.5, 255.5))
    e.correlate_all_device(0, 0)                        # an asynchronous solve into the engine's own record buffer ...
    for s in range(300, 400):                           # ... and appends (with growth steps) right behind it
        e.resetPolygon_rect(s, *rects[s])
        e.commit_sectors()
    e.synchronize()
    # the unsynchronised solve's records survived the growth steps: sequence state of the first 300 sectors = want
    e.adjust_initial_guess(1, False, np.zeros(6, np.float32), (255.5, 255.5))
    g = e.get_guesses()
    assert g[:300].tobytes() == np.ascontiguousarray(want["p"][:300]).tobytes() and not g[300:].any()
    got = e.correlate_all(np.zeros(6, np.float32))
    assert got.tobytes() == want.tobytes()
    e.close()
    ref.close()


@pytest.mark.gpu
def test_big_list_sector_on_a_small_frame_with_a_deep_pyramid():
    """A workgroup-wide lane group on an explicit list walks the software-pipelined sample loop, whose branch-free
    prefetch clamps its 4 x 4 window into the image - on pyramid levels smaller than 4 x 4 there is no such window and
    the plain loop must take over (96 x 96 frame, six levels: 3 x 3 pixels at the top)."""
    und, dfm = ca.speckle.speckle_pair(96, 96, p=(0.4, -0.3, 0.001, 0.0, 0.0, -0.001), seed=3)
    ys, xs = np.mgrid[2:94, 2:94]
    pts = np.stack([xs.T.ravel(), ys.T.ravel()], 1).astype(np.float32)      # 8464 samples: the 256-lane class, x outer / y inner
    rec = {}
    for kind in ("list", "rect"):
        e = ca.HipCorrelationEngine(py_stop=5)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        if kind == "list":
            e.set_sector_points(0, pts, center=(47.5, 47.5))
        else:
            e.resetPolygon_rect(0, 2, 2, 93, 93)
        e.commit_sectors()
        assert e.sector_level_count(0, 5) <= 9
        rec[kind] = e.correlate_all(np.zeros(6, np.float32))[0]
        e.close()
    assert rec["list"]["n_points"] == rec["rect"]["n_points"] == 8464
    assert rec["list"]["error_code"] == rec["rect"]["error_code"]
    if rec["rect"]["error_code"] == 0:
        assert np.abs(rec["list"]["p"] - rec["rect"]["p"])[:2].max() < 5e-3


@pytest.mark.gpu
def test_lists_from_the_host_get_a_row_major_evaluation_copy_too(oracle, monkeypatch):
    """lk_set_sector_points with lists in the reference's order (x outer / y inner): the commit sorts a copy of every
    such list by image row (stable counting sort) for the lane groups of the default mode, at every level - short lists
    on the host, long ones through the device decimation.  Same samples; records differ from the list-order walk by
    summation order only; a list that is row-major already is left alone."""
    und, dfm = ca.speckle.speckle_pair(768, 768, p=(0.9, 0.4, 0.001, 0.0005, -0.0005, 0.0015), seed=21)
    ann = [oracle.annular_points(np.float32(150.0 + 60 * i), np.float32(60.0), np.float32(j) * np.float32(np.pi / 2), np.float32(np.pi / 2),
                                 384.0, 380.0, 4) for i in range(3) for j in range(4)]
    xs, ys = np.meshgrid(np.arange(40, 140), np.arange(50, 120), indexing="xy")          # a rectangle given row by row
    rows_first = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float32)

    def run(flag, n_lists):
        monkeypatch.setenv("LK_EVAL_LISTS", flag)
        e = ca.HipCorrelationEngine(py_stop=2)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        for s, pts in enumerate(ann[:n_lists]):
            e.set_sector_points(s, pts)
        e.set_sector_points(n_lists, rows_first)
        e.commit_sectors()
        lists = None
        if flag == "1":
            lists = [[(e.level_xy(l, k), e.level_xy(l, k, evaluation_copy=True)) for l in range(3)] for k in range(e.n_sectors)]
        r = e.correlate_all(np.zeros(6, np.float32))
        e.close()
        return r, lists

    for n_lists in (2, 12):      # 2: below 32 768 samples in all - levels decimated on the host; 12: on the device
        r1, lists = run("1", n_lists)
        r0, _ = run("0", n_lists)
        for k, per_level in enumerate(lists):
            for l, (canon, ev) in enumerate(per_level):
                assert len(canon) == len(ev) and len(canon) > 0, (k, l)
                if k == n_lists:
                    assert np.array_equal(canon, ev), l                       # row-major already: untouched
                    continue
                order = np.argsort(canon[:, 1], kind="stable")
                assert np.array_equal(canon[order], ev), (k, l)               # stable by row
                assert not np.array_equal(canon, ev), (k, l)
        assert np.array_equal(r1["error_code"], r0["error_code"]) and (r1["error_code"] == 0).all()
        assert np.abs(r1["p"][:, :2] - r0["p"][:, :2]).max() < 5e-4
        assert (np.abs(r1["chi"] - r0["chi"]) / np.abs(r0["chi"])).max() < 3e-3
