"""Frame-pipelined windows (lk_correlate_sequence_async): every sector advances through the frames of a sequence on its
own - (frame, sector) tickets drawn frame-major, a per-sector chain of granules from frame to frame - instead of one
launch per pair (perform_multiframe_correlation's frame loop, manager_class.cpp:1380-1496; the guess rule
:2677-2699).  The records of every frame must be the BYTES of the one-pair-at-a-time loop
(lk_adjust_initial_guess + lk_correlate_all) in reference-order and in batch-invariant mode, on sectors with and
without starved pyramid levels, on several size classes at once, in one window or in several."""
import os

import numpy as np
import pytest

import correlation_amd as ca
from correlation_amd.workload import C2, C4

pytestmark = pytest.mark.gpu

ZERO = np.zeros(6, np.float32)


def make_engine(mode, model=ca.FM_UVUXUYVXVY, interpolation=ca.IM_BICUBIC):
    e = ca.HipCorrelationEngine(fitting_model=model, interpolation=interpolation)
    if mode == "batch_invariant":
        e.set_batch_invariant(True)
    elif mode.startswith("reference_order"):
        e.set_reference_order(int(mode.split(":")[1]) if ":" in mode else 1)
    return e


def domain(e, kind, size):
    lo, hi = 24.0, float(size - 25)
    if kind == "c2like":      # 19 x 19 samples: 361 / 90 / 25 per level
        n = int((hi - lo) // 19.7)
        e.set_rect_grid(lo, lo, hi, hi, n, n)
    elif kind == "c4like":    # 7 x 7 samples: two starved levels (9-16 and 1-4 samples)
        n = int((hi - lo) // 8.95)
        e.set_rect_grid(lo, lo, hi, hi, n, n)
    else:                     # three size classes side by side: 7 x 7 and 19 x 19 (16-lane rows), 41 x 41 (32 lanes), 81 x 81 (64)
        s = 0
        for y in range(40, size - 140, 23):
            for x in range(40, size - 140, 23):
                half = (3, 9, 3, 20, 9, 3, 40)[s % 7] if (x + y) % 5 else 3
                e.resetPolygon_rect(s, x, y, x + 2 * half, y + 2 * half)
                s += 1
    e.commit_sectors()


def loop(e, frames, first, n, velocity=True, center=None):
    """pairs first .. first+n-1 one at a time: (guesses [n][S][6], records [n][S])"""
    rec, gue = [], []
    for k in range(first, first + n):
        e.set_deformed_image(frames[k + 1])
        e.adjust_initial_guess(k, velocity, ZERO, center)
        gue.append(e.get_guesses())
        rec.append(e.correlate_all(None))
    return np.stack(gue), np.stack(rec)


def window(e, frames, first, n, velocity=True, center=None, slot0=0):
    for i in range(n):
        e.sequence_set_frame((slot0 + i), frames[first + i + 1])
    e.adjust_initial_guess(first, velocity, ZERO, center)
    rec = e.correlate_sequence(n, first_slot=slot0, constant_velocity=velocity, keep_guesses=True)
    return e.sequence_guesses(), rec


@pytest.fixture(scope="module")
def frames448():
    return ca.speckle.speckle_sequence(448, 448, 8, velocity=(0.8, -0.4), dilation=2e-4, seed=5)


@pytest.mark.parametrize("kind", ["c2like", "c4like", "mixed"])
@pytest.mark.parametrize("mode", ["reference_order:1", "reference_order:20", "batch_invariant"])
def test_window_records_are_the_one_pair_loops_bytes(frames448, mode, kind):
    frames, n, c = frames448, 7, (223.5, 223.5)
    a, b = make_engine(mode), make_engine(mode)
    for e in (a, b):
        e.set_undeformed_image(frames[0])
        domain(e, kind, 448)
    g_loop, r_loop = loop(a, frames, 0, n, center=c)
    b.sequence_reserve(n)
    g_win, r_win = window(b, frames, 0, n, center=c)
    assert b.sequence_is_pipelined, "every class of these domains has a frame-pipelined instance"
    assert (r_loop["error_code"] == 0).mean() > 0.9
    assert abs(np.median(r_loop["p"][-1][r_loop["error_code"][-1] == 0][:, 0]) - 5.6) < 0.3   # the sequence tracks: 7 x 0.8 px
    for f in range(n):
        assert g_win[f].tobytes() == g_loop[f].tobytes(), f"guesses of frame {f}"
        assert r_win[f].tobytes() == r_loop[f].tobytes(), f"records of frame {f}"
    st_a, st_b = a.stats(), b.stats()
    assert st_b["sectors"] == n * b.n_sectors and st_b["point_iterations"] > 6 * st_a["point_iterations"]
    a.close()
    b.close()


@pytest.mark.parametrize("pyramid", [(1, 1, 2), (0, 2, 2), (0, 1, 0), (0, 1, 3)], ids=lambda t: "levels_%d_%d_%d" % t)
@pytest.mark.parametrize("interpolation", [ca.IM_BICUBIC, ca.IM_BILINEAR, ca.IM_NEAREST], ids=["bicubic", "bilinear", "nearest"])
def test_window_bytes_on_other_pyramids_and_interpolators(frames448, pyramid, interpolation):
    """levels that stop above the image, skip a level, are the image alone, or go one level deeper; the two cheaper samplers:
    the window's frames still are the loop's bytes (batch-invariant mode; the mixed domain: every group width at once)"""
    if interpolation != ca.IM_BICUBIC and pyramid != (0, 1, 2) and pyramid != (1, 1, 2):
        pytest.skip("the samplers are crossed with one other pyramid only")
    frames, n, c = frames448, 5, (223.5, 223.5)
    engines = []
    for _ in range(2):
        e = ca.HipCorrelationEngine(interpolation=interpolation, py_start=pyramid[0], py_step=pyramid[1], py_stop=pyramid[2])
        e.set_batch_invariant(True)
        e.set_undeformed_image(frames[0])
        domain(e, "mixed", 448)
        engines.append(e)
    a, b = engines
    g_loop, r_loop = loop(a, frames, 0, n, center=c)
    b.sequence_reserve(n)
    g_win, r_win = window(b, frames, 0, n, center=c)
    assert (r_loop["error_code"] == 0).mean() > 0.5
    assert g_win.tobytes() == g_loop.tobytes() and r_win.tobytes() == r_loop.tobytes()
    # ... and the state the window leaves is the loop's: one more pair through the one-pair path on both
    g_a, r_a = loop(a, frames, n, 1, center=c)
    g_b, r_b = loop(b, frames, n, 1, center=c)
    assert g_a.tobytes() == g_b.tobytes() and r_a.tobytes() == r_b.tobytes()
    a.close()
    b.close()


@pytest.mark.parametrize("mode", ["reference_order:1", "batch_invariant"])
def test_windows_chain_like_the_loop(frames448, mode):
    """3 + 1 + 3 frames in three windows (ring slots reused), then a one-pair solve: the sequence state a window leaves
    (previous_resulting_parameters, last parameters) is the loop's"""
    frames, c = frames448, (223.5, 223.5)
    a, b = make_engine(mode), make_engine(mode)
    for e in (a, b):
        e.set_undeformed_image(frames[0])
        domain(e, "c4like", 448)
    g_loop, r_loop = loop(a, frames, 0, 7, center=c)
    b.sequence_reserve(4)
    got_g, got_r = [], []
    for first, n, slot0 in ((0, 3, 0), (3, 1, 3), (4, 2, 0)):
        g, r = window(b, frames, first, n, center=c, slot0=slot0)
        got_g.append(g)
        got_r.append(r)
    g, r = loop(b, frames, 6, 1, center=c)   # the last pair through the one-pair path
    got_g.append(g)
    got_r.append(r)
    got_g, got_r = np.concatenate(got_g), np.concatenate(got_r)
    assert got_g.tobytes() == g_loop.tobytes()
    assert got_r.tobytes() == r_loop.tobytes()
    a.close()
    b.close()


@pytest.mark.parametrize("model", [ca.FM_U, ca.FM_UV, ca.FM_UVQ])
def test_window_other_models_and_previous_image_reference(frames448, model):
    """the other warp models; reference = the previous image (und of frame i = def of frame i - 1, guess = p(f-1))"""
    frames, n = frames448, 5
    a, b = make_engine("batch_invariant", model), make_engine("batch_invariant", model)
    for e in (a, b):
        domain_kind = "c2like"
        e.set_undeformed_image(frames[0])
        domain(e, domain_kind, 448)
    rec = []
    for k in range(n):
        a.set_undeformed_image(frames[k])
        a.set_deformed_image(frames[k + 1])
        a.adjust_initial_guess(k, False, ZERO, (223.5, 223.5))
        rec.append(a.correlate_all(None))
    b.sequence_reserve(n)
    for i in range(n):
        b.sequence_set_frame(i, frames[i + 1])
    b.adjust_initial_guess(0, False, ZERO, (223.5, 223.5))
    got = b.correlate_sequence(n, reference_previous=True, constant_velocity=False)
    assert got.tobytes() == np.stack(rec).tobytes()
    assert np.abs(np.median(got["p"][:, :, 0], axis=1) - 0.8).max() < 0.1   # frame-to-frame displacement
    a.close()
    b.close()


def test_default_mode_window_stays_with_the_loop(frames448):
    """default mode: fixed lane groups and the fast flavour inside the window - not the loop's bytes (the loop widens
    groups by batch composition), the loop's results"""
    frames, n, c = frames448, 6, (223.5, 223.5)
    a, b = make_engine("default"), make_engine("default")
    for e in (a, b):
        e.set_undeformed_image(frames[0])
        domain(e, "c2like", 448)
    _, r_loop = loop(a, frames, 0, n, center=c)
    b.sequence_reserve(n)
    _, r_win = window(b, frames, 0, n, center=c)
    assert b.sequence_is_pipelined
    assert np.array_equal(r_win["error_code"], r_loop["error_code"])
    ok = r_loop["error_code"] == 0
    assert np.abs(r_win["p"] - r_loop["p"])[ok][:, :2].max() < 2e-3
    assert (np.abs(r_win["iterations"] - r_loop["iterations"])[ok] <= 1).mean() > 0.99
    a.close()
    b.close()


def test_small_blocks_of_small_sectors_keep_their_pipelined_instance(frames448):
    """One rank's block of a sharded sequence: a few thousand 9 x 9-sample sectors.  The one-pair classifier promotes such a
    class to wider lane groups (too few wavefronts to fill the chip); inside a window it must still run on the 16-lane rows
    that solve its starved level - it used to fall back to frame-after-frame launches (1.0 against 0.54 ms per pair on a
    quarter of the 9 x 9 grid).  Default mode: the loop's results within the mode's tolerance."""
    frames, n, c = frames448, 6, (223.5, 223.5)
    a, b = make_engine("default"), make_engine("default")
    for e in (a, b):
        e.set_undeformed_image(frames[0])
        e.set_rect_grid(24.0, 24.0, 423.0, 423.0, 40, 40)   # 40 x 40 sectors of 9 x 9 samples (pitch 9.98): levels of 81 / 25 / 9
        e.commit_sectors()
    assert a.sector_info(0)[0] == 81
    _, r_loop = loop(a, frames, 0, n, center=c)
    b.sequence_reserve(n)
    _, r_win = window(b, frames, 0, n, center=c)
    assert b.sequence_is_pipelined
    same = r_win["error_code"] == r_loop["error_code"]
    assert same.mean() > 0.995
    ok = same & (r_loop["error_code"] == 0)
    d = np.abs(r_win["p"] - r_loop["p"])[ok][:, :2].max(1)
    assert np.percentile(d, 99) < 0.08 and np.median(d) < 1e-4, (np.percentile(d, 99), np.median(d))   # (the bound of the one-pair default-mode tests on such sectors)
    a.close()
    b.close()


def test_domains_without_a_pipelined_instance_run_frame_after_frame(frames448):
    """a sector of more than 8192 samples (workgroup-wide lane group): same interface, the one-pair launches underneath"""
    frames, n, c = frames448, 4, (223.5, 223.5)
    a, b = make_engine("batch_invariant"), make_engine("batch_invariant")
    for e in (a, b):
        e.set_undeformed_image(frames[0])
        e.resetPolygon_rect(0, 60, 60, 60 + 120, 60 + 120)    # 121 x 121 = 14 641 samples
        e.resetPolygon_rect(1, 200, 200, 218, 218)
        e.commit_sectors()
    g_loop, r_loop = loop(a, frames, 0, n, center=c)
    b.sequence_reserve(n)
    g_win, r_win = window(b, frames, 0, n, center=c)
    assert not b.sequence_is_pipelined
    assert g_win.tobytes() == g_loop.tobytes() and r_win.tobytes() == r_loop.tobytes()
    g2, r2 = loop(b, frames, 4, 1, center=c)
    g1, r1 = loop(a, frames, 4, 1, center=c)
    assert g2.tobytes() == g1.tobytes() and r2.tobytes() == r1.tobytes()
    a.close()
    b.close()


@pytest.mark.parametrize("wl", [C2, C4], ids=["C2", "C4"])
@pytest.mark.parametrize("mode", ["reference_order:1", "batch_invariant"])
def test_64_frame_sequences_at_full_size(wl, mode):
    """BASELINE configs 2 and 4 as 64-pair sequences at full size (2048^2, 10 000 sectors of 19 x 19 / 50 176 of 7 x 7,
    constant-velocity guesses): every record of every frame equals the one-pair loop's, byte for byte"""
    n = int(os.environ.get("LK_TEST_SEQ_FRAMES", 64))
    frames = full_size_frames(n + 1)
    c = (1023.5, 1023.5)
    a, b = make_engine(mode), make_engine(mode)
    for e in (a, b):
        e.set_undeformed_image(frames[0])
        e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
        e.commit_sectors()
    b.sequence_reserve(n)
    for i in range(n):
        b.sequence_set_frame(i, frames[i + 1])
    b.adjust_initial_guess(0, True, ZERO, c)
    b.correlate_sequence_async(n, constant_velocity=True)
    got = b.wait_sequence()
    assert b.sequence_is_pipelined
    for k in range(n):
        a.set_deformed_image(frames[k + 1])
        a.adjust_initial_guess(k, True, ZERO, c)
        want = a.correlate_all(None)
        assert got[k].tobytes() == want.tobytes(), f"frame {k}"
    # (the image content moves by 0.8 n px: the sectors within that distance of the right edge lose it - out of image; config
    # 4's 7 x 7-sample sectors also run into max_iters on their singular levels, in the reference as here)
    ok = got["error_code"][-1] == 0
    assert (got["error_code"][0] == 0).mean() > 0.99 and ok.mean() > (0.9 if wl is C2 else 0.7)
    u = 0.8 * n + 1e-4 * n * (got["und_cx"][-1] - 1024.0)
    assert np.median(np.abs(got["p"][-1][:, 0] - u)[ok]) < 0.05
    a.close()
    b.close()


_FULL = {}


def full_size_frames(n):
    if len(_FULL.get("frames", ())) < n:
        _FULL["frames"] = ca.speckle.speckle_sequence(2048, 2048, n, velocity=(0.8, -0.4), dilation=1e-4, seed=7, device="cuda")
    return _FULL["frames"]


def test_frames_from_pinned_host_memory_upload_without_waiting(frames448):
    """lk_pin_host_memory: image slots, the next-image slot and ring slots filled from page-locked frames (the call
    returns when the copy is enqueued) give the records of the same frames from pageable memory"""
    import ctypes as C
    frames = frames448
    pinned = [np.ascontiguousarray(f).copy() for f in frames[:4]]
    a, b = make_engine("batch_invariant"), make_engine("batch_invariant")
    for f in pinned:
        assert b.lib.lk_pin_host_memory(C.c_void_p(f.ctypes.data), C.c_size_t(f.nbytes)) == 0
    for e, src in ((a, frames), (b, pinned)):
        e.set_undeformed_image(src[0])
        e.set_deformed_image(src[1])
        domain(e, "c2like", 448)
    want1, got1 = a.correlate_all(ZERO), b.correlate_all(ZERO)
    assert got1.tobytes() == want1.tobytes()
    for e, src in ((a, frames), (b, pinned)):
        e.set_next_image(src[2])
        e.makeDefPyramidFromNxt()
    want2, got2 = a.correlate_all(ZERO), b.correlate_all(ZERO)
    assert got2.tobytes() == want2.tobytes() and got2.tobytes() != got1.tobytes()
    for e, src in ((a, frames), (b, pinned)):
        e.sequence_reserve(3)
        for i in range(3):
            e.sequence_set_frame(i, src[i + 1])
        e.adjust_initial_guess(0, True, ZERO, (223.5, 223.5))
    assert a.correlate_sequence(3).tobytes() == b.correlate_sequence(3).tobytes()
    b.synchronize()
    for f in pinned:
        assert b.lib.lk_unpin_host_memory(C.c_void_p(f.ctypes.data)) == 0
    a.close()
    b.close()


def test_sharded_window_sequence_device_path_on_one_rank(frames448):
    """correlation_amd/distributed.py: ShardedWindowSequence with frames and records in HBM (the bench's sharded sequences;
    its collectives are covered by the 2-rank gloo test on the CPU): windows of 3 through the class = one window of 7"""
    import torch
    from correlation_amd.distributed import ShardedWindowSequence
    frames = np.stack(frames448)
    d = torch.from_numpy(frames).to("cuda")
    e = make_engine("batch_invariant")
    sq = ShardedWindowSequence(e, None, torch.device("cuda", 0), window=3)
    n = int((448 - 49) // 8.95)
    sq.set_rect_grid(24.0, 24.0, 448.0 - 25.0, 448.0 - 25.0, n, n)
    got = sq.run(d, center=(223.5, 223.5))
    a = make_engine("batch_invariant")
    a.set_undeformed_image(frames[0])
    domain(a, "c4like", 448)
    a.sequence_reserve(7)
    _, want = window(a, frames, 0, 7, center=(223.5, 223.5))
    assert got.shape == want.shape and got.tobytes() == want.tobytes()
    e.close()
    a.close()


def test_two_windows_at_once_need_no_co_residency():
    """Two engines launch a window each, at the same time, each grid sized for an empty GPU (config 2's 10 000 sectors: 3200
    wavefronts per window for 4096 slots): the second kernel is only partly resident while the first one runs.  A group waits
    only for tickets that were drawn earlier - by wavefronts that are running - so both windows finish, with the bytes of their
    solo runs."""
    n = 6
    frames = full_size_frames(2 * n + 1)
    c = (1023.5, 1023.5)

    def engine(first_frame):
        e = make_engine("batch_invariant")
        e.set_undeformed_image(frames[0])
        e.set_rect_grid(C2.x_begin, C2.x_begin, C2.x_end, C2.x_end, C2.hs, C2.vs)
        e.commit_sectors()
        e.sequence_reserve(n)
        for i in range(n):
            e.sequence_set_frame(i, frames[first_frame + i + 1])
        e.adjust_initial_guess(0, False, ZERO, c)
        return e

    a, b = engine(0), engine(n)
    solo_a = a.correlate_sequence(n, constant_velocity=False)
    solo_b = b.correlate_sequence(n, constant_velocity=False)
    a.adjust_initial_guess(0, False, ZERO, c)
    b.adjust_initial_guess(0, False, ZERO, c)
    a.correlate_sequence_async(n, constant_velocity=False)
    b.correlate_sequence_async(n, constant_velocity=False)
    both_b = b.wait_sequence()
    both_a = a.wait_sequence()
    assert both_a.tobytes() == solo_a.tobytes() and both_b.tobytes() == solo_b.tobytes()
    assert (solo_b["error_code"][0] == 0).mean() > 0.9
    a.close()
    b.close()


def test_a_frame_that_never_publishes_voids_the_window_instead_of_hanging_it(frames448, monkeypatch):
    """The wait of a (frame, sector) for its sector's previous frame is bounded.  Test hook LK_SEQ_FAULT = f + 1: frame f of the
    first sector never publishes its parameters (what a lost wavefront would look like); the group that holds that sector's next
    frame gives up after ~1 s and raises the window's flag, every other waiter sees it and leaves, the grid drains, and
    lk_wait_sequence reports LK_ERROR_DEVICE.  The engine stays usable: the same window solved again (a sequence restarts from its
    first frame after a void window) gives the loop's records."""
    frames, n, c = frames448, 5, (223.5, 223.5)
    e = make_engine("batch_invariant")
    e.set_undeformed_image(frames[0])
    domain(e, "c2like", 448)
    e.sequence_reserve(n)
    for i in range(n):
        e.sequence_set_frame(i, frames[i + 1])
    e.adjust_initial_guess(0, True, ZERO, c)
    monkeypatch.setenv("LK_SEQ_FAULT", "2")          # frame 1 of sector 0 is lost
    e.correlate_sequence_async(n)
    with pytest.raises(ca.LkError) as err:
        e.wait_sequence()
    assert err.value.code == ca.ERROR_DEVICE and "bound" in str(err.value)
    monkeypatch.delenv("LK_SEQ_FAULT")
    e.adjust_initial_guess(0, True, ZERO, c)
    got = e.correlate_sequence(n)
    a = make_engine("batch_invariant")
    a.set_undeformed_image(frames[0])
    domain(a, "c2like", 448)
    _, want = loop(a, frames, 0, n, center=c)
    assert got.tobytes() == want.tobytes()
    a.close()
    e.close()
