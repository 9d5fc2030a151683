"""include/lk_group.h: the single-process multi-GPU engine.
CPU: the partition rule.  GPU (one MI355X): a 1-device group - RCCL communicator, broadcast and
all-gather included - must equal the plain engine bit for bit; groups of several ranks on the SAME
device (the rehearsal transport: copies instead of RCCL, every other line of code the same) must
equal it too, for grids, mixed sectors, uneven shards and a tracked constant-velocity sequence."""
import ctypes as C

import numpy as np
import pytest

import correlation_amd as ca
from correlation_amd import _ffi
from correlation_amd.workload import shard_range


def test_partition_rule(engine_lib):
    f, c = C.c_int(), C.c_int()
    for S in (1, 2, 7, 8, 9, 63, 64, 65, 10000, 50176, 199809):
        for n in (1, 2, 3, 4, 5, 8):
            if S < n:
                continue
            nxt, sizes = 0, []
            for r in range(n):
                assert engine_lib.lk_group_shard_range(S, r, n, C.byref(f), C.byref(c)) == 0
                assert (f.value, c.value) == shard_range(S, r, n)      # the Python harness uses the same rule
                assert f.value == nxt                                   # contiguous, in rank order
                nxt += c.value
                sizes.append(c.value)
            assert nxt == S and max(sizes) - min(sizes) <= 1 and max(sizes) <= (S + n - 1) // n
    assert engine_lib.lk_group_shard_range(10, 3, 3, C.byref(f), C.byref(c)) == ca.ERROR_BAD_DOMAIN


def test_group_needs_a_device(engine_lib):
    if engine_lib.lk_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(ca.LkError):
        ca.HipCorrelationGroup(1)


def plain_engine(und, dfm, invariant=True, **kw):
    e = ca.HipCorrelationEngine(**kw)
    e.set_batch_invariant(invariant)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    return e


@pytest.mark.gpu
@pytest.mark.parametrize("invariant", [True, False])
def test_one_device_group_equals_the_engine(speckle512, invariant):
    """n = 1 goes through ncclCommInitAll, ncclBroadcast and ncclAllGather like any other size."""
    und, dfm = speckle512
    e = plain_engine(und, dfm, invariant)
    e.set_rect_grid(24.0, 24.0, 487.0, 487.0, 21, 19)
    e.commit_sectors()
    want = e.correlate_all(np.zeros(6, np.float32))
    e.close()
    g = ca.HipCorrelationGroup(1)
    g.for_each_engine("lk_set_batch_invariant", int(invariant))
    g.set_image(ca.IMG_UND, und)
    g.set_image(ca.IMG_DEF, dfm)
    g.set_rect_grid(24.0, 24.0, 487.0, 487.0, 21, 19)
    g.commit_sectors()
    assert g.size == 1 and g.n_sectors == 399 and g.shard(0) == (0, 399)
    got = g.correlate_all(np.zeros(6, np.float32))
    assert got.tobytes() == want.tobytes()
    st = g.stats()
    assert st["sectors"] == 399 and st["point_iterations"] > 399 * 3
    # records left on the device, fetched later
    g.correlate_all(np.zeros(6, np.float32), fetch=False)
    g.synchronize()
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks", [2, 3, 5])
def test_group_of_several_ranks_equals_the_engine(oracle, speckle512, n_ranks):
    und, dfm = speckle512
    # a grid whose size is not a multiple of the rank count, then individually registered sectors of three kinds
    e = plain_engine(und, dfm)
    e.set_rect_grid(24.0, 24.0, 487.0, 487.0, 13, 11)
    e.commit_sectors()
    rng = np.random.default_rng(4)
    guesses = (rng.standard_normal((143, 6)) * [0.3, 0.3, 1e-3, 1e-3, 1e-3, 1e-3]).astype(np.float32)
    want_grid = e.correlate_all(guesses)
    e.clear_sectors()
    pts = oracle.rect_points(300, 310, 330, 345)
    regs = []
    for s in range(17):
        if s % 3 == 0:
            regs.append(("rect", (20 + 25 * s, 30 + 20 * s, 60 + 25 * s, 75 + 20 * s)))
        elif s % 3 == 1:
            regs.append(("annular", (40.0 + s, 30.0, 0.3 * s, 1.1, 256.0, 250.0, 4)))
        else:
            regs.append(("points", (pts + np.float32([s, -s]), (315.0 + s, 327.0 - s) if s % 2 else None)))
    for s, (kind, a) in enumerate(regs):
        if kind == "rect":
            e.resetPolygon_rect(s, *a)
        elif kind == "annular":
            e.resetPolygon_annular(s, *a)
        else:
            e.set_sector_points(s, a[0], center=a[1])
    e.commit_sectors()
    want_mixed = e.correlate_all(np.zeros(6, np.float32))
    e.close()

    g = ca.HipCorrelationGroup([0] * n_ranks)
    g.for_each_engine("lk_set_batch_invariant", 1)
    g.set_image(ca.IMG_UND, und)
    g.set_image(ca.IMG_DEF, dfm)
    g.set_rect_grid(24.0, 24.0, 487.0, 487.0, 13, 11)
    g.commit_sectors()
    assert [g.shard(r) for r in range(n_ranks)] == [shard_range(143, r, n_ranks) for r in range(n_ranks)]
    got = g.correlate_all(guesses)
    assert got.tobytes() == want_grid.tobytes()
    assert g.stats()["sectors"] == 143
    for s, (kind, a) in enumerate(regs):
        if kind == "rect":
            g.set_sector_rect(s, *a)
        elif kind == "annular":
            g.set_sector_annular(s, *a)
        else:
            g.set_sector_points(s, a[0], center=a[1])
    g.commit_sectors()
    assert g.correlate_all(np.zeros(6, np.float32)).tobytes() == want_mixed.tobytes()
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("chain", ["0", "2"])
def test_group_in_reference_order_mode_carries_the_stale_iteration_count_across_shards(oracle, speckle512, chain, monkeypatch):
    """Reference-order mode: a sector whose very first evaluation fails reports the iteration count the sector
    BEFORE it left behind (correlation_class.cpp:413-419, :870).  In a group the sector before the first one of
    a shard lives on another rank: the markers are resolved over the gathered records, in global order."""
    monkeypatch.setenv("LK_REF_STARVED_CHAIN", chain)   # (2: starved levels by the one-lane kernel first - it must leave the markers too)
    und, dfm = speckle512
    bad = oracle.rect_points(0, 0, 20, 20)           # out of the image at evaluation #0 of the coarsest level
    good = [oracle.rect_points(60 + 40 * i, 80 + 30 * i, 90 + 40 * i, 105 + 30 * i) for i in range(9)]
    lists, cens = [], []
    for s in range(9):                                # shards of 3 ranks: [0,3) [3,6) [6,9); failing sectors lead shards 1 and 2
        if s in (3, 4, 6):
            lists.append(bad), cens.append((10.0, 10.0))
        else:
            lists.append(good[s]), cens.append((75.0 + 40 * s, 92.0 + 30 * s))
    o = oracle.Oracle()
    o.set_image(0, und)
    o.set_image(1, dfm)
    want = o.correlate_sectors(lists, centers=np.array(cens, np.float32))
    assert list(want["error_code"][[3, 4, 6]]) == [2, 2, 2] and want["iterations"][3] == want["iterations"][2] > 0
    for n_ranks in (1, 3):
        g = ca.HipCorrelationGroup([0] * n_ranks)
        g.for_each_engine("lk_set_reference_order", 1)
        g.set_image(ca.IMG_UND, und)
        g.set_image(ca.IMG_DEF, dfm)
        for s in range(9):
            g.set_sector_points(s, lists[s], center=cens[s])
        g.commit_sectors()
        got = g.correlate_all(np.zeros(6, np.float32))
        assert got.tobytes() == want.tobytes(), (n_ranks, got["iterations"], want["iterations"])
        # the carry into the NEXT solve is the last sector's count: solve again with the failing sector first
        got2 = g.correlate_all(np.zeros(6, np.float32))
        assert got2.tobytes() == want.tobytes()
        g.close()


@pytest.mark.gpu
def test_a_member_that_fails_alone_fails_the_call_instead_of_hanging_it(speckle512, monkeypatch):
    """A member that fails before the all-gather (a deferred re-commit, an allocation, the guess upload; here the
    LK_GROUP_FAULT hook) must make the CALL fail on every member - nobody enters the collective - and leave the group
    usable: round 2's group returned early on that rank and left the others' collective waiting for ever."""
    und, dfm = speckle512
    g = ca.HipCorrelationGroup([0, 0, 0])
    g.for_each_engine("lk_set_batch_invariant", 1)
    g.set_image(ca.IMG_UND, und)
    g.set_image(ca.IMG_DEF, dfm)
    g.set_rect_grid(24.0, 24.0, 487.0, 487.0, 9, 8)
    g.commit_sectors()
    want = g.correlate_all(np.zeros(6, np.float32))
    for rank in (1, 0, 2):
        monkeypatch.setenv("LK_GROUP_FAULT", str(rank))
        with pytest.raises(ca.LkError) as ei:
            g.correlate_all(np.zeros(6, np.float32))
        assert ei.value.code == ca.ERROR_DEVICE and f"rank {rank}" in str(ei.value)
        monkeypatch.delenv("LK_GROUP_FAULT")
        assert g.correlate_all(np.zeros(6, np.float32)).tobytes() == want.tobytes()      # the group is still usable
    g.close()


@pytest.mark.gpu
def test_group_tracks_a_constant_velocity_sequence():
    """BASELINE config 4's shape: the guess history lives with the engine that owns the sector; per frame
    only the new image goes out and the records come back (und fixed, def <- nxt rotation)."""
    frames = ca.speckle.speckle_sequence(384, 384, 5, velocity=(0.8, -0.4), dilation=1e-4, seed=9)
    gg, centre = np.zeros(6, np.float32), (191.5, 191.5)

    def run(target, is_group):
        out = []
        target.set_image(ca.IMG_UND, frames[0]) if is_group else target.set_undeformed_image(frames[0])
        target.set_image(ca.IMG_DEF, frames[1]) if is_group else target.set_deformed_image(frames[1])
        target.set_rect_grid(24.0, 24.0, 359.0, 359.0, 14, 15)
        target.commit_sectors()
        for k in range(4):
            if k > 0:
                if is_group:
                    target.set_image(ca.IMG_NXT, frames[k + 1])
                    target.rotate_def_from_nxt()
                else:
                    target.set_next_image(frames[k + 1])
                    target.makeDefPyramidFromNxt()
            target.adjust_initial_guess(k, True, gg, centre)
            out.append(target.correlate_all(None))
        return out

    e = ca.HipCorrelationEngine()
    e.set_batch_invariant(True)
    want = run(e, False)
    e.close()
    g = ca.HipCorrelationGroup([0, 0, 0])
    g.for_each_engine("lk_set_batch_invariant", 1)
    got = run(g, True)
    g.close()
    for k in range(4):
        assert got[k].tobytes() == want[k].tobytes(), k
    u = np.array([np.median(r["p"][:, 0]) for r in want])
    assert np.allclose(u, 0.8 * np.arange(1, 5), atol=0.05)     # the sequence really moves 0.8 px per frame


@pytest.mark.gpu
def test_sharded_correlator_device_path_on_one_rank(speckle512):
    """correlation_amd/distributed.py with a torch device: frames and records stay in HBM, torch / RCCL and
    the engine share one HIP stream (the broadcast, the pyramid launch that reads the broadcast tensor,
    the solve and the gather are ordered by that stream)."""
    import torch
    from correlation_amd.distributed import ShardedCorrelator
    und, dfm = speckle512
    e = plain_engine(und, dfm)
    e.set_rect_grid(24.0, 24.0, 487.0, 487.0, 12, 9)
    e.commit_sectors()
    want = e.correlate_all(np.zeros(6, np.float32))
    e.close()
    e = ca.HipCorrelationEngine()
    e.set_batch_invariant(True)
    sc = ShardedCorrelator(e, None, torch.device("cuda", 0))
    for rep in range(3):     # frames replaced while earlier launches may still be queued
        sc.broadcast_frame(ca.IMG_UND, und if rep == 2 else dfm)
        sc.broadcast_frame(ca.IMG_DEF, dfm if rep == 2 else und)
    sc.set_rect_grid(24.0, 24.0, 487.0, 487.0, 12, 9)
    got = sc.correlate_all(np.zeros(6, np.float32))
    assert got.tobytes() == want.tobytes()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["batch_invariant", "reference_order"])
def test_group_windows_equal_one_engine_and_transfers_overlap_the_solve(mode):
    """Three ranks on one device (the copy transport; every other line of code is the multi-GPU path): two
    frame-pipelined windows of a sequence, the frames of window 1 enqueued while window 0 is being solved.
    Records: the bytes of ONE engine solving the pairs one at a time.  Overlap: HIP event times on rank 1's device -
    the transfer of window 1's frames begins after window 0's solve began and ends before it ended."""
    K, n_ranks = 5, 3
    frames = ca.speckle.speckle_sequence(640, 640, 2 * K + 1, velocity=(0.8, -0.4), dilation=2e-4, seed=5)
    c, zero = (319.5, 319.5), np.zeros(6, np.float32)

    def setup(obj, engines):
        for fn in engines:
            fn("lk_set_batch_invariant", 1) if mode == "batch_invariant" else fn("lk_set_reference_order", 1)

    e = ca.HipCorrelationEngine()
    setup(e, [lambda name, v: getattr(e.lib, name)(e._h, v)])
    e.set_undeformed_image(frames[0])
    e.set_rect_grid(24.0, 24.0, 615.0, 615.0, 66, 66)     # 7 x 7-sample sectors: two starved levels
    e.commit_sectors()
    want = []
    for k in range(2 * K):
        e.set_deformed_image(frames[k + 1])
        e.adjust_initial_guess(k, True, zero, c)
        want.append(e.correlate_all(None))
    want = np.stack(want)
    e.close()

    g = ca.HipCorrelationGroup([0] * n_ranks)
    setup(g, [g.for_each_engine])
    g.set_image(ca.IMG_UND, frames[0])
    g.set_rect_grid(24.0, 24.0, 615.0, 615.0, 66, 66)
    g.commit_sectors()
    g.sequence_reserve(2 * K)
    g.sequence_set_frames(0, frames[1:K + 1])
    g.adjust_initial_guess(0, True, zero, c)
    g.correlate_sequence_async(K, first_slot=0)
    g.sequence_set_frames(K, frames[K + 1:2 * K + 1])     # window 1's frames: behind window 0's solve, not behind the host
    got0 = g.wait_sequence()
    probe = g.probe_overlap(1)
    g.adjust_initial_guess(K, True, zero, c)
    g.correlate_sequence_async(K, first_slot=K)
    got1 = g.wait_sequence()
    got = np.concatenate([got0, got1])
    assert got.shape == want.shape
    for k in range(2 * K):
        assert got[k].tobytes() == want[k].tobytes(), f"pair {k}"
    assert probe[2] > 0 and probe[3] > 0, f"transfer {probe[0]:.3f} ms, solve {probe[1]:.3f} ms, begin offset {probe[2]:.3f}, end offset {probe[3]:.3f}"
    # one pair at a time through the group, the next frame travelling behind the running solve (LK_IMG_NXT)
    g.set_image(ca.IMG_DEF, frames[1])
    g.adjust_initial_guess(0, True, zero, c)
    g.correlate_all(None, fetch=False)
    g.set_image(ca.IMG_NXT, frames[2])
    probe = g.probe_overlap(2)
    assert probe[2] > 0, "the next frame's transfer was enqueued behind the solve's begin, without a host wait"
    g.rotate_def_from_nxt()
    g.adjust_initial_guess(1, True, zero, c)
    r1 = g.correlate_all(None)
    assert r1.tobytes() == want[1].tobytes()
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["batch_invariant", "reference_order"])
def test_group_window_records_fetched_behind_the_next_window(mode):
    """lk_group_sequence_records: window w's records are exchanged and downloaded while window w + 1 is being solved
    (wait_sequence without a fetch -> launch the next window -> fetch).  Three windows on three ranks of one device; the
    bytes are ONE engine's pair-by-pair loop, and a fetch without an exchanged window is refused."""
    K, n_ranks, W = 3, 3, 3
    frames = ca.speckle.speckle_sequence(512, 512, W * K + 1, velocity=(0.8, -0.4), dilation=2e-4, seed=11)
    c, zero = (255.5, 255.5), np.zeros(6, np.float32)

    def setup(fn):
        fn("lk_set_batch_invariant", 1) if mode == "batch_invariant" else fn("lk_set_reference_order", 1)

    e = ca.HipCorrelationEngine()
    setup(lambda name, v: getattr(e.lib, name)(e._h, v))
    e.set_undeformed_image(frames[0])
    e.set_rect_grid(24.0, 24.0, 487.0, 487.0, 40, 37)
    e.commit_sectors()
    want = []
    for k in range(W * K):
        e.set_deformed_image(frames[k + 1])
        e.adjust_initial_guess(k, True, zero, c)
        want.append(e.correlate_all(None))
    want = np.stack(want)
    e.close()

    g = ca.HipCorrelationGroup([0] * n_ranks)
    setup(g.for_each_engine)
    g.set_image(ca.IMG_UND, frames[0])
    g.set_rect_grid(24.0, 24.0, 487.0, 487.0, 40, 37)
    g.commit_sectors()
    g.sequence_reserve(2 * K)
    with pytest.raises(ca.LkError):
        g.sequence_records()
    got = []
    g.sequence_set_frames(0, frames[1:K + 1])
    g.adjust_initial_guess(0, True, zero, c)
    g.correlate_sequence_async(K, first_slot=0)
    for w in range(W):
        if w + 1 < W:   # the next window's frames into the other half of the ring, behind the running solve
            g.sequence_set_frames(((w + 1) % 2) * K, frames[(w + 1) * K + 1:(w + 2) * K + 1])
        g.wait_sequence(fetch=False)               # window w is solved everywhere, its exchange is on its way
        if w + 1 < W:
            g.adjust_initial_guess((w + 1) * K, True, zero, c)
            g.correlate_sequence_async(K, first_slot=((w + 1) % 2) * K)
        got.append(g.sequence_records())           # ... and arrives while window w + 1 runs
    got = np.concatenate(got)
    g.close()
    assert got.shape == want.shape
    for k in range(W * K):
        assert got[k].tobytes() == want[k].tobytes(), f"pair {k}"
