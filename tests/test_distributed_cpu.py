"""The N>1 path (shard by sector, broadcast the frame, all-gather the records) exercised
with world_size 2 on the gloo backend.  The compute is a CPU stand-in with the engine's
interface, backed by the oracle (tests may use the oracle); on the GPU box the same
ShardedCorrelator drives HipCorrelationEngine over RCCL (bench.py)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from correlation_amd.workload import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_sector_index():
    for S in (1, 7, 100, 10000, 50176, 199809):
        for G in (1, 2, 3, 4, 8):
            blocks = [shard_range(S, r, G) for r in range(G)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == S
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["LK_ROOT"])
    import torch.distributed as dist
    from correlation_amd import speckle
    from correlation_amd.distributed import ShardedCorrelator
    from oracle import lk_oracle as lo

    class OracleEngine:  # CPU stand-in with the engine's interface
        def __init__(self):
            self.o = lo.Oracle(model=lo.FM_UVUXUYVXVY)
        def set_image(self, slot, px):
            self.o.set_image(slot, px)
        def set_rect_grid(self, x0, y0, x1, y1, hs, vs, first=0, count=-1):
            xd, yd, cen = lo.rect_sector_geometry(x0, y0, x1, y1, hs, vs)
            count = hs * vs - first if count < 0 else count
            self.cen = cen[first:first + count].astype(np.float32)
            self.lists = [lo.rect_points(cx - xd, cy - yd, cx + xd, cy + yd) for cx, cy in cen[first:first + count]]
        def commit_sectors(self):
            pass
        def correlate_all(self, guesses=None):
            return self.o.correlate_sectors(self.lists, centers=self.cen)

    dist.init_process_group("gloo")
    rank = dist.get_rank()
    und, dfm = speckle.speckle_pair(160, 160, p=(0.6, -0.3, 0.001, 0, 0, 0.001), seed=4)
    if rank != 0:
        und[:] = 0; dfm[:] = 0          # only rank 0 has the frames
    sc = ShardedCorrelator(OracleEngine(), dist)
    sc.broadcast_frame(0, und)
    sc.broadcast_frame(1, dfm)
    sc.set_rect_grid(20.0, 20.0, 139.0, 139.0, 3, 3)
    res = sc.correlate_all()
    np.save(os.path.join(os.environ["LK_OUT"], f"res{rank}.npy"), res.view(np.uint8))
    dist.destroy_process_group()
""")


def test_two_rank_gloo_run_equals_single_process(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, LK_ROOT=ROOT, LK_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    from correlation_amd import speckle
    und, dfm = speckle.speckle_pair(160, 160, p=(0.6, -0.3, 0.001, 0, 0, 0.001), seed=4)
    o = oracle.Oracle(model=oracle.FM_UVUXUYVXVY)
    o.set_image(0, und)
    o.set_image(1, dfm)
    xd, yd, cen = oracle.rect_sector_geometry(20.0, 20.0, 139.0, 139.0, 3, 3)
    lists = [oracle.rect_points(cx - xd, cy - yd, cx + xd, cy + yd) for cx, cy in cen]
    want = o.correlate_sectors(lists, centers=cen.astype(np.float32))
    for rank in (0, 1):
        got = np.load(tmp_path / f"res{rank}.npy").view(oracle.RESULT_DTYPE).reshape(-1)
        assert got.tobytes() == want.tobytes(), f"rank {rank}: gathered records differ from the 1-process run"


SEQ_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["LK_ROOT"])
    import torch.distributed as dist
    import correlation_amd as ca
    from correlation_amd import speckle, tracker as tk
    from correlation_amd.distributed import ShardedSequence
    from oracle import lk_oracle as lo

    class OracleEngine:  # CPU stand-in with the engine's interface (sequence subset)
        def __init__(self):
            self.o = lo.Oracle(model=lo.FM_UVUXUYVXVY)
            self.lists, self.cen = [], []
        def set_image(self, slot, px):
            self.o.set_image(slot, px)
        def makeUndPyramidFromDef(self):
            self.o.und_from_def()
        def clear_sectors(self):
            self.lists, self.cen = [], []
        def resetPolygon_rect(self, s, x0, y0, x1, y1):
            assert s == len(self.lists)
            self.lists.append(lo.rect_points(x0, y0, x1, y1))
            self.cen.append(((x0 + x1) * 0.5, (y0 + y1) * 0.5))
        def commit_sectors(self):
            pass
        def translate_sectors(self, offsets, centers=None):
            for s, pts in enumerate(self.lists):
                out = np.empty_like(pts)
                out[:, 0] = np.trunc((np.float32(offsets[s][0]) + pts[:, 0]).astype(np.float32) + np.float32(0.5))
                out[:, 1] = np.trunc((np.float32(offsets[s][1]) + pts[:, 1]).astype(np.float32) + np.float32(0.5))
                self.lists[s] = out
                self.cen[s] = tuple(centers[s])
        def correlate_all(self, guesses=None):
            return self.o.correlate_sectors(self.lists, centers=np.array(self.cen, np.float32), guesses=guesses)

    dist.init_process_group("gloo")
    rank = dist.get_rank()
    frames = speckle.speckle_sequence(192, 192, 4, velocity=(0.9, -0.5), dilation=4e-4, seed=3)
    if rank != 0:
        frames = [np.zeros_like(f) for f in frames]      # only rank 0 has the images
    mode = int(os.environ["LK_SEQ_MODE"])
    t = tk.SequenceTracker(ca.FM_UVUXUYVXVY, tk.DOMAIN_RECT, mode, tk.REF_PREVIOUS if mode == 1 else tk.REF_FIRST,
                           tk.ERRMODE_CONTINUE, [0.5, -0.25, 0, 0, 0, 0])
    t.set_rect_domain(30.0, 30.0, 161.0, 161.0, 95.5, 95.5, 3, 3)
    seq = ShardedSequence(OracleEngine(), t, dist)
    assert seq.run(frames, [f"f{i}" for i in range(4)]) == 3
    open(os.path.join(os.environ["LK_OUT"], f"report{rank}.csv"), "w").write(t.report())
    dist.destroy_process_group()
""")


def test_two_rank_sharded_sequence_equals_the_manager_oracle(tmp_path, oracle, engine_lib):
    """BASELINE config 4's shape on two gloo ranks: every rank keeps the full tracker and its own
    block of sectors; per frame one broadcast of the image and one all-gather of the records.
    Both ranks must end with the report the single-process manager oracle writes."""
    import pytest
    from correlation_amd import speckle
    from oracle import lk_manager_oracle as mo
    script = tmp_path / "seq_worker.py"
    script.write_text(SEQ_WORKER)
    frames = speckle.speckle_sequence(192, 192, 4, velocity=(0.9, -0.5), dilation=4e-4, seed=3)
    for mode, ref, port in ((mo.DEF_EULERIAN, mo.REF_FIRST, "29541"), (mo.DEF_LAGRANGIAN, mo.REF_PREVIOUS, "29542")):
        env = dict(os.environ, LK_ROOT=ROOT, LK_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", LK_SEQ_MODE=str(mode))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
               "--master-addr", "127.0.0.1", "--master-port", port, str(script)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        o = oracle.Oracle(model=oracle.FM_UVUXUYVXVY)
        o.set_image(0, frames[0])
        o.set_image(1, frames[1])
        m = mo.ManagerOracle(o, 3, mo.DOMAIN_RECT, mode, ref, mo.ERRMODE_CONTINUE, [0.5, -0.25, 0, 0, 0, 0])
        m.set_rect_domain(30.0, 30.0, 161.0, 161.0, 95.5, 95.5, 3, 3)
        for k in range(3):
            if k > 0:
                if ref == mo.REF_PREVIOUS:
                    o.und_from_def()
                o.set_image(2, frames[k + 1])
                o.def_from_nxt()
            m.run_frame(k, "f0" if ref == mo.REF_FIRST else f"f{k}", f"f{k + 1}")
        for rank in (0, 1):
            assert open(tmp_path / f"report{rank}.csv").read() == m.report_text(), (mode, rank)


WIN_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["LK_ROOT"])
    import torch.distributed as dist
    from correlation_amd import speckle
    from correlation_amd.distributed import ShardedWindowSequence
    from oracle import lk_oracle as lo

    class OracleWindowEngine:  # CPU stand-in with the engine's WINDOW interface: a window = its pairs one after the other
        def __init__(self):
            self.o = lo.Oracle(model=lo.FM_UVUXUYVXVY)
            self.ring = {}
        def set_image(self, slot, px):
            self.o.set_image(slot, px)
        def set_rect_grid(self, x0, y0, x1, y1, hs, vs, first=0, count=-1):
            xd, yd, cen = lo.rect_sector_geometry(x0, y0, x1, y1, hs, vs)
            count = hs * vs - first if count < 0 else count
            self.cen = cen[first:first + count].astype(np.float32)
            self.lists = [lo.rect_points(cx - xd, cy - yd, cx + xd, cy + yd) for cx, cy in cen[first:first + count]]
            self.res = np.zeros((count, 6), np.float32)
            self.prev = np.zeros((count, 6), np.float32)
            self.guess = np.zeros((count, 6), np.float32)
        def commit_sectors(self):
            pass
        def sequence_reserve(self, n):
            pass
        def sequence_set_frame(self, slot, px):
            self.ring[slot] = np.array(px, copy=True)
        def adjust_initial_guess(self, frame, cv, gg, center):     # manager_class.cpp:2602-2707, per sector
            for s in range(len(self.lists)):
                self.guess[s], self.prev[s] = lo.adjust_initial_guess(lo.FM_UVUXUYVXVY, frame, cv, gg, self.cen[s, 0], self.cen[s, 1],
                                                                      center[0], center[1], self.res[s], self.prev[s])
        def correlate_sequence_async(self, k, first_slot=0, constant_velocity=True, host_records=True, **kw):
            out = []
            for i in range(k):
                if i > 0:
                    self.adjust_initial_guess(1, constant_velocity, np.zeros(6, np.float32), (0.0, 0.0))
                self.o.set_image(1, self.ring[first_slot + i])
                r = self.o.correlate_sectors(self.lists, centers=self.cen, guesses=self.guess)
                self.res = r["p"].copy()
                out.append(r)
            self.out = np.stack(out)
        def wait_sequence(self, host_records=True):
            return self.out

    dist.init_process_group("gloo")
    rank = dist.get_rank()
    frames = speckle.speckle_sequence(176, 176, 6, velocity=(0.7, -0.4), dilation=3e-4, seed=9)
    if rank != 0:
        frames = [np.zeros_like(f) for f in frames]      # only rank 0 has the images
    seq = ShardedWindowSequence(OracleWindowEngine(), dist, window=2)
    seq.set_rect_grid(24.0, 24.0, 151.0, 151.0, 3, 3)
    rec = seq.run(frames, constant_velocity=True, center=(87.5, 87.5))
    np.save(os.path.join(os.environ["LK_OUT"], f"win{rank}.npy"), rec.view(np.uint8))
    dist.destroy_process_group()
""")


def test_two_rank_sharded_window_sequence_equals_the_one_process_loop(tmp_path, oracle):
    """The sharded WINDOW loop (correlation_amd/distributed.py: ShardedWindowSequence - per window one broadcast of its
    frames, every rank's block through the window, one all-gather of [frames][block] records) on two gloo ranks, five pairs in
    windows of two: both ranks end with the records of the pair-by-pair loop over all sectors, constant-velocity guesses
    (manager_class.cpp:2677-2686) included."""
    from correlation_amd import speckle
    script = tmp_path / "win_worker.py"
    script.write_text(WIN_WORKER)
    env = dict(os.environ, LK_ROOT=ROOT, LK_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29551", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    frames = speckle.speckle_sequence(176, 176, 6, velocity=(0.7, -0.4), dilation=3e-4, seed=9)
    o = oracle.Oracle(model=oracle.FM_UVUXUYVXVY)
    o.set_image(0, frames[0])
    xd, yd, cen = oracle.rect_sector_geometry(24.0, 24.0, 151.0, 151.0, 3, 3)
    lists = [oracle.rect_points(cx - xd, cy - yd, cx + xd, cy + yd) for cx, cy in cen]
    res, prev, want = np.zeros((9, 6), np.float32), np.zeros((9, 6), np.float32), []
    for k in range(5):
        o.set_image(1, frames[k + 1])
        g = np.zeros((9, 6), np.float32)
        for s in range(9):
            g[s], prev[s] = oracle.adjust_initial_guess(oracle.FM_UVUXUYVXVY, k, True, np.zeros(6), cen[s, 0], cen[s, 1], 87.5, 87.5, res[s], prev[s])
        rk = o.correlate_sectors(lists, centers=cen.astype(np.float32), guesses=g)
        res = rk["p"].copy()
        want.append(rk)
    want = np.stack(want)
    assert np.abs(want["p"][-1][:, 0] - 3.5).max() < 0.3          # the sequence tracks: 5 x 0.7 px
    for rank in (0, 1):
        got = np.load(tmp_path / f"win{rank}.npy").view(oracle.RESULT_DTYPE).reshape(5, 9)
        assert got.tobytes() == want.tobytes(), f"rank {rank}: gathered window records differ from the 1-process loop"
