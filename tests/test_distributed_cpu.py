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
