"""Multi-GPU driver: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on ROCm; "gloo" on CPU for the tests).

Sectors are independent (own parameters, chi, iteration count; shared read-only images), so
the path shards by sector with no collective inside a solve:
  * rank 0 owns the frame source; every new frame is broadcast (4 MiB at 2048^2, 64 MiB at
    8192^2) and each rank builds its own pyramids from it - cheaper than broadcasting the
    1.33x larger pyramid;
  * rank r correlates the contiguous block [r*S/G, (r+1)*S/G) of the sector index
    (iSector = i*vs + j keeps a rank's image footprint a band of columns);
  * the 48-byte result records are all-gathered.
The engine object only has to provide set_image / set_rect_grid / commit_sectors /
correlate_all, so the tests drive this module with a CPU stand-in under gloo.
"""
import numpy as np

from .workload import shard_range


class ShardedCorrelator:
    def __init__(self, engine, dist=None, device=None):
        self.e = engine
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.device = device
        self.first = 0
        self.count = 0
        self.total = 0

    def set_rect_grid(self, x_begin, y_begin, x_end, y_end, hs, vs):
        """Every rank registers only its own block of the hs*vs sector grid."""
        self.total = hs * vs
        self.first, self.count = shard_range(self.total, self.rank, self.world)
        self.e.set_rect_grid(x_begin, y_begin, x_end, y_end, hs, vs, self.first, self.count)
        self.e.commit_sectors()

    def broadcast_frame(self, slot, pixels):
        """pixels: uint8 (H, W) array on rank 0 (ignored elsewhere, only its shape is used)."""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(pixels, np.uint8))
        if self.device is not None:
            t = t.to(self.device)
        if self.dist is not None and self.world > 1:
            self.dist.broadcast(t, src=0)
        if self.device is not None and hasattr(self.e, "set_image_device"):
            self.e.set_image_device(slot, t.data_ptr(), t.shape[0], t.shape[1])
            self._keep = t  # the engine copies during the call; keep alive until then
        else:
            self.e.set_image(slot, t.cpu().numpy())

    def correlate_all(self, guesses=None):
        """Returns the records of ALL sectors (same on every rank), in sector order."""
        import torch
        local = self.e.correlate_all(guesses)
        if self.dist is None or self.world == 1:
            return local
        rec = local.dtype
        # blocks differ by at most one sector: pad to the largest, gather, trim
        cap = (self.total + self.world - 1) // self.world
        buf = np.zeros((cap, rec.itemsize), np.uint8)
        buf[:len(local)] = local.view(np.uint8).reshape(len(local), rec.itemsize)
        t = torch.from_numpy(buf)
        if self.device is not None:
            t = t.to(self.device)
        out = torch.empty((self.world * cap, rec.itemsize), dtype=torch.uint8, device=t.device)
        self.dist.all_gather_into_tensor(out, t)
        out = out.cpu().numpy().reshape(self.world, cap, rec.itemsize)
        parts = []
        for r in range(self.world):
            _, cnt = shard_range(self.total, r, self.world)
            parts.append(out[r, :cnt].reshape(-1).view(rec))
        return np.concatenate(parts)
