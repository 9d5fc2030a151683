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
    """device = a torch device: frames and records stay in HBM.  torch, RCCL and the engine then share
    ONE HIP stream (a torch.cuda.Stream handed to lk_set_stream): RCCL orders its kernels against
    torch's current stream only, so the broadcast, the engine's upload + pyramid launch, the solve and
    the all-gather are ordered by that stream alone - no event hand-off between two streams to forget."""

    def __init__(self, engine, dist=None, device=None):
        self.e = engine
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.device = device
        self.first = 0
        self.count = 0
        self.total = 0
        self.stream = None
        self._frames = {}      # slot -> device tensor the engine's (asynchronous) pyramid launch reads
        self._rec = None
        if device is not None and hasattr(engine, "set_stream"):
            import torch
            self.stream = torch.cuda.Stream(device)
            engine.set_stream(self.stream.cuda_stream)

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
        if self.device is None or self.stream is None:
            if self.dist is not None and self.world > 1:
                self.dist.broadcast(t, src=0)
            self.e.set_image(slot, t.numpy())
            return
        with torch.cuda.stream(self.stream):
            d = t.to(self.device, non_blocking=False)
            if self.dist is not None and self.world > 1:
                self.dist.broadcast(d, src=0)                   # RCCL, on self.stream
            self.e.set_image_device(slot, d.data_ptr(), d.shape[0], d.shape[1])   # same stream: after the broadcast
            d.record_stream(self.stream)
        # the previous tensor of this slot may still be read by a queued pyramid launch: it is
        # released only after that launch, because everything runs on self.stream in order and the
        # caching allocator hands memory back to the stream the tensor was recorded on
        self._frames[slot] = d

    def correlate_all(self, guesses=None):
        """Returns the records of ALL sectors (same on every rank), in sector order."""
        import torch
        rec = np.dtype(self._record_dtype())
        cap = (self.total + self.world - 1) // self.world if self.total else 0
        if self.device is not None and self.stream is not None and hasattr(self.e, "correlate_all_device"):
            # device path: guesses up once, records gathered in HBM, one copy down
            with torch.cuda.stream(self.stream):
                n = self.count or self.e.n_sectors
                cap = max(cap, n)
                g = torch.zeros((n, 6), dtype=torch.float32, device=self.device)
                if guesses is not None:
                    ga = np.asarray(guesses, np.float32)
                    gh = np.zeros((n, 6), np.float32)
                    if ga.ndim == 1:
                        gh[:, :ga.shape[0]] = ga
                    else:
                        gh[:, :ga.shape[1]] = ga
                    g.copy_(torch.from_numpy(gh))
                if self._rec is None or self._rec.shape[0] != cap:
                    self._rec = torch.zeros((cap, rec.itemsize), dtype=torch.uint8, device=self.device)
                self.e.correlate_all_device(g.data_ptr(), self._rec.data_ptr())
                if self.dist is None or self.world == 1:
                    out = self._rec[:n].cpu().numpy()
                    return out.reshape(-1).view(rec)
                gathered = torch.empty((self.world * cap, rec.itemsize), dtype=torch.uint8, device=self.device)
                self.dist.all_gather_into_tensor(gathered, self._rec)      # RCCL, same stream: after the solve
                out = gathered.cpu().numpy().reshape(self.world, cap, rec.itemsize)
        else:
            local = self.e.correlate_all(guesses)
            if self.dist is None or self.world == 1:
                return local
            # blocks differ by at most one sector: pad to the largest, gather, trim
            buf = np.zeros((cap, rec.itemsize), np.uint8)
            buf[:len(local)] = local.view(np.uint8).reshape(len(local), rec.itemsize)
            t = torch.from_numpy(buf)
            gathered = torch.empty((self.world * cap, rec.itemsize), dtype=torch.uint8)
            self.dist.all_gather_into_tensor(gathered, t)
            out = gathered.numpy().reshape(self.world, cap, rec.itemsize)
        parts = []
        for r in range(self.world):
            _, cnt = shard_range(self.total, r, self.world)
            parts.append(out[r, :cnt].reshape(-1).view(rec))
        return np.concatenate(parts)

    @staticmethod
    def _record_dtype():
        from ._ffi import RESULT_DTYPE
        return RESULT_DTYPE


class ShardedSequence:
    """A tracked sequence (include/lk_tracker.h) over several GPUs - BASELINE config 4's shape.

    Every rank keeps the full tracker (host bookkeeping of ALL sectors is cheap and, fed with
    the same all-gathered records, identical everywhere) and an engine that holds only its
    block of sectors, including their guess history and moved sample lists - so a frame costs
    one broadcast of the new image and one all-gather of 48-byte records, nothing else
    (SURVEY.md section 8e).  `engine` needs the HipCorrelationEngine methods used below."""

    def __init__(self, engine, tracker, dist=None, device=None):
        self.sc = ShardedCorrelator(engine, dist, device)
        self.e, self.t = engine, tracker
        self.S = tracker.n_sectors
        self.first, self.count = shard_range(self.S, self.sc.rank, self.sc.world)
        self.sc.total, self.sc.first, self.sc.count = self.S, self.first, self.count

    def _apply(self, frame, cmds):
        from . import tracker as tk
        mine = cmds[self.first:self.first + self.count]
        if frame == 0:
            self.e.clear_sectors()
            for s, c in enumerate(mine):
                kind = int(c["kind"])
                if kind == tk.SECTOR_RECT:
                    self.e.resetPolygon_rect(s, int(c["x0"]), int(c["y0"]), int(c["x1"]), int(c["y1"]))
                elif kind == tk.SECTOR_ANNULAR:
                    self.e.resetPolygon_annular(s, c["r"], c["dr"], c["a"], c["da"], c["cx"], c["cy"], int(c["as"]))
                else:
                    raise ValueError("a blob domain is one sector: nothing to shard")
            self.e.commit_sectors()
        elif len(mine) and int(mine[0]["kind"]) != tk.SECTOR_KEEP:
            centers = np.stack([mine["center_x"], mine["center_y"]], 1) if int(mine[0]["use_center"]) else None
            if int(mine[0]["kind"]) == tk.SECTOR_TRANSLATE:
                self.e.translate_sectors(np.stack([mine["offset_x"], mine["offset_y"]], 1), centers)
            else:
                self.e.rewarp_sectors(centers)

    def run(self, frames, names=None):
        """frames: list of uint8 images (their content matters on rank 0 only).  Returns the
        number of pairs correlated; the report is `tracker.report()` (same on every rank)."""
        from . import tracker as tk
        from ._ffi import IMG_DEF, IMG_UND
        names = names or [f"frame{i}" for i in range(len(frames))]
        previous = self.t.cfg.reference_image == tk.REF_PREVIOUS
        self.sc.broadcast_frame(IMG_UND, frames[0])
        und_name, done = names[0], 0
        for k in range(len(frames) - 1):
            if k > 0 and previous:      # image roles, manager_class.cpp:1386-1407
                self.e.makeUndPyramidFromDef()
                und_name = names[k]
            self.sc.broadcast_frame(IMG_DEF, frames[k + 1])
            cmds, guesses = self.t.begin_frame(k)
            self._apply(k, cmds)
            records = self.sc.correlate_all(guesses[self.first:self.first + self.count])
            first_unsolved, stop = self.t.end_frame(k, und_name, names[k + 1], records)
            if first_unsolved < self.first + self.count and k > 0:
                self.e.restore_sectors(max(0, first_unsolved - self.first))
            done = k + 1
            if stop:
                break
        return done


class ShardedWindowSequence:
    """An Eulerian sequence in frame-pipelined WINDOWS over several GPUs (DESIGN.md sections 3.5, 7): every rank keeps the
    undeformed pyramid (built once) and its block of sectors with their guess history; per window of K pairs ONE
    broadcast of the K new frames from rank 0 - issued before the window it overlaps is launched, into the other half of
    a double buffer - every rank solves the window for its block (lk_correlate_sequence_async), ONE all-gather of
    K x block records.  `engine` needs sequence_reserve / sequence_set_frame[_device] / adjust_initial_guess /
    correlate_sequence_async / wait_sequence (+ copy_sequence_records_device on the device path); the CPU tests drive
    it with a stand-in under gloo."""

    def __init__(self, engine, dist=None, device=None, window=16):
        self.e, self.dist, self.device = engine, dist, device
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.K = int(window)
        self.total = self.first = self.count = 0
        self.comm = None
        if device is not None:
            import torch
            self.comm = torch.cuda.Stream(device)

    def set_rect_grid(self, x_begin, y_begin, x_end, y_end, hs, vs):
        self.total = hs * vs
        self.first, self.count = shard_range(self.total, self.rank, self.world)
        self.e.set_rect_grid(x_begin, y_begin, x_end, y_end, hs, vs, self.first, self.count)
        self.e.commit_sectors()

    def run(self, frames, constant_velocity=True, global_guess=None, center=(0.0, 0.0), fetch=True, sharded=True):
        """frames: [n + 1, H, W] uint8 - a numpy array / list (host path) or a torch tensor on `device`; their content matters
        on rank 0 only (sharded) - frame 0 is the undeformed image.  Returns the records [n][S] of ALL sectors in sector
        order (fetch; same on every rank), or None (the last window's gathered records stay on the device)."""
        import torch
        from ._ffi import IMG_UND, RESULT_DTYPE
        n = len(frames) - 1
        K = min(self.K, n)
        n_win = (n + K - 1) // K
        rec_bytes = RESULT_DTYPE.itemsize
        cap = (self.total + self.world - 1) // self.world
        use_dist = self.dist is not None and self.world > 1 and sharded
        gg = np.zeros(6, np.float32) if global_guess is None else np.asarray(global_guess, np.float32)
        dev = self.device
        H, W = int(frames[0].shape[0]), int(frames[0].shape[1])
        self.e.sequence_reserve(2 * K)
        out = np.zeros((n, self.total), RESULT_DTYPE) if fetch else None
        if dev is None:   # ---- host path (numpy frames, CPU collectives: the gloo tests)
            und = torch.from_numpy(np.ascontiguousarray(frames[0], np.uint8).copy())
            if use_dist:
                self.dist.broadcast(und, src=0)
            self.e.set_image(IMG_UND, und.numpy())
            for w in range(n_win):
                k = min(K, n - w * K)
                stage = torch.from_numpy(np.ascontiguousarray(np.stack([frames[1 + w * K + i] for i in range(k)]), np.uint8).copy())
                if use_dist:
                    self.dist.broadcast(stage, src=0)
                for i in range(k):
                    self.e.sequence_set_frame((w % 2) * K + i, stage[i].numpy())
                self.e.adjust_initial_guess(w * K, constant_velocity, gg, center)
                self.e.correlate_sequence_async(k, first_slot=(w % 2) * K, constant_velocity=constant_velocity, host_records=True)
                local = self.e.wait_sequence(True)                                   # [k][count]
                block = np.zeros((K, cap, rec_bytes), np.uint8)
                block[:k, :self.count] = local.view(np.uint8).reshape(k, self.count, rec_bytes)
                if use_dist:
                    gathered = torch.empty((self.world, K, cap, rec_bytes), dtype=torch.uint8)
                    self.dist.all_gather_into_tensor(gathered.view(-1, rec_bytes), torch.from_numpy(block).view(-1, rec_bytes))
                    allb = gathered.numpy()
                else:
                    allb = block[None]
                if fetch:
                    self._scatter(out, allb, w * K, k, use_dist)
            return out
        # ---- device path: frames and records stay in HBM; collectives on a communication stream of their own
        stage = [torch.empty((K, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
        d_block = torch.zeros((K, cap, rec_bytes), dtype=torch.uint8, device=dev)
        d_all = [torch.empty((self.world, K, cap, rec_bytes), dtype=torch.uint8, device=dev) for _ in range(2)] if use_dist else None

        def fetch_window(w, buf):
            lo, k = 1 + w * K, min(K, n - w * K)
            with torch.cuda.stream(self.comm):
                if self.rank == 0 or not use_dist:
                    stage[buf][:k].copy_(frames[lo:lo + k], non_blocking=True)
                return self.dist.broadcast(stage[buf], src=0, async_op=True) if use_dist else None

        self.e.set_image_device(IMG_UND, frames[0].data_ptr(), H, W)               # the undeformed pyramid: once per sequence
        h = fetch_window(0, 0)
        gather = None
        pending = None                                                              # (window, frames) whose gathered records are to be copied out
        for w in range(n_win):
            k = min(K, n - w * K)
            if h is not None:
                h.wait()
            self.comm.synchronize()                                                  # the frames of window w have arrived
            for i in range(k):
                self.e.sequence_set_frame_device((w % 2) * K + i, stage[w % 2][i].data_ptr(), H, W)
            if w + 1 < n_win:
                h = fetch_window(w + 1, (w + 1) % 2)                                 # travels while window w is solved
            self.e.adjust_initial_guess(w * K, constant_velocity, gg, center)
            self.e.correlate_sequence_async(k, first_slot=(w % 2) * K, constant_velocity=constant_velocity, host_records=False)
            self.e.wait_sequence(False)
            if gather is not None:
                gather.wait()                                                        # (the previous all-gather has read d_block)
            if fetch and pending is not None:
                self._scatter(out, d_all[pending[0] % 2].cpu().numpy(), pending[0] * K, pending[1], True)
                pending = None
            self.e.copy_sequence_records_device(d_block.data_ptr(), cap)             # [k][count] -> the padded block [K][cap]
            self.e.synchronize()
            if use_dist:
                with torch.cuda.stream(self.comm):
                    gather = self.dist.all_gather_into_tensor(d_all[w % 2].view(-1, rec_bytes), d_block.view(-1, rec_bytes), async_op=True)
                pending = (w, k)
            elif fetch:
                self._scatter(out, d_block.cpu().numpy()[None], w * K, k, False)
        if gather is not None:
            gather.wait()
        self.comm.synchronize()
        if fetch and pending is not None:
            self._scatter(out, d_all[pending[0] % 2].cpu().numpy(), pending[0] * K, pending[1], True)
        self.e.synchronize()
        return out

    def _scatter(self, out, blocks, first_pair, k, gathered):
        """blocks [ranks][K][cap][48] -> out[first_pair : first_pair + k] in global sector order"""
        from ._ffi import RESULT_DTYPE
        for r in range(blocks.shape[0]):
            f, c = shard_range(self.total, r, self.world) if gathered else (0, self.count)
            rec = np.ascontiguousarray(blocks[r, :k, :c]).reshape(-1).view(RESULT_DTYPE).reshape(k, c)
            out[first_pair:first_pair + k, f:f + c] = rec
