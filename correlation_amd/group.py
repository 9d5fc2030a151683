"""ctypes host of include/lk_group.h: the single-process multi-GPU engine (one lk_engine, host
thread and HIP stream per device; RCCL broadcast of frames, all-gather of records)."""
import ctypes as C

import numpy as np

from . import _ffi
from .engine import LkError


class HipCorrelationGroup:
    def __init__(self, devices, interpolation=_ffi.IM_BICUBIC, fitting_model=_ffi.FM_UVUXUYVXVY, precision=1e-3,
                 max_iters=50, py_start=0, py_step=1, py_stop=2, lib=None):
        self.lib = lib or _ffi.load_library()
        devs = list(range(devices)) if isinstance(devices, int) else list(devices)
        cfg = _ffi.LkConfig(interpolation, fitting_model, precision, max_iters, py_start, py_step, py_stop, 0)
        arr = (C.c_int * len(devs))(*devs)
        self._h = C.c_void_p()
        rc = self.lib.lk_group_create(C.byref(cfg), len(devs), arr, C.byref(self._h))
        if rc:
            raise LkError(rc, "lk_group_create failed (a HIP device per entry is required)")
        self.n_params = _ffi.N_PARAMS[fitting_model]

    def _chk(self, rc):
        if rc:
            raise LkError(rc, (self.lib.lk_group_last_error_string(self._h) or b"").decode())

    def close(self):
        if self._h:
            self.lib.lk_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return self.lib.lk_group_size(self._h)

    @property
    def comm_ranks(self):
        """ranks of the RCCL communicator as ncclCommCount reports them (0: the one-GPU rehearsal transport)"""
        return self.lib.lk_group_comm_ranks(self._h)

    def engine_handle(self, rank):
        h = C.c_void_p()
        self._chk(self.lib.lk_group_engine(self._h, rank, C.byref(h)))
        return h

    def for_each_engine(self, fn_name, *args):
        """e.g. for_each_engine("lk_set_batch_invariant", 1)"""
        for r in range(self.size):
            rc = getattr(self.lib, fn_name)(self.engine_handle(r), *args)
            if rc:
                raise LkError(rc, f"{fn_name} failed on rank {r}")

    def shard(self, rank):
        f, c = C.c_int(), C.c_int()
        self._chk(self.lib.lk_group_shard(self._h, rank, C.byref(f), C.byref(c)))
        return f.value, c.value

    def set_image(self, slot, pixels):
        a = np.ascontiguousarray(pixels, np.uint8)
        self._chk(self.lib.lk_group_set_image(self._h, slot, a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1],
                                              a.strides[0]))

    def set_image_device(self, slot, dev_ptr, rows, cols, step=None):
        self._chk(self.lib.lk_group_set_image_device(self._h, slot, C.c_void_p(dev_ptr), rows, cols, step or cols))

    def rotate_und_from_def(self):
        self._chk(self.lib.lk_group_rotate_und_from_def(self._h))

    def rotate_def_from_nxt(self):
        self._chk(self.lib.lk_group_rotate_def_from_nxt(self._h))

    def clear_sectors(self):
        self._chk(self.lib.lk_group_clear_sectors(self._h))

    def set_sector_rect(self, s, x0, y0, x1, y1):
        self._chk(self.lib.lk_group_set_sector_rect(self._h, s, x0, y0, x1, y1))

    def set_rect_grid(self, x_begin, y_begin, x_end, y_end, hs, vs):
        self._chk(self.lib.lk_group_set_rect_grid(self._h, x_begin, y_begin, x_end, y_end, hs, vs))

    def set_sector_annular(self, s, r, dr, a, da, cx, cy, as_):
        self._chk(self.lib.lk_group_set_sector_annular(self._h, s, r, dr, a, da, cx, cy, as_))

    def set_sector_points(self, s, xy, center=None):
        a = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        cx, cy = center if center is not None else (0.0, 0.0)
        self._chk(self.lib.lk_group_set_sector_points(self._h, s, _ffi.fptr(a), a.shape[0], int(center is not None), cx, cy))

    def commit_sectors(self):
        self._chk(self.lib.lk_group_commit_sectors(self._h))

    @property
    def n_sectors(self):
        return self.lib.lk_group_sector_count(self._h)

    def correlate_all(self, guesses=None, fetch=True):
        S = self.n_sectors
        gp = None
        if guesses is not None:
            g = np.zeros((S, 6), np.float32)
            ga = np.asarray(guesses, np.float32)
            if ga.ndim == 1:
                g[:, :ga.shape[0]] = ga
            else:
                g[:, :ga.shape[1]] = ga
            gp = _ffi.fptr(g)
        out = np.zeros(S, _ffi.RESULT_DTYPE) if fetch else None
        self._chk(self.lib.lk_group_correlate_all(self._h, gp, out.ctypes.data_as(C.c_void_p) if fetch else None))
        return out

    def adjust_initial_guess(self, frame, constant_velocity, global_guess, global_center):
        g = np.zeros(6, np.float32)
        gg = np.asarray(global_guess, np.float32)
        g[:len(gg)] = gg
        self._chk(self.lib.lk_group_adjust_initial_guess(self._h, frame, int(bool(constant_velocity)), _ffi.fptr(g),
                                                         float(global_center[0]), float(global_center[1])))

    # ---- frame-pipelined windows ----------------------------------------------------------
    def sequence_reserve(self, n_slots):
        self._chk(self.lib.lk_group_sequence_reserve(self._h, int(n_slots)))

    def sequence_set_frames(self, first_slot, frames):
        """host frames (a list of 2-D u8 arrays of one size) into consecutive ring slots, in one transfer"""
        arrs = [np.ascontiguousarray(f, np.uint8) for f in frames]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        self._chk(self.lib.lk_group_sequence_set_frames(self._h, int(first_slot), len(arrs), ptrs, arrs[0].shape[0],
                                                        arrs[0].shape[1], arrs[0].strides[0]))

    def sequence_set_frames_device(self, first_slot, n_frames, dev_ptr, rows, cols, step=None):
        self._chk(self.lib.lk_group_sequence_set_frames_device(self._h, int(first_slot), int(n_frames), C.c_void_p(dev_ptr), rows, cols,
                                                               step or cols))

    def correlate_sequence_async(self, n_frames, first_slot=0, und_slot=-1, reference_previous=False, constant_velocity=True):
        self._seq_frames = int(n_frames)
        self._chk(self.lib.lk_group_correlate_sequence_async(self._h, int(und_slot), int(first_slot), int(n_frames),
                                                             int(bool(reference_previous)), int(bool(constant_velocity))))

    def wait_sequence(self, fetch=True):
        out = np.zeros((self._seq_frames, self.n_sectors), _ffi.RESULT_DTYPE) if fetch else None
        self._chk(self.lib.lk_group_wait_sequence(self._h, out.ctypes.data_as(C.c_void_p) if fetch else None))
        self._exchanged_frames = self._seq_frames
        return out

    def sequence_records(self):
        """the records of the window wait_sequence(fetch=False) last exchanged ([frames][S]); waits for that exchange only -
        call it after launching the next window and the records travel while that one is solved"""
        n = getattr(self, "_exchanged_frames", 0)   # (0: no window exchanged yet - the library says so)
        out = np.zeros((max(n, 1), self.n_sectors), _ffi.RESULT_DTYPE)
        self._chk(self.lib.lk_group_sequence_records(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def probe_overlap(self, rank=0):
        """(transfer ms, solve ms, transfer begin - solve begin, solve end - transfer end) on `rank`'s device"""
        ms = np.zeros(4, np.float32)
        self._chk(self.lib.lk_group_probe_overlap(self._h, int(rank), _ffi.fptr(ms)))
        return ms

    def synchronize(self):
        self._chk(self.lib.lk_group_synchronize(self._h))

    def stats(self):
        s = _ffi.LkStats()
        self._chk(self.lib.lk_group_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _ffi.LkStats._fields_}
