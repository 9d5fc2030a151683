"""HipCorrelationEngine - thin Python host mirror of the reference's GPU engine surface
(`CudaClass`, cuda_class.cuh:46-79) on top of the C-ABI in include/lk_engine.h.

Method names follow the reference so call sites read like manager_class.cpp; every call
goes straight into liblk_engine.so (HIP).  Used by tests/ and bench.py; the C++ adapter a
maintainer would compile into the Qt application is include/lk_cuda_class_adapter.hpp.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import (FM_U, FM_UV, FM_UVQ, FM_UVUXUYVXVY, IM_BICUBIC, IM_BILINEAR,  # noqa: F401
                   IM_NEAREST, IMG_DEF, IMG_NXT, IMG_UND, RESULT_DTYPE, LkConfig, LkStats)


class LkError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"lk_engine error {code}: {message}")
        self.code = code


class HipCorrelationEngine:
    _lib = None

    def __init__(self, interpolation=IM_BICUBIC, fitting_model=FM_UVUXUYVXVY, precision=1e-3,
                 max_iters=50, py_start=0, py_step=1, py_stop=2, device=0):
        if HipCorrelationEngine._lib is None:
            HipCorrelationEngine._lib = _ffi.load_library()
        self.lib = HipCorrelationEngine._lib
        self.cfg = LkConfig(interpolation, fitting_model, precision, max_iters, py_start, py_step,
                            py_stop, device)
        self.n_params = _ffi.N_PARAMS[fitting_model]
        self._h = C.c_void_p()
        rc = self.lib.lk_create(C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise LkError(rc, "lk_create failed (no usable HIP device, or bad configuration)")

    # ---- plumbing -----------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            msg = self.lib.lk_last_error_string(self._h)
            raise LkError(rc, msg.decode() if msg else "")
        return rc

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.lk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @staticmethod
    def device_count():  # CudaClass::initialize
        if HipCorrelationEngine._lib is None:
            HipCorrelationEngine._lib = _ffi.load_library()
        return HipCorrelationEngine._lib.lk_device_count()

    def set_stream(self, hip_stream):
        self._chk(self.lib.lk_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_timing(self, enabled):
        self._chk(self.lib.lk_set_timing(self._h, int(bool(enabled))))

    def set_batch_invariant(self, enabled):
        self._chk(self.lib.lk_set_batch_invariant(self._h, int(bool(enabled))))

    def set_reference_order(self, threads=1):
        """Bit-identical records to the CPU engine with number_of_threads = threads (0: off)."""
        self._chk(self.lib.lk_set_reference_order(self._h, int(threads)))

    def set_pairs_in_flight(self, n):
        """lk_set_pairs_in_flight: how many engines solve side by side on this GPU (before commit)."""
        self._chk(self.lib.lk_set_pairs_in_flight(self._h, int(n)))

    def synchronize(self):
        self._chk(self.lib.lk_synchronize(self._h))

    # ---- images (resetImagePyramids / resetNextPyramid / make*PyramidFrom*) -------------
    def set_image(self, slot, pixels):
        a = np.ascontiguousarray(pixels, dtype=np.uint8)
        assert a.ndim == 2
        self._chk(self.lib.lk_set_image(self._h, slot, a.ctypes.data_as(C.c_void_p), a.shape[0],
                                        a.shape[1], a.strides[0]))

    def set_image_device(self, slot, dev_ptr, rows, cols, step=None):
        self._chk(self.lib.lk_set_image_device(self._h, slot, C.c_void_p(dev_ptr), rows, cols,
                                               cols if step is None else step))

    def set_image_pair_device(self, und_ptr, def_ptr, rows, cols, und_step=None, def_step=None):
        self._chk(self.lib.lk_set_image_pair_device(self._h, C.c_void_p(und_ptr), cols if und_step is None else und_step,
                                                    C.c_void_p(def_ptr), cols if def_step is None else def_step,
                                                    rows, cols))

    def set_undeformed_image(self, px):
        self.set_image(IMG_UND, px)

    def set_deformed_image(self, px):
        self.set_image(IMG_DEF, px)

    def set_next_image(self, px):
        self.set_image(IMG_NXT, px)

    def makeUndPyramidFromDef(self):
        self._chk(self.lib.lk_rotate_und_from_def(self._h))

    def makeDefPyramidFromNxt(self):
        self._chk(self.lib.lk_rotate_def_from_nxt(self._h))

    def get_pyramid_level(self, slot, level):
        r, c = C.c_int(), C.c_int()
        self._chk(self.lib.lk_get_pyramid_level(self._h, slot, level, None, C.byref(r), C.byref(c)))
        out = np.empty((r.value, c.value), np.uint8)
        self._chk(self.lib.lk_get_pyramid_level(self._h, slot, level, out.ctypes.data_as(C.c_void_p),
                                                C.byref(r), C.byref(c)))
        return out

    # ---- sectors (resetPolygon overloads) -----------------------------------------------
    def clear_sectors(self):
        self._chk(self.lib.lk_clear_sectors(self._h))

    def resetPolygon_rect(self, sector, x0, y0, x1, y1):
        self._chk(self.lib.lk_set_sector_rect(self._h, sector, x0, y0, x1, y1))

    def resetPolygon_annular(self, sector, r, dr, a, da, cx, cy, as_):
        self._chk(self.lib.lk_set_sector_annular(self._h, sector, r, dr, a, da, cx, cy, as_))

    def set_sectors_annular(self, first_sector, params, as_):
        """lk_set_sectors_annular: params [count][6] = (r, dr, a, da, cx, cy) per sector."""
        q = np.ascontiguousarray(params, np.float32).reshape(-1, 6)
        self._chk(self.lib.lk_set_sectors_annular(self._h, int(first_sector), len(q), _ffi.fptr(q), int(as_)))

    def resetPolygon_blob(self, sector, contour_xy):
        c = np.ascontiguousarray(contour_xy, dtype=np.float32).reshape(-1, 2)
        self._chk(self.lib.lk_set_sector_blob(self._h, sector, _ffi.fptr(c), c.shape[0]))

    def set_sector_points(self, sector, xy, center=None):
        a = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        cx, cy = (center if center is not None else (0.0, 0.0))
        self._chk(self.lib.lk_set_sector_points(self._h, sector, _ffi.fptr(a), a.shape[0],
                                                int(center is not None), cx, cy))

    def set_rect_grid(self, x_begin, y_begin, x_end, y_end, hs, vs, first=0, count=-1):
        self._chk(self.lib.lk_set_rect_grid(self._h, x_begin, y_begin, x_end, y_end, hs, vs, first,
                                            count))

    def commit_sectors(self):
        self._chk(self.lib.lk_commit_sectors(self._h))

    def translate_sectors(self, offsets, centers=None):
        """Lagrangian description: move every sector's samples by add_pair(offset)."""
        o = np.ascontiguousarray(offsets, np.float32).reshape(-1, 2)
        c = None if centers is None else np.ascontiguousarray(centers, np.float32).reshape(-1, 2)
        self._chk(self.lib.lk_translate_sectors(self._h, _ffi.fptr(o), None if c is None else _ffi.fptr(c)))

    def rewarp_sectors(self, centers=None):
        """Strict Lagrangian description: und samples <- def samples of the last solve."""
        c = None if centers is None else np.ascontiguousarray(centers, np.float32).reshape(-1, 2)
        self._chk(self.lib.lk_rewarp_sectors(self._h, None if c is None else _ffi.fptr(c)))

    def update_sector(self, sector, mode):
        """CudaClass::updatePolygon(iSector, deformationDescription), CPU-manager semantics."""
        self._chk(self.lib.lk_update_sector(self._h, sector, mode))

    def restore_sectors(self, first_sector):
        self._chk(self.lib.lk_restore_sectors(self._h, first_sector))

    def last_evaluated_parameters(self):
        out = np.zeros((self.n_sectors, 6), np.float32)
        self._chk(self.lib.lk_get_last_evaluated_parameters(self._h, _ffi.fptr(out)))
        return out

    @property
    def n_sectors(self):
        return self.lib.lk_sector_count(self._h)

    def sector_info(self, sector):
        n, cx, cy = C.c_int(), C.c_float(), C.c_float()
        self._chk(self.lib.lk_get_sector_info(self._h, sector, C.byref(n), C.byref(cx), C.byref(cy)))
        return n.value, cx.value, cy.value

    def sector_level_count(self, sector, level):
        n = C.c_int()
        self._chk(self.lib.lk_get_sector_level_count(self._h, sector, level, C.byref(n)))
        return n.value

    def getUndXY0ToCPU(self, sector):
        n = C.c_int()
        self._chk(self.lib.lk_get_und_xy(self._h, sector, None, 0, C.byref(n)))
        out = np.empty((n.value, 2), np.float32)
        self._chk(self.lib.lk_get_und_xy(self._h, sector, _ffi.fptr(out), n.value, C.byref(n)))
        return out

    def level_xy(self, level, sector, evaluation_copy=False):
        """the device's sample list of a sector at a pyramid level: the reference's order, or the row-major copy
        the lane groups of the default mode walk (lk_get_level_xy)"""
        n = C.c_int()
        self._chk(self.lib.lk_get_level_xy(self._h, level, int(evaluation_copy), sector, None, 0, C.byref(n)))
        out = np.empty((n.value, 2), np.float32)
        if n.value:
            self._chk(self.lib.lk_get_level_xy(self._h, level, int(evaluation_copy), sector, _ffi.fptr(out), n.value, C.byref(n)))
        return out

    def getDefXY0ToCPU(self, sector, p):
        pp = np.zeros(6, np.float32)
        pp[:len(p)] = p
        n = C.c_int()
        self._chk(self.lib.lk_get_def_xy(self._h, sector, _ffi.fptr(pp), None, 0, C.byref(n)))
        out = np.empty((n.value, 2), np.float32)
        self._chk(self.lib.lk_get_def_xy(self._h, sector, _ffi.fptr(pp), _ffi.fptr(out), n.value,
                                         C.byref(n)))
        return out

    # ---- solve --------------------------------------------------------------------------
    def correlate(self, sector, initial_guess):
        """CudaClass::correlate: returns (result record, updated guess)."""
        g = np.zeros(6, np.float32)
        g[:self.n_params] = np.asarray(initial_guess, np.float32)[:self.n_params]
        out = np.zeros(1, RESULT_DTYPE)
        self._chk(self.lib.lk_correlate(self._h, sector, _ffi.fptr(g), out.ctypes.data_as(C.c_void_p)))
        return out[0], g

    def correlate_all(self, guesses=None):
        S = self.n_sectors
        out = np.zeros(S, RESULT_DTYPE)
        if guesses is None:
            gp = None
        else:
            g = np.zeros((S, 6), np.float32)
            ga = np.asarray(guesses, np.float32)
            if ga.ndim == 1:
                g[:, :ga.shape[0]] = ga
            else:
                g[:, :ga.shape[1]] = ga
            gp = _ffi.fptr(g)
        self._chk(self.lib.lk_correlate_all(self._h, gp, out.ctypes.data_as(C.c_void_p)))
        return out

    def correlate_all_device(self, d_guesses_ptr, d_results_ptr):
        self._chk(self.lib.lk_correlate_all_device(self._h, C.c_void_p(d_guesses_ptr),
                                                   C.c_void_p(d_results_ptr)))

    def correlate_all_async(self):
        """lk_correlate_all_async: the engine-held guesses, no waiting; wait_results() fetches."""
        self._chk(self.lib.lk_correlate_all_async(self._h))

    def wait_results(self):
        out = np.zeros(self.n_sectors, RESULT_DTYPE)
        self._chk(self.lib.lk_wait_results(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- frame-pipelined windows (lk_correlate_sequence_async) --------------------------
    def sequence_reserve(self, n_slots):
        self._chk(self.lib.lk_sequence_reserve(self._h, int(n_slots)))

    def sequence_set_frame(self, slot, pixels):
        a = np.ascontiguousarray(pixels, dtype=np.uint8)
        assert a.ndim == 2
        self._chk(self.lib.lk_sequence_set_frame(self._h, int(slot), a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1],
                                                 a.strides[0]))

    def sequence_set_frame_device(self, slot, dev_ptr, rows, cols, step=None):
        self._chk(self.lib.lk_sequence_set_frame_device(self._h, int(slot), C.c_void_p(dev_ptr), rows, cols,
                                                        cols if step is None else step))

    def correlate_sequence_async(self, n_frames, first_slot=0, und_slot=-1, reference_previous=False,
                                 constant_velocity=True, host_records=True, keep_guesses=False):
        self._seq_frames = int(n_frames)
        self._chk(self.lib.lk_correlate_sequence_async(self._h, int(und_slot), int(first_slot), int(n_frames),
                                                       int(bool(reference_previous)), int(bool(constant_velocity)),
                                                       (1 if host_records else 0) | (2 if keep_guesses else 0)))

    def wait_sequence(self, host_records=True):
        """records [n_frames][S] of the window (or None when the window keeps them on the device)"""
        out = np.zeros((self._seq_frames, self.n_sectors), RESULT_DTYPE) if host_records else None
        self._chk(self.lib.lk_wait_sequence(self._h, out.ctypes.data_as(C.c_void_p) if host_records else None))
        return out

    def correlate_sequence(self, n_frames, **kw):
        self.correlate_sequence_async(n_frames, **kw)
        return self.wait_sequence(kw.get("host_records", True))

    def sequence_results_device(self):
        r, g = C.c_void_p(), C.c_void_p()
        self._chk(self.lib.lk_get_sequence_results_device(self._h, C.byref(r), C.byref(g)))
        return r.value, g.value

    def copy_sequence_records_device(self, dst_ptr, pitch_records):
        self._chk(self.lib.lk_copy_sequence_records_device(self._h, C.c_void_p(dst_ptr), C.c_size_t(int(pitch_records))))

    def sequence_guesses(self):
        g = np.zeros((self._seq_frames, self.n_sectors, 6), np.float32)
        self._chk(self.lib.lk_get_sequence_guesses(self._h, _ffi.fptr(g)))
        return g

    @property
    def sequence_is_pipelined(self):
        return bool(self.lib.lk_sequence_is_pipelined(self._h))

    def adjust_initial_guess(self, frame, constant_velocity, global_guess, global_center):
        g = np.zeros(6, np.float32)
        g[:len(global_guess)] = global_guess
        self._chk(self.lib.lk_adjust_initial_guess(self._h, frame, int(constant_velocity),
                                                   _ffi.fptr(g), global_center[0], global_center[1]))

    def get_guesses(self):
        g = np.zeros((self.n_sectors, 6), np.float32)
        self._chk(self.lib.lk_get_guesses(self._h, _ffi.fptr(g)))
        return g

    # ---- stand-alone pieces -------------------------------------------------------------
    def evaluate(self, sector, level, p):
        pp = np.zeros(6, np.float32)
        pp[:len(p)] = p
        A = np.zeros((6, 6), np.float32)
        b = np.zeros(6, np.float32)
        chi, err = C.c_float(), C.c_int()
        self._chk(self.lib.lk_evaluate(self._h, sector, level, _ffi.fptr(pp), _ffi.fptr(A),
                                       _ffi.fptr(b), C.byref(chi), C.byref(err)))
        return A, b, chi.value, err.value

    def sample(self, slot, level, xy):
        a = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        out = np.zeros((a.shape[0], 4), np.float32)
        self._chk(self.lib.lk_sample(self._h, slot, level, _ffi.fptr(a), a.shape[0], _ffi.fptr(out)))
        return out

    def damped_solve(self, A, b, lam, scaling, reference_solver=False):
        A = np.ascontiguousarray(A, np.float32)
        n = A.shape[0]
        b = np.ascontiguousarray(b, np.float32)
        dp = np.zeros(n, np.float32)
        self._chk(self.lib.lk_damped_solve(self._h, n, _ffi.fptr(A), _ffi.fptr(b), lam, scaling,
                                           int(reference_solver), _ffi.fptr(dp)))
        return dp

    def stats(self):
        s = LkStats()
        self._chk(self.lib.lk_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in LkStats._fields_}

    def sector_stats(self):
        """[S][4] uint32 of the last solve: evaluations, sample evaluations, point iterations, ill-conditioned solves."""
        out = np.zeros((self.n_sectors, 4), np.uint32)
        self._chk(self.lib.lk_get_sector_stats(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out
