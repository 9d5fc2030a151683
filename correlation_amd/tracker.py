"""SequenceTracker / run_sequence - Python host mirror of include/lk_tracker.h.

The tracker is managerClass's bookkeeping around the solve (sector positions per frame,
initial guesses, frame_results, the CSV report); `run_sequence` is
perform_multiframe_correlation on a HipCorrelationEngine.  All logic lives in
liblk_engine.so (correlation_amd/csrc/lk_tracker.cpp); this file only marshals.
"""
import ctypes as C

import numpy as np

from . import _ffi

DEF_STRICT_LAGRANGIAN, DEF_LAGRANGIAN, DEF_EULERIAN = 0, 1, 2          # deformationDescriptionEnum
ERRMODE_STOP_ALL, ERRMODE_STOP_FRAME, ERRMODE_CONTINUE = 0, 1, 2        # errorHandlingModeEnum
REF_FIRST, REF_PREVIOUS = 0, 1                                          # referenceImageEnum
DOMAIN_RECT, DOMAIN_ANNULAR, DOMAIN_BLOB = 0, 1, 2                      # domainEnum
(SECTOR_KEEP, SECTOR_RECT, SECTOR_ANNULAR, SECTOR_BLOB, SECTOR_TRANSLATE, SECTOR_REWARP) = range(6)


class TrackerConfig(C.Structure):
    _fields_ = [("fitting_model", C.c_int), ("domain_type", C.c_int), ("deformation", C.c_int),
                ("reference_image", C.c_int), ("error_mode", C.c_int), ("global_guess", C.c_float * 6)]


FRAME_RESULT_DTYPE = np.dtype([
    ("und_center_x", np.float32), ("und_center_y", np.float32), ("und_angle", np.float32), ("und_e", np.float32),
    ("und_global_ro", np.float32), ("und_global_ri", np.float32), ("und_global_angle", np.float32),
    ("und_global_center_x", np.float32), ("und_global_center_y", np.float32), ("und_global_e", np.float32),
    ("def_center_x", np.float32), ("def_center_y", np.float32), ("def_angle", np.float32), ("def_e", np.float32),
    ("def_global_ro", np.float32), ("def_global_ri", np.float32), ("def_global_angle", np.float32),
    ("def_global_center_x", np.float32), ("def_global_center_y", np.float32), ("def_global_e", np.float32),
    ("resulting_parameters", np.float32, (6,)), ("previous_resulting_parameters", np.float32, (6,)),
    ("initial_guess", np.float32, (6,)),
    ("number_of_points", np.int32), ("chi", np.float32), ("iterations", np.int32), ("error_status", np.int32),
    ("error_code", np.int32), ("past_und_center_x", np.float32), ("past_und_center_y", np.float32)])

COMMAND_DTYPE = np.dtype([
    ("kind", np.int32), ("use_center", np.int32), ("center_x", np.float32), ("center_y", np.float32),
    ("x0", np.int32), ("y0", np.int32), ("x1", np.int32), ("y1", np.int32),
    ("r", np.float32), ("dr", np.float32), ("a", np.float32), ("da", np.float32), ("cx", np.float32),
    ("cy", np.float32), ("as", np.int32), ("offset_x", np.float32), ("offset_y", np.float32)])

FRAME_PROVIDER = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                             C.POINTER(C.c_int), C.POINTER(C.c_char_p))


class SequenceTracker:
    def __init__(self, fitting_model, domain_type, deformation=DEF_EULERIAN, reference_image=REF_FIRST,
                 error_mode=ERRMODE_CONTINUE, global_guess=None, lib=None):
        self.lib = lib or _ffi.load_library()
        cfg = TrackerConfig(fitting_model, domain_type, deformation, reference_image, error_mode)
        g = np.zeros(6, np.float32)
        if global_guess is not None:
            ga = np.asarray(global_guess, np.float32)
            g[:len(ga)] = ga
        for i in range(6):
            cfg.global_guess[i] = float(g[i])
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = self.lib.lk_tracker_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise ValueError(f"lk_tracker_create failed ({rc})")

    def _chk(self, rc):
        if rc != 0:
            msg = self.lib.lk_tracker_last_error(self._h)
            raise RuntimeError(f"lk_tracker error {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.lk_tracker_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_rect_domain(self, x_begin, y_begin, x_end, y_end, x_center, y_center, hs, vs):
        self._chk(self.lib.lk_tracker_set_rect_domain(self._h, x_begin, y_begin, x_end, y_end, x_center,
                                                      y_center, hs, vs))

    def set_annular_domain(self, r_inside, r_outside, x_center, y_center, rs, as_):
        self._chk(self.lib.lk_tracker_set_annular_domain(self._h, r_inside, r_outside, x_center, y_center, rs, as_))

    def set_blob_domain(self, contour, x_center, y_center):
        c = np.ascontiguousarray(contour, np.float32).reshape(-1, 2)
        self._chk(self.lib.lk_tracker_set_blob_domain(self._h, _ffi.fptr(c), c.shape[0], x_center, y_center))

    def enable_report(self, enabled):
        self._chk(self.lib.lk_tracker_enable_report(self._h, int(bool(enabled))))

    @property
    def n_sectors(self):
        return self.lib.lk_tracker_sector_count(self._h)

    def begin_frame(self, frame):
        S = self.n_sectors
        cmds = np.zeros(S, COMMAND_DTYPE)
        guesses = np.zeros((S, 6), np.float32)
        self._chk(self.lib.lk_tracker_begin_frame(self._h, frame, cmds.ctypes.data_as(C.c_void_p),
                                                  _ffi.fptr(guesses)))
        return cmds, guesses

    def end_frame(self, frame, und_name, def_name, results):
        r = np.ascontiguousarray(results, _ffi.RESULT_DTYPE)
        first, stop = C.c_int(), C.c_int()
        self._chk(self.lib.lk_tracker_end_frame(self._h, frame, und_name.encode(), def_name.encode(),
                                                r.ctypes.data_as(C.c_void_p), C.byref(first), C.byref(stop)))
        return first.value, bool(stop.value)

    def results(self):
        out = np.zeros(self.n_sectors, FRAME_RESULT_DTYPE)
        self._chk(self.lib.lk_tracker_get_results(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def report(self):
        need = C.c_size_t()
        self._chk(self.lib.lk_tracker_report(self._h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        self._chk(self.lib.lk_tracker_report(self._h, buf, need.value, C.byref(need)))
        return buf.value.decode()


def sequence_frame(engine, tracker, frame, und_name="und", def_name="def"):
    """One pair on the engine (lk_sequence_frame); returns True when the sequence has to stop."""
    stop = C.c_int()
    rc = engine.lib.lk_sequence_frame(engine._h, tracker._h, frame, und_name.encode(), def_name.encode(),
                                      C.byref(stop))
    if rc != 0:
        engine._chk(rc)
    return bool(stop.value)


def run_sequence(engine, tracker, frames, names=None):
    """perform_multiframe_correlation: `frames` is a list of 2-D uint8 arrays.  Returns the
    number of pairs correlated."""
    frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
    names = [n.encode() for n in (names or [f"frame{i}" for i in range(len(frames))])]

    def provide(_user, index, rows, cols, step, name):
        f = frames[index]
        rows[0], cols[0], step[0] = f.shape[0], f.shape[1], f.strides[0]
        name[0] = names[index]
        return f.ctypes.data

    cb = FRAME_PROVIDER(provide)
    done = C.c_int()
    rc = engine.lib.lk_sequence_run(engine._h, tracker._h, len(frames), cb, None, C.byref(done))
    if rc != 0:
        engine._chk(rc)
    return done.value


def load_pgm(path, lib=None):
    lib = lib or _ffi.load_library()
    px, r, c = C.POINTER(C.c_uint8)(), C.c_int(), C.c_int()
    rc = lib.lk_load_pgm(path.encode(), C.byref(px), C.byref(r), C.byref(c))
    if rc != 0:
        raise IOError(f"lk_load_pgm({path}) failed ({rc})")
    out = np.ctypeslib.as_array(px, (r.value, c.value)).copy()
    lib.lk_free_image(px)
    return out


def load_image(path, lib=None):
    """lk_load_image: the file as 8-bit grey, what cv::imread(path, IMREAD_GRAYSCALE) hands the reference
    (PNG, uncompressed BMP, PNM)"""
    lib = lib or _ffi.load_library()
    px, r, c = C.POINTER(C.c_uint8)(), C.c_int(), C.c_int()
    rc = lib.lk_load_image(str(path).encode(), C.byref(px), C.byref(r), C.byref(c))
    if rc != 0:
        raise IOError(f"lk_load_image({path}) failed ({rc})")
    out = np.ctypeslib.as_array(px, (r.value, c.value)).copy()
    lib.lk_free_image(px)
    return out


def decode_image(data, lib=None):
    """lk_decode_image: load_image for a file already in memory (bytes)"""
    lib = lib or _ffi.load_library()
    px, r, c = C.POINTER(C.c_uint8)(), C.c_int(), C.c_int()
    rc = lib.lk_decode_image(bytes(data), len(data), C.byref(px), C.byref(r), C.byref(c))
    if rc != 0:
        raise IOError(f"lk_decode_image failed ({rc})")
    out = np.ctypeslib.as_array(px, (r.value, c.value)).copy()
    lib.lk_free_image(px)
    return out
