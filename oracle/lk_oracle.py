"""ctypes loader of the CPU ORACLE (oracle/liblk_oracle.so) - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"parity unpinned" except for the pieces listed in lk_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liblk_oracle.so")
REF_LIB_PATH = os.path.join(HERE, "_ref", "libref_pieces.so")

IM_NEAREST, IM_BILINEAR, IM_BICUBIC = 0, 1, 2
FM_U, FM_UV, FM_UVQ, FM_UVUXUYVXVY = 0, 1, 2, 3
IMG_UND, IMG_DEF, IMG_NXT = 0, 1, 2
N_PARAMS = {0: 1, 1: 2, 2: 3, 3: 6}


class Config(C.Structure):
    _fields_ = [("interp", C.c_int), ("model", C.c_int), ("precision", C.c_float),
                ("max_iters", C.c_int), ("py_start", C.c_int), ("py_step", C.c_int),
                ("py_stop", C.c_int), ("n_threads", C.c_int), ("solver", C.c_int), ("cache_mode", C.c_int)]


RESULT_DTYPE = np.dtype([("p", np.float32, (6,)), ("chi", np.float32), ("n_points", np.int32),
                         ("iterations", np.int32), ("error_code", np.int32),
                         ("und_cx", np.float32), ("und_cy", np.float32)])
TRACE_DTYPE = np.dtype([("level", np.int32), ("kind", np.int32), ("iteration", np.int32),
                        ("p_in", np.float32, (6,)), ("chi", np.float32), ("lam", np.float32),
                        ("A", np.float32, (36,)), ("b", np.float32, (6,)), ("dp", np.float32, (6,)),
                        ("error", np.int32)])

_F = C.POINTER(C.c_float)
_U8 = C.POINTER(C.c_uint8)
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", HERE, "all"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.lko_create.restype = C.c_void_p
        L.lko_create.argtypes = [C.POINTER(Config)]
        L.lko_destroy.argtypes = [C.c_void_p]
        L.lko_set_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.lko_und_from_def.argtypes = [C.c_void_p]
        L.lko_def_from_nxt.argtypes = [C.c_void_p]
        L.lko_get_level.restype = C.c_void_p
        L.lko_get_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.lko_newton_raphson.argtypes = [C.c_void_p, _F, C.c_int, _F, C.c_int, C.c_float, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.lko_correlate_sectors.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _F, C.c_int,
                                            _F, _F, C.c_void_p, C.c_int]
        L.lko_pyramid_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.lko_bicubic_coeffs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _F]
        L.lko_interpolate.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, _F, _F, _F]
        L.lko_model_point.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _F, _F, _F, _F, _F]
        L.lko_evaluate.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                   C.c_int, _F, C.c_int, C.c_float, C.c_float, _F, _F, _F, _F]
        L.lko_damped_solve.argtypes = [C.c_int, _F, _F, C.c_float, C.c_float, _F]
        L.lko_colpiv_qr_solve.argtypes = [C.c_int, _F, _F, _F]
        L.lko_decimate.argtypes = [_F, C.c_int, C.c_int, _F]
        L.lko_translate_parameters.argtypes = [C.c_int, _F, C.c_int, C.c_int]
        L.lko_rect_sector_geometry.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                               C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
        L.lko_rect_points.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _F, C.c_int]
        L.lko_annular_points.restype = C.c_int64
        L.lko_annular_points.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                         C.c_int, _F, C.c_int64]
        L.lko_blob_points.restype = C.c_int64
        L.lko_blob_points.argtypes = [_F, C.c_int, _F, C.c_int64]
        L.lko_adjust_initial_guess.argtypes = [C.c_int, C.c_int, C.c_int, _F, C.c_float, C.c_float,
                                               C.c_float, C.c_float, _F, _F, _F]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(_F)


class Oracle:
    """The reference's CorrelationClass, restated (see lk_oracle.h)."""

    def __init__(self, interp=IM_BICUBIC, model=FM_UVUXUYVXVY, precision=1e-3, max_iters=50,
                 py_start=0, py_step=1, py_stop=2, cache_mode=0, n_threads=1, solver=0):
        self.L = lib()
        self.cfg = Config(interp, model, precision, max_iters, py_start, py_step, py_stop, n_threads,
                          solver, cache_mode)
        self.n_params = N_PARAMS[model]
        self.h = self.L.lko_create(C.byref(self.cfg))
        if not self.h:
            raise ValueError("bad oracle configuration")

    def close(self):
        if self.h:
            self.L.lko_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_image(self, slot, px):
        a = np.ascontiguousarray(px, np.uint8)
        rc = self.L.lko_set_image(self.h, slot, a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1])
        assert rc == 0

    def und_from_def(self):
        self.L.lko_und_from_def(self.h)

    def def_from_nxt(self):
        self.L.lko_def_from_nxt(self.h)

    def get_level(self, slot, level):
        r, c = C.c_int(), C.c_int()
        p = self.L.lko_get_level(self.h, slot, level, C.byref(r), C.byref(c))
        assert p
        buf = (C.c_uint8 * (r.value * c.value)).from_address(p)
        return np.frombuffer(buf, np.uint8).reshape(r.value, c.value).copy()

    def newton_raphson(self, guess, xy, center=None, trace_cap=0):
        p = np.zeros(6, np.float32)
        p[:self.n_params] = np.asarray(guess, np.float32)[:self.n_params]
        a = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        out = np.zeros(1, RESULT_DTYPE)
        trace = np.zeros(max(trace_cap, 1), TRACE_DTYPE)
        nt = C.c_int()
        cx, cy = center if center is not None else (0.0, 0.0)
        self.L.lko_newton_raphson(self.h, _fp(p), a.shape[0], _fp(a), int(center is not None), cx, cy,
                                  out.ctypes.data_as(C.c_void_p),
                                  trace.ctypes.data_as(C.c_void_p) if trace_cap else None, trace_cap,
                                  C.byref(nt))
        if trace_cap:
            return out[0], trace[:min(nt.value, trace_cap)]
        return out[0]

    def correlate_sectors(self, xy_list, centers=None, guesses=None, nthreads=1):
        S = len(xy_list)
        cnt = np.array([len(x) for x in xy_list], np.int32)
        off = np.zeros(S, np.int64)
        off[1:] = np.cumsum(cnt[:-1])
        cat = np.ascontiguousarray(np.concatenate([np.asarray(x, np.float32).reshape(-1, 2) for x in xy_list]))
        return self.correlate_packed(cat, off, cnt, centers, guesses, nthreads)

    def correlate_packed(self, cat, off, cnt, centers=None, guesses=None, nthreads=1):
        S = len(cnt)
        g = np.zeros((S, 6), np.float32)
        if guesses is not None:
            ga = np.asarray(guesses, np.float32)
            if ga.ndim == 1:
                g[:, :ga.shape[0]] = ga
            else:
                g[:, :ga.shape[1]] = ga
        out = np.zeros(S, RESULT_DTYPE)
        cen = np.ascontiguousarray(centers, np.float32) if centers is not None else np.zeros((S, 2), np.float32)
        off = np.ascontiguousarray(off, np.int64)
        cnt = np.ascontiguousarray(cnt, np.int32)
        cat = np.ascontiguousarray(cat, np.float32)
        self.L.lko_correlate_sectors(self.h, S, off.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p),
                                     _fp(cat), int(centers is not None), _fp(cen), _fp(g),
                                     out.ctypes.data_as(C.c_void_p), nthreads)
        return out


# ---- stand-alone functions -----------------------------------------------------------------
def pyramid_level(src):
    a = np.ascontiguousarray(src, np.uint8)
    out = np.zeros((a.shape[0] // 2, a.shape[1] // 2), np.uint8)
    lib().lko_pyramid_level(a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1], out.ctypes.data_as(C.c_void_p))
    return out


def bicubic_coeffs(img, ix, iy):
    a = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(16, np.float32)
    lib().lko_bicubic_coeffs(a.ctypes.data_as(C.c_void_p), a.shape[1], ix, iy, _fp(out))
    return out


def _padded(img):
    """Two zero guard rows below the image, like the oracle's and the engine's own level
    buffers: the reference's nearest sampler reads one row past the last one."""
    a = np.ascontiguousarray(img, np.uint8)
    return np.ascontiguousarray(np.vstack([a, np.zeros((2, a.shape[1]), np.uint8)])), a.shape[0]


def interpolate(interp, img, x, y, _pad=None):
    a, rows = _pad if _pad is not None else _padded(img)
    w, wx, wy = C.c_float(), C.c_float(), C.c_float()
    err = lib().lko_interpolate(interp, a.ctypes.data_as(C.c_void_p), rows, a.shape[1], x, y,
                                C.byref(w), C.byref(wx), C.byref(wy))
    return w.value, wx.value, wy.value, err


def interpolate_many(interp, img, xy):
    out = np.zeros((len(xy), 4), np.float32)
    pad = _padded(img)
    for k, (x, y) in enumerate(np.asarray(xy, np.float32)):
        out[k] = interpolate(interp, img, float(x), float(y), pad)
    return out


def model_point(model, x, y, cx, cy, p):
    pp = np.zeros(6, np.float32)
    pp[:len(p)] = p
    xd, yd = C.c_float(), C.c_float()
    dTx, dTy = np.zeros(6, np.float32), np.zeros(6, np.float32)
    lib().lko_model_point(model, x, y, cx, cy, _fp(pp), C.byref(xd), C.byref(yd), _fp(dTx), _fp(dTy))
    return xd.value, yd.value, dTx, dTy


def evaluate(interp, model, und, dfm, xy, cx, cy, p):
    u = np.ascontiguousarray(und, np.uint8)
    d = np.ascontiguousarray(dfm, np.uint8)
    a = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    pp = np.zeros(6, np.float32)
    pp[:len(p)] = p
    A, b, chi = np.zeros((6, 6), np.float32), np.zeros(6, np.float32), C.c_float()
    err = lib().lko_evaluate(interp, model, u.ctypes.data_as(C.c_void_p), u.shape[0], u.shape[1],
                             d.ctypes.data_as(C.c_void_p), d.shape[0], d.shape[1], _fp(a), a.shape[0],
                             cx, cy, _fp(pp), _fp(A), _fp(b), C.byref(chi))
    return A, b, chi.value, err


def damped_solve(A, b, lam, scaling):
    A = np.array(A, np.float32, copy=True)
    b = np.array(b, np.float32, copy=True)
    n = A.shape[0]
    dp = np.zeros(n, np.float32)
    lib().lko_damped_solve(n, _fp(A), _fp(b), lam, scaling, _fp(dp))
    return dp


def colpiv_qr_solve(A, b):
    A = np.ascontiguousarray(np.asarray(A, np.float32).T)  # column-major
    b = np.ascontiguousarray(b, np.float32)
    x = np.zeros(A.shape[0], np.float32)
    lib().lko_colpiv_qr_solve(A.shape[0], _fp(A), _fp(b), _fp(x))
    return x


def decimate(xy, delta):
    a = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(a)
    m = lib().lko_decimate(_fp(a), a.shape[0], delta, _fp(out))
    return out[:m].copy()


def rect_sector_geometry(x_begin, y_begin, x_end, y_end, hs, vs):
    xd, yd = C.c_int(), C.c_int()
    cen = np.zeros((hs * vs, 2), np.int32)
    lib().lko_rect_sector_geometry(x_begin, y_begin, x_end, y_end, hs, vs, C.byref(xd), C.byref(yd),
                                   cen.ctypes.data_as(C.c_void_p))
    return xd.value, yd.value, cen


def rect_points(x0, y0, x1, y1):
    n = (x1 - x0 + 1) * (y1 - y0 + 1)
    out = np.zeros((n, 2), np.float32)
    m = lib().lko_rect_points(x0, y0, x1, y1, _fp(out), n)
    assert m == n
    return out


def annular_points(r, dr, a, da, cx, cy, as_):
    m = lib().lko_annular_points(r, dr, a, da, cx, cy, as_, None, 0)
    out = np.zeros((max(m, 1), 2), np.float32)
    lib().lko_annular_points(r, dr, a, da, cx, cy, as_, _fp(out), m)
    return out[:m]


def blob_points(contour):
    c = np.ascontiguousarray(contour, np.float32).reshape(-1, 2)
    m = lib().lko_blob_points(_fp(c), c.shape[0], None, 0)
    if m < 0:
        return None
    out = np.zeros((max(m, 1), 2), np.float32)
    lib().lko_blob_points(_fp(c), c.shape[0], _fp(out), m)
    return out[:m]


def adjust_initial_guess(model, frame, cv, global_guess, scx, scy, gcx, gcy, resulting, previous):
    gg = np.zeros(6, np.float32)
    gg[:len(global_guess)] = global_guess
    res = np.zeros(6, np.float32)
    res[:len(resulting)] = resulting
    prev = np.array(previous, np.float32, copy=True)
    out = np.zeros(6, np.float32)
    lib().lko_adjust_initial_guess(model, frame, int(cv), _fp(gg), scx, scy, gcx, gcy, _fp(res), _fp(prev), _fp(out))
    return out, prev


# ---- the partial reference build (only where /root/reference was present at build time) ----
def ref_lib():
    if not os.path.exists(REF_LIB_PATH):
        return None
    R = C.CDLL(REF_LIB_PATH)
    R.ref_compute_model.argtypes = [C.c_int, C.c_int, _F, _F, C.c_float, C.c_float, _F, _F]
    R.ref_blob_points.restype = C.c_long
    R.ref_blob_points.argtypes = [_F, C.c_int, _F, C.c_long]
    R.ref_best_rotation.restype = C.c_float
    R.ref_best_rotation.argtypes = [_F]
    R.ref_blob_center.restype = C.c_float
    R.ref_blob_center.argtypes = [_F, C.c_int, C.c_int]
    return R
