"""CPU ORACLE for the sequence / tracking driver around the hot path.

TEST INFRASTRUCTURE ONLY (tests/ import it; the product never does).

A restatement, in float32 scalar arithmetic, of the bookkeeping managerClass does around
`Newton_Raphson` for a sequence of frames - the parts that decide WHAT the engine is asked to
solve and how its records become the report:

  perform_multiframe_correlation            manager_class.cpp:1296-1496 (frame loop, image roles)
  perform_single_frame_correlation_*        :274-551 (rectangular), :553-813 (annular), :1000-1237 (blob)
  adjust_rectangular/annular/blob_domain    :2018-2090, :2092-2237, :2239-2310
  adjust_initial_guess                      :2602-2707
  add_pair (Lagrangian list translation)    :38-47
  update_results                            :2312-2428   (+ model*_distort_*, interpolation_class.cpp:3-43,
                                                            best_rotation_UVUxUyVxVy, parameters.cpp:55-58)
  update_global_results                     :2709-2753
  initializeReport / addFrameToReport       :2473-2525, :2430-2471

Parity status: "parity unpinned" - the reference has no tests or fixtures for this logic and
managerClass cannot be compiled here (Qt application class + OpenCV).  The solves themselves
come from oracle/lk_oracle.c.  GUI-only state (contours, overlay point lists) is not restated.
"""
import ctypes
import ctypes.util
import math

import numpy as np

from . import lk_oracle as lo

f32 = np.float32
PI = f32(3.14159265359)                      # parameters.hpp:23

# enums.hpp
DEF_STRICT_LAGRANGIAN, DEF_LAGRANGIAN, DEF_EULERIAN = 0, 1, 2
ERRMODE_STOP_ALL, ERRMODE_STOP_FRAME, ERRMODE_CONTINUE = 0, 1, 2
REF_FIRST, REF_PREVIOUS = 0, 1
DOMAIN_RECT, DOMAIN_ANNULAR, DOMAIN_BLOB = 0, 1, 2

_libm = ctypes.CDLL(ctypes.util.find_library("m"))
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]


def atan2f(y, x):
    return f32(_libm.atan2f(float(y), float(x)))


def fmt(v):
    """std::ostream << float / int / bool with default flags (6 significant digits, %g)."""
    if isinstance(v, (bool, np.bool_)):
        return "1" if v else "0"
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    return "%g" % float(v)


class Sector:
    """the scalar part of frame_results (domains.hpp:59-108)"""

    def __init__(self, n_params):
        z = f32(0)
        self.und_center_x = self.und_center_y = self.und_angle = self.und_e = z
        self.und_global_ro = self.und_global_ri = self.und_global_angle = z
        self.und_global_center_x = self.und_global_center_y = self.und_global_e = z
        self.def_center_x = self.def_center_y = self.def_angle = self.def_e = z
        self.def_global_ro = self.def_global_ri = self.def_global_angle = z
        self.def_global_center_x = self.def_global_center_y = self.def_global_e = z
        self.resulting = np.zeros(n_params, f32)
        self.previous_resulting = np.zeros(n_params, f32)
        self.initial_guess = np.zeros(n_params, f32)
        self.number_of_points = 0
        self.chi = z
        self.iterations = 0
        self.error_status = False
        self.error_code = 0
        self.past_und_center_x = self.past_und_center_y = z
        self.und_points = None     # und_inside_points
        self.def_points = None     # def_inside_points (getDefXY0)

    def snapshot(self):
        s = Sector(len(self.resulting))
        s.__dict__.update({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in self.__dict__.items()})
        return s


class ManagerOracle:
    def __init__(self, engine, model, domain_type, deformation=DEF_EULERIAN, reference_image=REF_FIRST,
                 error_mode=ERRMODE_CONTINUE, global_guess=None):
        self.o = engine                      # lk_oracle.Oracle with und/def(/nxt) images managed by the caller
        self.model = model
        self.P = lo.N_PARAMS[model] if hasattr(lo, "N_PARAMS") else {0: 1, 1: 2, 2: 3, 3: 6}[model]
        self.domain_type = domain_type
        self.deformation = deformation
        self.reference_image = reference_image
        self.error_mode = error_mode
        self.global_guess = np.zeros(self.P, f32)
        if global_guess is not None:
            self.global_guess[:] = np.asarray(global_guess, f32)[:self.P]
        self.report = []
        self.sectors = []
        self.error = False
        self.error_type = 0
        self.initialize_report()

    # ---- domains ------------------------------------------------------------------------
    def set_rect_domain(self, x_begin, y_begin, x_end, y_end, x_center, y_center, hs, vs):
        self.rect = dict(x_begin=f32(x_begin), y_begin=f32(y_begin), x_end=f32(x_end), y_end=f32(y_end),
                         x_center=f32(x_center), y_center=f32(y_center), hs=hs, vs=vs)
        self.results_i, self.results_j = hs, vs
        self.sectors = [Sector(self.P) for _ in range(hs * vs)]

    def set_annular_domain(self, r_inside, r_outside, x_center, y_center, rs, as_):
        self.ann = dict(ri=f32(r_inside), ro=f32(r_outside), cx=f32(x_center), cy=f32(y_center), rs=rs, as_=as_)
        self.results_i, self.results_j = rs, as_
        self.sectors = [Sector(self.P) for _ in range(rs * as_)]

    def set_blob_domain(self, contour, x_center, y_center):
        self.blob = dict(contour=np.asarray(contour, f32).reshape(-1, 2), cx=f32(x_center), cy=f32(y_center))
        self.results_i, self.results_j = 1, 1
        self.sectors = [Sector(self.P)]

    # ---- adjust_*_domain ----------------------------------------------------------------
    def _lagrangian_roll(self, s, with_annulus=False):
        s.und_global_center_x, s.und_global_center_y = s.def_global_center_x, s.def_global_center_y
        if with_annulus:
            s.und_global_ro, s.und_global_ri = s.def_global_ro, s.def_global_ri
            s.und_global_e = s.def_global_e
        s.und_global_angle = s.def_global_angle
        s.past_und_center_x, s.past_und_center_y = s.und_center_x, s.und_center_y
        s.und_center_x, s.und_center_y = s.def_center_x, s.def_center_y
        if with_annulus:
            s.und_e = s.def_e
        s.und_angle = s.def_angle

    def adjust_rectangular_domain(self, center_x, center_y, s, frame):       # :2018-2090
        if frame == 0:
            s.und_global_center_x, s.und_global_center_y = self.rect["x_center"], self.rect["y_center"]
            s.und_global_angle = s.und_global_e = f32(0)
            s.und_center_x, s.und_center_y = f32(center_x), f32(center_y)
            s.und_angle = f32(0)
            s.past_und_center_x, s.past_und_center_y = s.und_center_x, s.und_center_y
        elif self.deformation != DEF_EULERIAN:
            self._lagrangian_roll(s)
        return int(s.und_center_x + f32(0.5)), int(s.und_center_y + f32(0.5))

    def adjust_annular_domain(self, i, j, dr, da, s, frame):                   # :2092-2237
        as_ = self.ann["as_"]
        if frame == 0:
            s.und_global_center_x, s.und_global_center_y = self.ann["cx"], self.ann["cy"]
            s.und_global_ro, s.und_global_ri = self.ann["ro"], self.ann["ri"]
            s.und_global_e = s.und_global_angle = f32(0)
            if as_ > 1:
                center_angle = f32(f32(f32(0 + f32(j * da))) + f32(da / f32(2)))
                center_r = f32(f32(self.ann["ri"] + f32(i * dr)) + f32(dr / f32(2)))
                s.und_center_x = f32(s.und_global_center_x + f32(center_r * f32(math.cos(float(center_angle)))))
                s.und_center_y = f32(s.und_global_center_y + f32(center_r * f32(math.sin(float(center_angle)))))
            else:
                s.und_center_x, s.und_center_y = s.und_global_center_x, s.und_global_center_y
            s.past_und_center_x, s.past_und_center_y = s.und_center_x, s.und_center_y
            s.und_e = s.und_angle = f32(0)
        elif self.deformation != DEF_EULERIAN:
            self._lagrangian_roll(s, with_annulus=True)
        r = f32(self.ann["ri"] + f32(i * dr))
        a = f32(s.und_global_angle + f32(j * da))
        return r, a, s.und_global_center_x, s.und_global_center_y

    def adjust_blob_domain(self, s, frame):                                    # :2239-2310
        if frame == 0:
            s.und_global_center_x, s.und_global_center_y = self.blob["cx"], self.blob["cy"]
            s.und_global_angle = f32(0)
            s.und_center_x, s.und_center_y = self.blob["cx"], self.blob["cy"]
            s.past_und_center_x, s.past_und_center_y = s.und_center_x, s.und_center_y
            s.und_angle = f32(0)
        elif self.deformation != DEF_EULERIAN:
            self._lagrangian_roll(s)

    # ---- adjust_initial_guess (:2602-2707) ----------------------------------------------
    def adjust_initial_guess(self, s, frame):
        P, g = self.P, self.global_guess
        if frame == 0:
            s.initial_guess[:] = g
            dx = f32(s.und_center_x - s.und_global_center_x)
            dy = f32(s.und_center_y - s.und_global_center_y)
            if self.model in (0, 1, 2):
                # the reference reads global_initial_guess[2] for every one of these models; with
                # fewer than three parameters that is outside the array - treated as 0 here
                vx = g[2] if P > 2 else f32(0)
                s.initial_guess[0] = f32(s.initial_guess[0] + f32(f32(-dy) * vx))
                if P > 1:
                    s.initial_guess[1] = f32(s.initial_guess[1] + f32(dx * vx))
            else:
                ux, uy, vx, vy = g[2], g[3], g[4], g[5]
                s.initial_guess[0] = f32(s.initial_guess[0] + f32(f32(dx * ux) + f32(dy * uy)))
                s.initial_guess[1] = f32(s.initial_guess[1] + f32(f32(dx * vx) + f32(dy * vy)))
            s.previous_resulting[:] = s.initial_guess
        else:
            if self.deformation == DEF_EULERIAN and self.reference_image == REF_FIRST:
                for i in range(P):
                    s.initial_guess[i] = f32(s.resulting[i] + f32(s.resulting[i] - s.previous_resulting[i]))
            else:
                s.initial_guess[:] = s.resulting
            s.previous_resulting[:] = s.resulting
        return s.initial_guess.copy()

    # ---- the solve + update_results (:2312-2428) -----------------------------------------
    def _solve(self, s, guess, center):
        g = np.zeros(6, f32)
        g[:self.P] = guess
        res, tr = self.o.newton_raphson(g, s.und_points, center=center, trace_cap=4096)
        # CorrelationClass::getDefXY0 (correlation_class.cpp:884-896): def positions of the last
        # level-0 evaluation; with py_start > 0 the model is re-applied with the returned parameters
        p_last = np.array(res["p"], f32)
        if self.o.cfg.py_start == 0 and len(tr):
            p_last = np.array(tr[-1]["p_in"], f32)
        cx, cy = float(res["und_cx"]), float(res["und_cy"])
        pts = np.asarray(s.und_points, f32).reshape(-1, 2)
        s.def_points = np.array([lo.model_point(self.model, x, y, cx, cy, p_last)[:2] for x, y in pts], f32)
        return res

    def update_results(self, s, res):
        P = self.P
        s.chi = f32(res["chi"])
        s.number_of_points = int(res["n_points"])
        s.iterations = int(res["iterations"])
        s.error_code = int(res["error_code"])
        s.error_status = s.error_code != 0
        s.und_center_x, s.und_center_y = f32(res["und_cx"]), f32(res["und_cy"])
        s.resulting[:] = np.asarray(res["p"], f32)[:P]
        p = s.resulting
        if self.model == 2:
            s.def_angle = f32(p[2] + s.und_angle)
        elif self.model == 3:
            s.def_angle = f32(atan2f(f32(p[4] - p[3]), f32(f32(p[2] + p[5]) + f32(2))) + s.und_angle)
        else:
            s.def_angle = f32(0)
        s.def_e = f32(0)
        # def centre = the model applied to the und centre about itself (dx = dy = 0)
        x, y = s.und_center_x, s.und_center_y
        z = f32(0)
        if self.model == 0:
            s.def_center_x, s.def_center_y = f32(x + p[0]), y
        elif self.model == 1:
            s.def_center_x, s.def_center_y = f32(x + p[0]), f32(y + p[1])
        elif self.model == 2:
            s.def_center_x = f32(f32(x + p[0]) - f32(f32(y - y) * p[2]))
            s.def_center_y = f32(f32(y + p[1]) + f32(f32(x - x) * p[2]))
        else:
            s.def_center_x = f32(f32(f32(x + p[0]) + f32(z * p[2])) + f32(z * p[3]))
            s.def_center_y = f32(f32(f32(y + p[1]) + f32(z * p[4])) + f32(z * p[5]))

    def update_global_results(self):                                            # :2709-2753
        aa = acx = acy = ae = tn = f32(0)
        for s in self.sectors:
            n = f32(s.number_of_points)
            aa = f32(aa + f32(s.def_angle * n))
            acx = f32(acx + f32(s.def_center_x * n))
            acy = f32(acy + f32(s.def_center_y * n))
            ae = f32(ae + f32(s.def_e * n))
            tn = f32(tn + n)
        with np.errstate(all="ignore"):
            aa, acx, acy, ae = f32(aa / tn), f32(acx / tn), f32(acy / tn), f32(ae / tn)
            und_ro, und_ri = self.sectors[0].und_global_ro, self.sectors[0].und_global_ri
            def_ri = f32(f32(1) + f32(ae * f32(f32(und_ro / und_ri) - f32(1))))
        for s in self.sectors:
            s.def_global_angle, s.def_global_center_x, s.def_global_center_y, s.def_global_e = aa, acx, acy, ae
            s.def_global_ro = s.und_global_ro
            s.def_global_ri = def_ri

    # ---- Lagrangian list updates (:354-419) ----------------------------------------------
    def _update_points(self, s):
        if self.deformation == DEF_STRICT_LAGRANGIAN:
            s.und_points = s.def_points
        elif self.deformation == DEF_LAGRANGIAN:
            ox = f32(s.und_center_x - s.past_und_center_x)
            oy = f32(s.und_center_y - s.past_und_center_y)
            pts = np.asarray(s.und_points, f32).reshape(-1, 2)
            out = np.empty_like(pts)
            half = f32(0.5)
            # add_pair: (int)(offset + value + 0.5f) per coordinate, float32 sums
            out[:, 0] = np.trunc((ox + pts[:, 0]).astype(f32) + half).astype(f32)
            out[:, 1] = np.trunc((oy + pts[:, 1]).astype(f32) + half).astype(f32)
            s.und_points = out

    # ---- one frame ----------------------------------------------------------------------
    def run_frame(self, frame, und_name="und", def_name="def"):
        """perform_single_frame_correlation_* for the pair currently held by the oracle engine."""
        saved = None
        error = False
        stop_modes = (ERRMODE_STOP_ALL, ERRMODE_STOP_FRAME)
        if self.domain_type == DOMAIN_RECT:
            d = self.rect
            xdim, ydim, cen = lo.rect_sector_geometry(d["x_begin"], d["y_begin"], d["x_end"], d["y_end"], d["hs"], d["vs"])
            for k, s in enumerate(self.sectors):
                cx, cy = self.adjust_rectangular_domain(int(cen[k][0]), int(cen[k][1]), s, frame)
                guess = self.adjust_initial_guess(s, frame)
                if frame == 0:
                    s.und_points = lo.rect_points(cx - xdim, cy - ydim, cx + xdim, cy + ydim)
                else:
                    self._update_points(s)
                res = self._solve(s, guess, (float(cx), float(cy)))
                self.update_results(s, res)
                error = s.error_status
                if error:
                    self.error_type = s.error_code
                    if self.error_mode in stop_modes:
                        break
        elif self.domain_type == DOMAIN_ANNULAR:
            d = self.ann
            dr = f32(f32(d["ro"] - d["ri"]) / f32(d["rs"]))
            da = f32(f32(f32(2) * PI) / f32(d["as_"]))
            stop = False
            for i in range(d["rs"]):
                for j in range(d["as_"]):
                    s = self.sectors[i * d["as_"] + j]
                    r, a, cx, cy = self.adjust_annular_domain(i, j, dr, da, s, frame)
                    guess = self.adjust_initial_guess(s, frame)
                    if frame == 0:
                        s.und_points = lo.annular_points(r, dr, a, da, cx, cy, d["as_"])
                    else:
                        self._update_points(s)
                    res = self._solve(s, guess, None)
                    self.update_results(s, res)
                    error = s.error_status
                    if error:
                        self.error_type = s.error_code
                        if self.error_mode in stop_modes:
                            stop = True
                            break
                if stop:
                    break
        else:
            s = self.sectors[0]
            self.adjust_blob_domain(s, frame)
            guess = self.adjust_initial_guess(s, frame)
            if frame == 0:
                pts = lo.blob_points(self.blob["contour"])
                if pts is None:
                    self.error, self.error_type = True, 4
                    return True
                s.und_points = pts
            else:
                self._update_points(s)
            res = self._solve(s, guess, None)
            self.update_results(s, res)
            error = s.error_status
            if error:
                self.error_type = s.error_code
        self.update_global_results()
        self.error = error
        self.add_frame_to_report(frame, und_name, def_name)
        return error

    # ---- report (:2430-2525) --------------------------------------------------------------
    def initialize_report(self):
        cols = ["Frame#", "und_file_string", "def_file_string", "und_global_center_x", "und_global_center_y",
                "und_center_x", "und_center_y", "def_global_center_x", "def_global_center_y", "def_center_x",
                "def_center_y"]
        cols += [f"parameter_{p}" for p in range(self.P)] + [f"Initial_guess_{p}" for p in range(self.P)]
        cols += ["und_global_angle(rad)", "def_global_angle(rad)", "und_angle(rad)", "def_angle(rad)",
                 "def_angle(deg)", "chi", "number_of_points", "iterations", "error_status", "error_code"]
        self.report = [",".join(cols) + "\n"]

    def add_frame_to_report(self, frame, und_name, def_name):
        for s in self.sectors:
            v = [fmt(frame), und_name, def_name, fmt(s.und_global_center_x), fmt(s.und_global_center_y),
                 fmt(s.und_center_x), fmt(s.und_center_y), fmt(s.def_global_center_x),
                 fmt(s.def_global_center_y), fmt(s.def_center_x), fmt(s.def_center_y)]
            v += [fmt(x) for x in s.resulting] + [fmt(x) for x in s.initial_guess]
            v += [fmt(s.und_global_angle), fmt(s.def_global_angle), fmt(s.und_angle), fmt(s.def_angle),
                  fmt(f32(f32(s.def_angle * f32(180)) / PI)), fmt(s.chi), fmt(s.number_of_points),
                  fmt(s.iterations), fmt(s.error_status), fmt(s.error_code)]
            self.report.append(",".join(v) + "\n")

    def report_text(self):
        return "".join(self.report)
