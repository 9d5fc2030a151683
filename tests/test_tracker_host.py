"""Host logic of the sequence tracker (include/lk_tracker.h) against the manager oracle
(oracle/lk_manager_oracle.py), CPU only: the tracker's commands are applied to plain Python
sample lists, the solves come from the CPU oracle, and the tracker's frame_results and CSV
report must equal the oracle's restatement of managerClass exactly - both sides see the same
48-byte records, so every float of the bookkeeping is compared bit for bit."""
import numpy as np
import pytest

import correlation_amd as ca
from correlation_amd import tracker as tk
from oracle import lk_manager_oracle as mo


@pytest.fixture(scope="module")
def frames():
    return ca.speckle.speckle_sequence(256, 256, 4, velocity=(0.9, -0.5), dilation=4e-4, seed=3)


class OracleBackedSectors:
    """applies lk_sector_command records to sample lists and solves them with the CPU oracle -
    what lk_sequence_frame does with the HIP engine"""

    def __init__(self, oracle, o, model):
        self.oracle, self.o, self.model = oracle, o, model
        self.lists, self.centers, self.last = [], [], []

    def apply(self, cmds, tracker):
        for s, c in enumerate(cmds):
            kind = int(c["kind"])
            if kind == tk.SECTOR_RECT:
                pts = self.oracle.rect_points(int(c["x0"]), int(c["y0"]), int(c["x1"]), int(c["y1"]))
            elif kind == tk.SECTOR_ANNULAR:
                pts = self.oracle.annular_points(c["r"], c["dr"], c["a"], c["da"], c["cx"], c["cy"], int(c["as"]))
            elif kind == tk.SECTOR_BLOB:
                pts = self.oracle.blob_points(self.blob)
            elif kind == tk.SECTOR_TRANSLATE:
                p = self.lists[s]
                pts = np.empty_like(p)
                pts[:, 0] = np.trunc((c["offset_x"] + p[:, 0]).astype(np.float32) + np.float32(0.5))
                pts[:, 1] = np.trunc((c["offset_y"] + p[:, 1]).astype(np.float32) + np.float32(0.5))
            elif kind == tk.SECTOR_REWARP:
                pts = self.last[s]
            else:
                pts = self.lists[s]
            if s < len(self.lists):
                self.lists[s] = pts
            else:
                self.lists.append(pts)
                self.last.append(None)
        self.centers = [(float(c["center_x"]), float(c["center_y"])) if c["use_center"] else None for c in cmds]

    def solve(self, guesses):
        out = np.zeros(len(self.lists), ca.RESULT_DTYPE)
        for s, pts in enumerate(self.lists):
            res, tr = self.o.newton_raphson(guesses[s], pts, center=self.centers[s], trace_cap=4096)
            out[s] = res
            p_last = np.array(tr[-1]["p_in"], np.float32) if len(tr) else np.array(res["p"], np.float32)
            cx, cy = float(res["und_cx"]), float(res["und_cy"])
            self.last[s] = np.array([self.oracle.model_point(self.model, x, y, cx, cy, p_last)[:2]
                                     for x, y in pts], np.float32)
        return out


def make_domain(kind, t, m):
    if kind == tk.DOMAIN_RECT:
        args = (40.0, 44.0, 215.0, 211.0, 127.5, 127.5, 3, 2)
        t.set_rect_domain(*args)
        m.set_rect_domain(*args)
    elif kind == tk.DOMAIN_ANNULAR:
        args = (30.0, 78.0, 128.0, 126.0, 2, 3)
        t.set_annular_domain(*args)
        m.set_annular_domain(*args)
    else:
        ang = 2 * np.pi * np.arange(7) / 7
        contour = np.stack([128 + 45 * np.cos(ang), 126 + 38 * np.sin(ang)], 1).astype(np.float32)
        t.set_blob_domain(contour, 128.0, 126.0)
        m.set_blob_domain(contour, 128.0, 126.0)
        return contour
    return None


CASES = [(tk.DOMAIN_RECT, tk.DEF_EULERIAN, tk.REF_FIRST, ca.FM_UVUXUYVXVY),
         (tk.DOMAIN_RECT, tk.DEF_LAGRANGIAN, tk.REF_PREVIOUS, ca.FM_UVUXUYVXVY),
         (tk.DOMAIN_RECT, tk.DEF_STRICT_LAGRANGIAN, tk.REF_PREVIOUS, ca.FM_UV),
         (tk.DOMAIN_ANNULAR, tk.DEF_EULERIAN, tk.REF_FIRST, ca.FM_UVQ),
         (tk.DOMAIN_ANNULAR, tk.DEF_LAGRANGIAN, tk.REF_PREVIOUS, ca.FM_UVUXUYVXVY),
         (tk.DOMAIN_BLOB, tk.DEF_STRICT_LAGRANGIAN, tk.REF_PREVIOUS, ca.FM_UVUXUYVXVY),
         (tk.DOMAIN_BLOB, tk.DEF_EULERIAN, tk.REF_PREVIOUS, ca.FM_U)]


@pytest.mark.parametrize("threaded", [False, True])
@pytest.mark.parametrize("domain,deformation,reference,model", CASES)
def test_tracker_equals_manager_oracle(oracle, engine_lib, frames, monkeypatch, domain, deformation, reference, model,
                                       threaded):
    # threaded: one sector per block, so the helper threads of the per-sector loops and of the
    # report (which only start on grids of thousands of sectors) run on these small domains too
    monkeypatch.setenv("LK_TRACKER_MIN_BLOCK", "1" if threaded else "4096")
    guess = [0.5, -0.25, 1e-3, 0.0, 0.0, 2e-3]
    t = tk.SequenceTracker(model, domain, deformation, reference, tk.ERRMODE_CONTINUE, guess, lib=engine_lib)

    def engines():
        o = oracle.Oracle(model=model)
        o.set_image(0, frames[0])
        o.set_image(1, frames[1])
        return o

    o_t, o_m = engines(), engines()
    m = mo.ManagerOracle(o_m, model, domain, deformation, reference, mo.ERRMODE_CONTINUE, guess)
    sectors = OracleBackedSectors(oracle, o_t, model)
    sectors.blob = make_domain(domain, t, m)
    assert t.n_sectors == len(m.sectors)
    for k in range(len(frames) - 1):
        if k > 0:   # image roles (manager_class.cpp:1386-1407, :166-243)
            for o in (o_t, o_m):
                if reference == tk.REF_PREVIOUS:
                    o.und_from_def()
                o.set_image(2, frames[k + 1])
                o.def_from_nxt()
        und = "f0" if reference == tk.REF_FIRST else f"f{k}"
        cmds, guesses = t.begin_frame(k)
        sectors.apply(cmds, t)
        first, stop = t.end_frame(k, und, f"f{k + 1}", sectors.solve(guesses))
        assert first == t.n_sectors and not stop
        m.run_frame(k, und, f"f{k + 1}")
        got = t.results()
        for s, want in enumerate(m.sectors):
            for name in ("und_center_x", "und_center_y", "und_angle", "und_global_center_x", "und_global_center_y",
                         "und_global_angle", "def_center_x", "def_center_y", "def_global_center_x",
                         "def_global_center_y", "def_global_angle", "chi", "past_und_center_x",
                         "past_und_center_y", "und_global_ro", "und_global_ri", "def_global_ro"):
                a, b = got[name][s], np.float32(getattr(want, name))
                assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), (k, s, name, a, b)
            assert abs(float(got["def_angle"][s]) - float(want.def_angle)) <= 1e-7   # atan2f vs libm call
            P = len(want.resulting)
            assert np.array_equal(got["resulting_parameters"][s][:P], want.resulting)
            assert np.array_equal(got["initial_guess"][s][:P], want.initial_guess)
            assert np.array_equal(got["previous_resulting_parameters"][s][:P], want.previous_resulting)
            assert (got["number_of_points"][s], got["iterations"][s], got["error_code"][s]) == \
                   (want.number_of_points, want.iterations, want.error_code)
    assert t.report() == m.report_text()
    assert t.report().count("\n") == 1 + (len(frames) - 1) * t.n_sectors
    t.close()


@pytest.mark.parametrize("threaded", [False, True])
def test_tracker_stop_policy(oracle, engine_lib, frames, monkeypatch, threaded):
    """stopFrame / stopAll: sectors after the first failing one keep the state they had before
    the frame, and stopAll ends the sequence (manager_class.cpp:520-546, :1485-1486)."""
    monkeypatch.setenv("LK_TRACKER_MIN_BLOCK", "1" if threaded else "4096")
    for mode in (tk.ERRMODE_STOP_FRAME, tk.ERRMODE_STOP_ALL, tk.ERRMODE_CONTINUE):
        t = tk.SequenceTracker(ca.FM_UV, tk.DOMAIN_RECT, tk.DEF_LAGRANGIAN, tk.REF_PREVIOUS, mode, lib=engine_lib)
        t.set_rect_domain(40.0, 44.0, 215.0, 211.0, 127.5, 127.5, 2, 2)
        cmds, guesses = t.begin_frame(0)
        res = np.zeros(4, ca.RESULT_DTYPE)
        res["n_points"] = 100
        res["p"][:, 0] = [1.0, 2.0, 3.0, 4.0]
        res["und_cx"], res["und_cy"] = cmds["center_x"], cmds["center_y"]
        first, stop = t.end_frame(0, "a", "b", res)
        assert first == 4 and not stop
        before = t.results()
        cmds, guesses = t.begin_frame(1)
        assert list(cmds["kind"]) == [tk.SECTOR_TRANSLATE] * 4
        assert np.allclose(cmds["offset_x"], [1, 2, 3, 4])
        res["error_code"] = [0, ca.ERROR_INTERPOLATION_OUT_OF_IMAGE, 0, 0]
        res["und_cx"], res["und_cy"] = cmds["center_x"], cmds["center_y"]
        first, stop = t.end_frame(1, "b", "c", res)
        after = t.results()
        if mode == tk.ERRMODE_CONTINUE:
            assert first == 4 and not stop
        else:
            assert first == 2 and stop == (mode == tk.ERRMODE_STOP_ALL)
            for name in ("und_center_x", "past_und_center_x", "def_center_x", "resulting_parameters",
                         "initial_guess", "previous_resulting_parameters", "error_code"):
                assert np.array_equal(after[name][2:], before[name][2:]), name
            assert after["error_code"][1] == ca.ERROR_INTERPOLATION_OUT_OF_IMAGE
        rows = t.report().strip().split("\n")
        assert len(rows) == 1 + 8 and rows[6].split(",")[-1] == str(ca.ERROR_INTERPOLATION_OUT_OF_IMAGE)
        t.close()


def test_report_of_a_large_grid_is_in_sector_order(engine_lib):
    """Grids of thousands of sectors format their report rows on several threads: the text must
    be the sequential loop's (manager_class.cpp:2430-2471), row for row."""
    t = tk.SequenceTracker(ca.FM_UVQ, tk.DOMAIN_RECT, tk.DEF_EULERIAN, tk.REF_FIRST, tk.ERRMODE_CONTINUE,
                           lib=engine_lib)
    t.set_rect_domain(20.0, 24.0, 2027.0, 2023.0, 1024.0, 1024.0, 90, 80)
    S = t.n_sectors
    assert S == 7200
    rng = np.random.default_rng(5)
    want = []
    for k in range(2):
        cmds, guesses = t.begin_frame(k)
        res = np.zeros(S, ca.RESULT_DTYPE)
        res["p"][:, :3] = rng.normal(0, 1.5, (S, 3)).astype(np.float32) * np.float32([1, 1, 1e-2])
        # the number formatter: every decade from 1e-30 to 1e30, random mantissas, and values
        # on or next to a rounding boundary of the sixth digit
        wild = (rng.integers(27, 227, 3 * 3000).astype(np.uint32) << 23 | rng.integers(0, 1 << 23, 3 * 3000).astype(np.uint32)
                | rng.integers(0, 2, 3 * 3000).astype(np.uint32) << 31).view(np.float32)
        res["p"][:3000, :3] = wild.reshape(3000, 3)
        edge = np.float32([1234565.0, 0.5, 2.5, 1e5, 999999.5, 9.999995, 100000.5, 1e-5, 1e-4, 1.234565e-5, 999999.0,
                           999999.44, 1e6, 123456.5, 123457.5, 0.000123456, 16777216.0, 1e-16, 1e16, 3.0000002e-20,
                           8388608.5, 0.1, 0.3, 1.0, 10.0, 99999.95, 0.999999, 0.9999995, 1.5e-5, 4.5, 1e22, 1e23])
        edge = np.concatenate([edge, np.nextafter(edge, np.float32(np.inf)), np.nextafter(edge, np.float32(0)), -edge])
        edge = np.resize(edge, 3 * ((len(edge) + 2) // 3))
        res["p"][3000:3000 + len(edge) // 3, :3] = edge.reshape(-1, 3)
        res["chi"] = rng.uniform(0.5, 900.0, S).astype(np.float32)
        res["n_points"] = rng.integers(1, 500, S)
        res["iterations"] = rng.integers(1, 40, S)
        res["error_code"] = np.where(rng.random(S) < 0.02, ca.ERROR_CORRELATION_MAX_ITERS_REACHED, 0)
        res["und_cx"], res["und_cy"] = cmds["center_x"], cmds["center_y"]
        first, stop = t.end_frame(k, f"und{k}", f"def{k}", res)
        assert first == S and not stop
        got = t.results()
        for s in range(S):
            r = got[s]
            v = [mo.fmt(k), f"und{k}", f"def{k}"]
            v += [mo.fmt(r[n]) for n in ("und_global_center_x", "und_global_center_y", "und_center_x", "und_center_y",
                                         "def_global_center_x", "def_global_center_y", "def_center_x", "def_center_y")]
            v += [mo.fmt(x) for x in r["resulting_parameters"][:3]] + [mo.fmt(x) for x in r["initial_guess"][:3]]
            v += [mo.fmt(r[n]) for n in ("und_global_angle", "def_global_angle", "und_angle", "def_angle")]
            v += [mo.fmt(np.float32(np.float32(r["def_angle"] * np.float32(180)) / mo.PI)), mo.fmt(r["chi"]),
                  mo.fmt(int(r["number_of_points"])), mo.fmt(int(r["iterations"])), mo.fmt(bool(r["error_status"])),
                  mo.fmt(int(r["error_code"]))]
            want.append(",".join(v))
    rows = t.report().split("\n")
    assert rows[-1] == "" and len(rows) == 2 + 2 * S
    assert rows[1:-1] == want
    t.close()


def test_pgm_loader(engine_lib, tmp_path):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    path = tmp_path / "a.pgm"
    path.write_bytes(b"P5\n# a comment\n53 37\n255\n" + img.tobytes())
    assert np.array_equal(tk.load_pgm(str(path), engine_lib), img)
    (tmp_path / "bad.pgm").write_bytes(b"P2\n1 1\n255\n0\n")
    with pytest.raises(IOError):
        tk.load_pgm(str(tmp_path / "bad.pgm"), engine_lib)
    (tmp_path / "short.pgm").write_bytes(b"P5\n4 4\n255\n" + b"\0" * 5)
    with pytest.raises(IOError):
        tk.load_pgm(str(tmp_path / "short.pgm"), engine_lib)
