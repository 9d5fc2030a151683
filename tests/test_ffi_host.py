"""CPU-side checks of the product boundary: the C-ABI library builds for gfx950, loads,
exports every symbol include/lk_engine.h, lk_tracker.h and lk_group.h declare, and refuses to run without a HIP
device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import correlation_amd as ca
from correlation_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for header in ("lk_engine.h", "lk_tracker.h", "lk_group.h"):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        hdr = re.sub(r"typedef[^;]*\(\*lk_[a-z_0-9]+\)[^;]*;", "", hdr)      # function-pointer typedefs
        names |= set(re.findall(r"\b(lk_[a-z_0-9]+)\s*\(", hdr))
    return sorted(names)


def test_library_exports_every_declared_symbol(engine_lib):
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(engine_lib, n), f"{n} declared in include/*.h but not exported"
        assert n in _ffi.SYMBOLS, f"{n} has no ctypes prototype"
    assert sorted(_ffi.SYMBOLS) == names


def test_result_record_layout_matches_reference_struct():
    # CorrelationResult (domains.hpp:110-118): 6 floats, float, int, int, enum, 2 floats
    d = ca.RESULT_DTYPE
    assert d.itemsize == 48
    assert [d.fields[k][1] for k in ("p", "chi", "n_points", "iterations", "error_code", "und_cx", "und_cy")] == \
        [0, 24, 28, 32, 36, 40, 44]


def test_code_object_is_gfx950(engine_lib):
    blob = open(ca.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"lk_solve_kernel" in blob and b"lk_pyramid_kernel" in blob


def test_create_validates_configuration(engine_lib):
    h = C.c_void_p()
    bad = _ffi.LkConfig(2, 9, 1e-3, 50, 0, 1, 2, 0)  # unknown fitting model
    assert engine_lib.lk_create(C.byref(bad), C.byref(h)) == ca.ERROR_BAD_DOMAIN
    bad = _ffi.LkConfig(2, 3, 1e-3, 50, 0, 2, 3, 0)  # (stop-start) not a multiple of step
    assert engine_lib.lk_create(C.byref(bad), C.byref(h)) == ca.ERROR_BAD_DOMAIN


def test_fails_loudly_without_a_device(engine_lib):
    if engine_lib.lk_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(ca.LkError) as ei:
        ca.HipCorrelationEngine()
    assert ei.value.code == ca.ERROR_DEVICE


def test_missing_library_is_an_import_error(tmp_path):
    with pytest.raises(ImportError):
        _ffi.load_library(str(tmp_path / "nope.so"))


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under correlation_amd/, include/, scripts/ or
    bench.py's product leg may import, link or open it."""
    for base in ("correlation_amd", "include", "scripts"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".sh")):
                    txt = open(os.path.join(dp, f)).read()
                    assert "lk_oracle" not in txt and "lko_" not in txt and "oracle/" not in txt, \
                        os.path.join(dp, f)


def test_speckle_generator_is_deterministic():
    a1, b1 = ca.speckle.speckle_pair(96, 128, seed=3)
    a2, b2 = ca.speckle.speckle_pair(96, 128, seed=3)
    assert a1.dtype == np.uint8 and a1.shape == (96, 128)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2) and not np.array_equal(a1, b1)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference headers not present on this box")
def test_cuda_class_adapter_compiles_against_reference_headers(tmp_path):
    """include/lk_cuda_class_adapter.hpp is meant to be compiled inside the reference tree
    (it uses the reference's own enums.hpp / domains.hpp): check that it does."""
    import subprocess
    src = tmp_path / "adapter_check.cpp"
    src.write_text('#include "lk_cuda_class_adapter.hpp"\n'
                   "int main() { HipCudaClass c; frame_results fr{}; float g[6] = {0};\n"
                   "  c.set_fitting_model(fm_UVUxUyVxVy); c.set_interpolation_model(im_bicubic);\n"
                   "  CorrelationResult *r = c.correlate(0, g, fr); v_points p = c.getUndXY0ToCPU(0);\n"
                   "  return (int)p.size() + (int)r->errorCode + (int)c.resetPolygon(0, 1, 2, 3, 4); }\n")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-I", "/root/reference", "-c", str(src), "-o", str(tmp_path / "a.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_cpp_example_builds_against_the_c_abi(engine_lib, tmp_path):
    """examples/track_sequence.cpp uses nothing but include/*.h and liblk_engine.so."""
    import subprocess
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "track_sequence.cpp"), "-L" + os.path.join(ROOT, "correlation_amd"),
           "-llk_engine", "-Wl,-rpath," + os.path.join(ROOT, "correlation_amd"), "-o", str(tmp_path / "track_sequence")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(tmp_path / "track_sequence")], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_host_roi_code_equals_oracle_and_reference_goldens(engine_lib, oracle):
    """The product's ROI -> sample-list code (lk_roi.hpp, behind lk_set_sector_* and
    lk_commit_sectors) through its host-only entry points: rectangular grid geometry, annular
    sectors, blob rasterisation (incl. the reference's own polygonBlob_class outputs) and the
    per-level decimation, bit for bit."""
    L = engine_lib
    # rectangular grids
    for args in ((24.0, 24.0, 2023.0, 2023.0, 100, 100), (24.0, 24.0, 2023.0, 2023.0, 224, 224), (3.5, 7.25, 490.0, 300.5, 7, 5)):
        xd, yd = C.c_int(), C.c_int()
        cen = np.zeros((args[4] * args[5], 2), np.int32)
        assert L.lk_roi_rect_grid(*args, C.byref(xd), C.byref(yd), cen.ctypes.data_as(C.POINTER(C.c_int))) == 0
        wxd, wyd, wcen = oracle.rect_sector_geometry(*args)
        assert (xd.value, yd.value) == (wxd, wyd) and np.array_equal(cen, wcen)
    # annular sectors
    for args in ((30.0, 24.0, 0.4, 0.9, 128.0, 126.0, 7), (600.0, 150.0, 1.1, 0.19634954, 2048.0, 2048.0, 32),
                 (10.0, 40.0, 0.0, 6.2831855, 64.0, 64.0, 1)):
        n = L.lk_roi_annular_points(*args, None, 0)
        got = np.zeros((max(n, 1), 2), np.float32)
        assert L.lk_roi_annular_points(*args, _ffi.fptr(got), n) == n
        want = oracle.annular_points(*args)
        assert n == len(want) and got[:n].tobytes() == want.tobytes()
    # blobs: the reference's own polygonBlob_class outputs (tests/golden/ref_blob.npz)
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_blob.npz"))
    cases = sorted({k[:-len("_contour")] for k in g.files if k.endswith("_contour")})
    assert len(cases) >= 4
    for c in cases:
        contour, count = np.ascontiguousarray(g[c + "_contour"], np.float32), int(np.asarray(g[c + "_count"]).reshape(-1)[0])
        n = L.lk_roi_blob_points(_ffi.fptr(contour), len(contour), None, 0)
        if count < 0:            # the reference rejects the contour (self-intersecting loop)
            assert n == -1, c
            continue
        want = g[c + "_pts"]
        got = np.zeros((max(n, 1), 2), np.float32)
        L.lk_roi_blob_points(_ffi.fptr(contour), len(contour), _ffi.fptr(got), n)
        assert n == count and got[:n].tobytes() == np.ascontiguousarray(want, np.float32).reshape(-1, 2)[:n].tobytes(), c
    # a blob large enough for the threaded scan fill: triangles filled in parallel, joined in clipping order
    ang = 2 * np.pi * np.arange(40) / 40
    rad = np.where(np.arange(40) % 2 == 0, 520.0, 330.0)
    star = np.stack([700.3 + rad * np.cos(ang), 650.7 + rad * np.sin(ang)], 1).astype(np.float32)
    n = L.lk_roi_blob_points(_ffi.fptr(star), len(star), None, 0)
    got = np.zeros((n, 2), np.float32)
    L.lk_roi_blob_points(_ffi.fptr(star), len(star), _ffi.fptr(got), n)
    want = oracle.blob_points(star)
    assert n == len(want) > 400000 and got.tobytes() == want.tobytes()
    # decimation
    rng = np.random.default_rng(4)
    pts = (rng.random((500, 2)) * 64).astype(np.float32)
    pts[::3] = np.round(pts[::3])
    for delta in (1, 2):
        out = np.zeros_like(pts)
        k = L.lk_roi_decimate(_ffi.fptr(pts), len(pts), delta, _ffi.fptr(out))
        want = oracle.decimate(pts, delta)
        assert k == len(want) and out[:k].tobytes() == want.tobytes()
