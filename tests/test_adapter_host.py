"""include/lk_cuda_class_adapter.hpp EXECUTED, not only compiled:
 * here (no GPU) against a CPU mock of the C-ABI (tests/host/lk_engine_mock.cpp), under
   AddressSanitizer + UBSan, in the reference manager's call order;
 * on the GPU box (-m gpu) against liblk_engine.so itself: the per-sector loop must give the
   records of the batched path, and frame 1 must move every sector from its own frame-0 record
   (the engine keeps per-sector state across the one-commit-per-sector first frame)."""
import os
import subprocess

import numpy as np
import pytest

import correlation_amd as ca

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")
REF = "/root/reference"


def ref_include():
    """the reference's own headers where they exist, else the test-only restatement of the few types"""
    return REF if os.path.isdir(REF) else os.path.join(HOST, "ref_min")


def write_frames(tmp_path, frames):
    paths = []
    for i, f in enumerate(frames):
        p = tmp_path / f"f{i}.raw"
        np.ascontiguousarray(f, np.uint8).tofile(p)
        paths.append(str(p))
    return paths


def test_adapter_against_the_mock_in_manager_order(tmp_path):
    exe = tmp_path / "adapter_mock"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-DADAPTER_DRIVER_MOCK",
                        "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I" + os.path.join(ROOT, "include"), "-I" + ref_include(),
                        os.path.join(HOST, "adapter_driver.cpp"), os.path.join(HOST, "lk_engine_mock.cpp"), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rng = np.random.default_rng(0)
    frames = write_frames(tmp_path, [rng.integers(0, 256, (128, 128), dtype=np.uint8) for _ in range(3)])
    hs, vs = 3, 2
    out = tmp_path / "out.bin"
    r = subprocess.run([str(exe)] + frames + ["128", "128", "8", "119", str(hs), str(vs), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr      # (a sanitizer report is a non-zero exit)
    lines = r.stdout.splitlines()
    S = hs * vs
    at = lines.index(f"commits {S}")          # (the driver's timing and sample lines come before it)
    j = lines[at + 1:]
    assert j[0].startswith("create interp=2 model=3 prec=0.001 iters=50 py=0/1/2")
    assert [x.split()[1] for x in j[1:4]] == ["slot=0", "slot=1", "slot=2"]
    k = 4
    for s in range(S):   # frame 0: register, commit (once per sector), solve - sector after sector
        assert j[k].startswith(f"set_rect {s} ") and j[k + 1] == f"commit S={s + 1}" and j[k + 2] == f"correlate {s} guess=(0,0)"
        k += 3
    assert j[k].split()[1] == "slot=2"        # the preload queue (CudaClass::tempQ): the queued frame goes to the next-image slot
    assert j[k + 1] == "def_from_nxt"
    k += 2
    rec = np.fromfile(out, ca.RESULT_DTYPE).reshape(2, S)
    for s in range(S):   # frame 1: every sector moves by its OWN frame-0 record and starts from it
        assert j[k].startswith(f"update {s} mode=1 solved=1 ")
        assert j[k + 1].startswith(f"correlate {s} guess=(")
        k += 2
    assert k == len(j)
    # records pass through the adapter untouched; the guess array is in/out (cuda_class.cu:289-290)
    assert (rec["iterations"] == 2).all() and (rec["error_code"] == 0).all()
    cx = rec["und_cx"][0]
    assert np.allclose(rec["p"][0][:, 0], 0.0025 * (cx - 64.0))
    assert np.allclose(rec["p"][1][:, 0], rec["p"][0][:, 0] + 0.0025 * (rec["und_cx"][1] - 64.0))


def test_adapter_with_several_devices_against_the_mock(tmp_path):
    """set_deviceCount(3): the same HipCudaClass calls go to lk_group (one engine per device, sectors in
    contiguous blocks); the records must be the one-device run's, sector for sector."""
    exe = tmp_path / "adapter_mock"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-DADAPTER_DRIVER_MOCK",
                        "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I" + os.path.join(ROOT, "include"), "-I" + ref_include(),
                        os.path.join(HOST, "adapter_driver.cpp"), os.path.join(HOST, "lk_engine_mock.cpp"), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rng = np.random.default_rng(1)
    frames = write_frames(tmp_path, [rng.integers(0, 256, (160, 160), dtype=np.uint8) for _ in range(3)])
    recs = {}
    for devices in (1, 3):
        out = tmp_path / f"out{devices}.bin"
        r = subprocess.run([str(exe)] + frames + ["160", "160", "8", "151", "5", "4", str(out), str(devices)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        recs[devices] = np.fromfile(out, ca.RESULT_DTYPE)
        if devices == 3:
            assert "group_create n=3" in r.stdout and "group_commit S=20 over 3" in r.stdout
            assert r.stdout.count("group_correlate_all S=20") == 2
    assert recs[1].tobytes() == recs[3].tobytes()


@pytest.mark.gpu
def test_adapter_on_a_group_moves_its_sectors_like_the_single_engine(tmp_path):
    """set_deviceCount(3) on the engine (three ranks on one GPU, LK_GROUP_DEVICES): correlateAll, updatePolygon
    (Lagrangian) for every sector, correlateAll again.  updatePolygon moves a sector by the OWNING ENGINE's record
    of it, so the group's solve must leave the records there (round 2's group solved into a buffer of its own:
    every sector then moved by (0, 0)).  Must equal the one-engine run of the same driver, byte for byte."""
    exe = tmp_path / "adapter_gpu"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HOST, "ref_min"),
                        os.path.join(HOST, "adapter_driver.cpp"), "-L" + os.path.join(ROOT, "correlation_amd"),
                        "-llk_engine", "-Wl,-rpath," + os.path.join(ROOT, "correlation_amd"), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    und, d1 = ca.speckle.speckle_pair(256, 256, p=(2.2, -1.4, 0.002, 0.0, 0.0, -0.001), seed=5)
    _, d2 = ca.speckle.speckle_pair(256, 256, p=(4.4, -2.8, 0.004, 0.0, 0.0, -0.002), seed=5)
    frames = write_frames(tmp_path, [und, d1, d2])
    recs = {}
    for devices in (1, 3):
        out = tmp_path / f"out{devices}.bin"
        r = subprocess.run([str(exe)] + frames + ["256", "256", "24.0", "231.0", "6", "5", str(out), str(devices)],
                           capture_output=True, text=True, env=dict(os.environ, LK_GROUP_DEVICES="0,0,0"))
        assert r.returncode == 0, r.stdout + r.stderr
        recs[devices] = np.fromfile(out, ca.RESULT_DTYPE).reshape(2, 30)
    assert recs[1].tobytes() == recs[3].tobytes()
    f0, f1 = recs[3]
    assert np.abs(f1["und_cx"] - f0["und_cx"] - np.round(f0["p"][:, 0])).max() <= 1.0 and \
        np.abs(f1["und_cx"] - f0["und_cx"]).min() >= 1.0          # the sectors did move, by their own records


@pytest.mark.gpu
def test_adapter_per_sector_loop_equals_the_batched_path(tmp_path):
    """The literal drop-in: HipCudaClass driven like managerClass drives CudaClass, on the engine."""
    exe = tmp_path / "adapter_gpu"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HOST, "ref_min"),
                        os.path.join(HOST, "adapter_driver.cpp"), "-L" + os.path.join(ROOT, "correlation_amd"),
                        "-llk_engine", "-Wl,-rpath," + os.path.join(ROOT, "correlation_amd"), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    und, d1 = ca.speckle.speckle_pair(256, 256, p=(2.2, -1.4, 0.002, 0.0, 0.0, -0.001), seed=5)
    _, d2 = ca.speckle.speckle_pair(256, 256, p=(4.4, -2.8, 0.004, 0.0, 0.0, -0.002), seed=5)
    frames = write_frames(tmp_path, [und, d1, d2])
    hs, vs, x0, x1 = 6, 5, 24.0, 231.0
    out = tmp_path / "out.bin"
    r = subprocess.run([str(exe)] + frames + ["256", "256", str(x0), str(x1), str(hs), str(vs), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    S = hs * vs
    got = np.fromfile(out, ca.RESULT_DTYPE).reshape(2, S)
    # the same two frames through the batched entry points
    e = ca.HipCorrelationEngine()
    e.set_batch_invariant(True)
    e.set_undeformed_image(und)
    e.set_deformed_image(d1)
    e.set_next_image(d2)
    e.set_rect_grid(x0, x0, x1, x1, hs, vs)
    e.commit_sectors()
    f0 = e.correlate_all(np.zeros(6, np.float32))
    assert got[0].tobytes() == f0.tobytes()
    assert (f0["error_code"] == 0).all() and np.abs(f0["p"][:, 0] - 2.2).max() < 0.4
    e.makeDefPyramidFromNxt()
    for s in range(S):
        e.update_sector(s, 1)     # Lagrangian: the sector follows its own frame-0 record
    f1 = e.correlate_all(f0["p"])
    assert got[1].tobytes() == f1.tobytes()
    assert np.abs(f1["und_cx"] - f0["und_cx"] - np.round(f0["p"][:, 0])).max() <= 1.0   # the sectors did move
    assert np.abs(f1["p"][:, 0] - 4.4).max() < 0.6
    e.close()


@pytest.mark.gpu
def test_literal_per_sector_loop_over_config2_is_linear_in_the_sector_count(tmp_path):
    """The reference's own loop, unchanged: resetPolygon(i); correlate(i) for each of config 2's 10 000 sectors on the
    first frame, updatePolygon(i); correlate(i) on the second (manager_class.cpp:304-460).  Registering sector i
    behind i committed ones appends it (no rebuild, no re-upload), a rectangle that moves by whole pixels is patched
    in place - the loop is O(S).  Records equal the batched solve's in batch-invariant mode; the wall time of the
    loops is bounded (a full commit per sector took minutes)."""
    import re
    exe = tmp_path / "adapter_gpu"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HOST, "ref_min"),
                        os.path.join(HOST, "adapter_driver.cpp"), "-L" + os.path.join(ROOT, "correlation_amd"),
                        "-llk_engine", "-Wl,-rpath," + os.path.join(ROOT, "correlation_amd"), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    from correlation_amd.workload import C2 as w
    und, d1 = ca.speckle.speckle_pair(w.size, w.size, p=w.truth, seed=7)
    _, d2 = ca.speckle.speckle_pair(w.size, w.size, p=tuple(2 * np.array(w.truth)), seed=7)
    frames = write_frames(tmp_path, [und, d1, d2])
    out = tmp_path / "out.bin"
    r = subprocess.run([str(exe)] + frames + [str(w.size), str(w.size), str(w.x_begin), str(w.x_end), str(w.hs), str(w.vs), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    S = w.hs * w.vs
    got = np.fromfile(out, ca.RESULT_DTYPE).reshape(2, S)
    e = ca.HipCorrelationEngine()
    e.set_batch_invariant(True)
    e.set_undeformed_image(und)
    e.set_deformed_image(d1)
    e.set_next_image(d2)
    e.set_rect_grid(w.x_begin, w.x_begin, w.x_end, w.x_end, w.hs, w.vs)
    e.commit_sectors()
    f0 = e.correlate_all(np.zeros(6, np.float32))
    assert got[0].tobytes() == f0.tobytes()
    e.makeDefPyramidFromNxt()
    for s in range(S):
        e.update_sector(s, 1)
    f1 = e.correlate_all(f0["p"])
    assert got[1].tobytes() == f1.tobytes()
    assert (f1["error_code"] == 0).mean() > 0.99 and np.abs(np.median(f1["p"][:, 0]) - 2.6) < 0.1
    e.close()
    ms = [float(x) for x in re.search(r"sector by sector\): ([0-9.]+) ms; frame 1 \(move \+ solve\): ([0-9.]+) ms", r.stdout).groups()]
    print("per-sector loops over 10 000 sectors:", ms, "ms")
    assert ms[0] < 30000 and ms[1] < 30000, ms       # ~0.3 ms per sector; minutes with a full commit per sector
