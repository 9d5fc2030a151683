"""Sanitizer builds of the host-only code (the reference has none, and lists
its own benign races there): the CPU oracle under ASan + UBSan (`make -C oracle asan`), and the
product's host code - csrc/lk_tracker.cpp (bookkeeping, report writer, PGM reader, frame loop with
helper threads) with csrc/lk_roi.hpp (ROI -> sample lists) - under ASan + UBSan and under
ThreadSanitizer, driven by tests/host/tracker_driver.cpp over the CPU mock of the engine.
GPU AddressSanitizer is not available on the MI355X pool; the kernels' indexing is covered by the
parity tests instead."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")


def test_oracle_under_asan_and_ubsan():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0 and "oracle selfcheck ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_tracker_and_roi_host_code_under_sanitizers(tmp_path, sanitizer):
    exe = tmp_path / "tracker_driver"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-fsanitize=" + sanitizer,
                        "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(HOST, "tracker_driver.cpp"), os.path.join(HOST, "lk_engine_mock.cpp"),
                        os.path.join(ROOT, "correlation_amd", "csrc", "lk_tracker.cpp"), "-lpthread", "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    for min_block in ("1", "64"):   # 1: every per-sector loop runs on the tracker's helper threads
        env = dict(os.environ, LK_TRACKER_MIN_BLOCK=min_block, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1")
        r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "tracker_driver ok" in r.stdout, r.stdout[-1000:] + r.stderr[-6000:]
