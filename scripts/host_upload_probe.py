#!/usr/bin/env python3
"""What the boundary costs when frames arrive in host memory (the reference hands over cv::Mat
pixels): lk_set_image from pageable memory, from memory the caller registered with
hipHostRegister, and lk_set_image_device for comparison.  2048^2 and 8192^2 frames."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402

hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

for size in (2048, 8192):
    rng = np.random.default_rng(1)
    frames = [rng.integers(0, 256, (size, size), dtype=np.uint8) for _ in range(4)]
    e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY, py_stop=2)
    n = 40

    def timed(fn):
        fn(0)
        e.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        e.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    pageable = timed(lambda i: e.set_image(ca.IMG_DEF, frames[i % 4]))
    for f in frames:
        assert hip.hipHostRegister(f.ctypes.data, f.nbytes, 0) == 0
    registered = timed(lambda i: e.set_image(ca.IMG_DEF, frames[i % 4]))
    t0 = time.perf_counter()
    for f in frames:
        hip.hipHostUnregister(f.ctypes.data)
        assert hip.hipHostRegister(f.ctypes.data, f.nbytes, 0) == 0
    reg_cost = (time.perf_counter() - t0) / 4 * 1e3
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), frames[0].nbytes) == 0
    hip.hipMemcpy(d, frames[0].ctypes.data, frames[0].nbytes, 1)
    device = timed(lambda i: e.set_image_device(ca.IMG_DEF, d.value, size, size))
    mb = frames[0].nbytes / 1e6
    print(f"{size}^2 ({mb:.1f} MB/frame): lk_set_image pageable {pageable:.3f} ms ({mb / pageable:.1f} GB/s), "
          f"registered {registered:.3f} ms ({mb / registered:.1f} GB/s), unregister+register {reg_cost:.3f} ms, "
          f"lk_set_image_device {device:.3f} ms")
    for f in frames:
        hip.hipHostUnregister(f.ctypes.data)
    e.close()
