#!/usr/bin/env python3
"""where the per-pair time of the host-frame loop goes: synchronous solve, async solve + wait, with / without the prefetch"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import correlation_amd as ca  # noqa: E402
from correlation_amd.workload import C2 as wl  # noqa: E402

und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7)
e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
e.set_undeformed_image(und)
e.set_deformed_image(dfm)
e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
e.commit_sectors()
g0 = np.zeros(6, np.float32)
e.adjust_initial_guess(0, False, g0, (1023.5, 1023.5))
nxt = np.ascontiguousarray(dfm).copy()
print("pin rc", e.lib.lk_pin_host_memory(C.c_void_p(nxt.ctypes.data), C.c_size_t(nxt.nbytes)))


def t(label, fn, n=40):
    for _ in range(3):
        fn()
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    e.synchronize()
    print(f"{label:60s} {(time.perf_counter() - t0) / n * 1e3:.4f} ms")


t("correlate_all(g0) [guesses up, solve, records down, sync]", lambda: e.correlate_all(g0))
t("correlate_all(None)", lambda: e.correlate_all(None))
t("correlate_all_async + wait_results", lambda: (e.correlate_all_async(), e.wait_results()))
t("set_next_image(pinned) alone + synchronize", lambda: (e.set_next_image(nxt), e.synchronize()))
t("set_next_image(pageable) alone", lambda: e.set_next_image(dfm))
e.set_next_image(nxt)
t("correlate_all(None); rotate; set_next(pinned)", lambda: (e.correlate_all(None), e.makeDefPyramidFromNxt(), e.set_next_image(nxt)))
t("async; wait; rotate; set_next(pinned)", lambda: (e.correlate_all_async(), e.wait_results(), e.makeDefPyramidFromNxt(), e.set_next_image(nxt)))
t("set_next(pinned); async; wait; rotate", lambda: (e.set_next_image(nxt), e.correlate_all_async(), e.wait_results(), e.makeDefPyramidFromNxt()))
t("set_next(pageable); correlate_all(None); rotate", lambda: (e.set_next_image(dfm), e.correlate_all(None), e.makeDefPyramidFromNxt()))

# the bench's own order of things
ee = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
ee.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
ee.commit_sectors()


def pair(two):
    if two:
        ee.set_undeformed_image(und)
    ee.set_deformed_image(dfm)
    return ee.correlate_all(g0)


for two in (True, False):
    for _ in range(3):
        pair(two)
    t0 = time.perf_counter()
    for _ in range(20):
        pair(two)
    print("pair", two, (time.perf_counter() - t0) / 20 * 1e3)
nxt2 = np.ascontiguousarray(dfm).copy()
print("pin rc", ee.lib.lk_pin_host_memory(C.c_void_p(nxt2.ctypes.data), C.c_size_t(nxt2.nbytes)))
ee.set_deformed_image(dfm)
ee.adjust_initial_guess(0, False, g0, (1024.0, 1024.0))


def prefetched(n):
    ee.set_next_image(nxt2)
    for _ in range(n):
        ee.correlate_all_async()
        ee.wait_results()
        ee.makeDefPyramidFromNxt()
        ee.set_next_image(nxt2)
    ee.synchronize()


prefetched(3)
for rep in range(3):
    t0 = time.perf_counter()
    prefetched(20)
    print("prefetched", (time.perf_counter() - t0) / 20 * 1e3)
