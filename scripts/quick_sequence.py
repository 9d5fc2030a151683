#!/usr/bin/env python3
"""Frame-pipelined window against the one-pair loop on a sequence of a named workload (no CPU baseline):
   scripts/quick_sequence.py C2|C4|C4B [frames] [repeats]
   LK_MODE=default|batch_invariant|reference_order   LK_SEQ_GRID=permille (tuning)   LK_SEQ_LOOP=0 (skip the loop)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402
from correlation_amd.workload import C2, C4, C4B, C5  # noqa: E402

wl = {"C2": C2, "C4": C4, "C4B": C4B, "C5": C5}[sys.argv[1] if len(sys.argv) > 1 else "C2"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mode = os.environ.get("LK_MODE", "default")
cache = f"/tmp/speckle_seq_{wl.size}_{n + 1}.npy"
if os.path.exists(cache):
    frames = np.load(cache)
else:
    frames = np.stack(ca.speckle.speckle_sequence(wl.size, wl.size, n + 1, velocity=(0.8, -0.4) if n > 1 else (1.3, -0.7), dilation=1e-4 if n > 1 else 5e-4, seed=7, device="cuda"))
    np.save(cache, frames)
c = (wl.size / 2 - 0.5, wl.size / 2 - 0.5)
ZERO = np.zeros(6, np.float32)


def engine():
    e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
    if mode == "batch_invariant":
        e.set_batch_invariant(True)
    elif mode == "reference_order":
        e.set_reference_order(1)
    e.set_undeformed_image(frames[0])
    e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
    e.commit_sectors()
    return e


e = engine()
e.sequence_reserve(n)
for i in range(n):
    e.sequence_set_frame(i, frames[i + 1])
e.synchronize()
ms_wall, ms_ev = [], []
for r in range(reps + 1):
    e.adjust_initial_guess(0, True, ZERO, c)
    e.synchronize()
    t0 = time.perf_counter()
    e.correlate_sequence_async(n, constant_velocity=True, host_records=False)
    e.wait_sequence(False)
    t1 = time.perf_counter()
    if r:
        ms_wall.append((t1 - t0) * 1e3)
        ms_ev.append(e.stats()["solve_ms"])
st = e.stats()
pipelined = e.sequence_is_pipelined
e.adjust_initial_guess(0, True, ZERO, c)
e.correlate_sequence_async(n, constant_velocity=True, host_records=True)
rec = e.wait_sequence()
ms = float(np.median(ms_ev))
print(f"{wl.name}\n mode {mode}, window of {n} frames, pipelined {pipelined}: {ms / n:.4f} ms per pair (events; wall {np.median(ms_wall) / n:.4f})"
      f"  pit/s {st['point_iterations'] / (ms * 1e-3):.3e}  alg GB/s {st['algorithmic_bytes'] / (ms * 1e-3) / 1e9:.1f}"
      f"  frac {st['algorithmic_bytes'] / (ms * 1e-3) / 8e12:.4f}  evals/sector/frame {st['evaluations'] / st['sectors']:.2f}"
      f"  ill {st['ill_conditioned_solves']}  errfree {(rec['error_code'] == 0).mean():.4f}")
if os.environ.get("LK_SEQ_LOOP", "1") != "0":
    a = engine()
    a.set_timing(True)
    t_loop, ev_loop, same = [], [], 0
    a.set_deformed_image(frames[1])
    if n == 1:   # (a window of one pair against the one-pair launch chain of the same mode: engine-timed solves)
        a.adjust_initial_guess(0, True, ZERO, c)
        ms1 = []
        for _ in range(6):
            got = a.correlate_all(None)
            ms1.append(a.stats()["solve_ms"])
        print(f" one-pair launch chain: solve_ms min {min(ms1[1:]):.4f} median {np.median(ms1[1:]):.4f}; identical bytes {got.tobytes() == rec[0].tobytes()}")
        a.close()
        e.close()
        sys.exit(0)
    a.set_next_image(frames[2])
    a.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        a.adjust_initial_guess(k, True, ZERO, c)
        a.correlate_all_async()
        got = a.wait_results()
        same += got.tobytes() == rec[k].tobytes()
        if k + 1 < n:
            a.makeDefPyramidFromNxt()
            if k + 2 < n:
                a.set_next_image(frames[k + 3])
    t1 = time.perf_counter()
    print(f" one pair at a time: {(t1 - t0) * 1e3 / n:.4f} ms per pair (wall, incl. uploads of the next frame and record copies); frames with identical bytes {same} of {n}")
    a.close()
e.close()
