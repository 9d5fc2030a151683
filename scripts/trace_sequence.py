#!/usr/bin/env python3
"""Per-wavefront account of ONE frame-pipelined window (tuning build with -DLK_TRACE, scripts/tune_build.sh):
   LK_ENGINE_LIB=build/tune/liblk_trace_seq.so python scripts/trace_sequence.py C2|C4 [frames]
   (-D'LK_TRACE_PICK(G,S)=((G)==32&&!(S))' picks config 2's default instance, '((S)&&(G)==16)' the 16-lane SAFE / reference-order ones)
Every wavefront of the persistent grid leaves: start, end (100 MHz device clock), shader cycles inside the evaluations,
steps, lane groups without work summed over its steps (waiting for their sector's previous frame, or out of tickets),
rounds in which the whole wavefront slept."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402
from correlation_amd.workload import C2, C4, C4B  # noqa: E402

wl = {"C2": C2, "C4": C4, "C4B": C4B}[sys.argv[1] if len(sys.argv) > 1 else "C2"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mode = os.environ.get("LK_MODE", "default")
cache = f"/tmp/speckle_seq_{wl.size}_{n + 1}.npy"
if os.path.exists(cache):
    frames = np.load(cache)
else:
    frames = np.stack(ca.speckle.speckle_sequence(wl.size, wl.size, n + 1, velocity=(0.8, -0.4), dilation=1e-4, seed=7, device="cuda"))
    np.save(cache, frames)
e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
if mode == "batch_invariant":
    e.set_batch_invariant(True)
elif mode == "reference_order":
    e.set_reference_order(1)
e.set_undeformed_image(frames[0])
e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
e.commit_sectors()
e.sequence_reserve(n)
for i in range(n):
    e.sequence_set_frame(i, frames[i + 1])
c = (wl.size / 2 - 0.5, wl.size / 2 - 0.5)
for _ in range(2):
    e.adjust_initial_guess(0, True, np.zeros(6, np.float32), c)
    e.correlate_sequence(n, host_records=False)
st = e.stats()
lib = C.CDLL(ca.LIB_PATH)
buf = np.zeros(8 * 16384, np.uint64)
assert lib.lk_debug_trace(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
t = buf.reshape(-1, 8)
W = int(t[0, 3] >> 32)
t = t[:W].astype(np.float64)
t[:, 3] = (buf.reshape(-1, 8)[:W, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64)
start, end = (t[:, 0] - t[:, 0].min()) / 100.0, (t[:, 1] - t[:, 0].min()) / 100.0      # us
steps, idle, sleeps, cycles, ev = t[:, 3], t[:, 4], t[:, 5], t[:, 6], t[:, 2]
groups = 64 // {"C2": 32}.get(sys.argv[1] if len(sys.argv) > 1 else "C2", 16) if mode == "default" else 4
groups = int(os.environ.get("LK_TRACE_GROUPS", groups))
span = end.max()
print(f"{wl.name}\nmode {mode}: window of {n} pairs, solve_ms {st['solve_ms']:.3f} ({st['solve_ms'] / n:.4f} per pair), "
      f"{W} wavefronts of {groups} lane groups, kernel span {span:.0f} us")
print(f"wavefront life: start p99 {np.percentile(start, 99):.1f} us, end min {end.min():.0f} p1 {np.percentile(end, 1):.0f} median {np.median(end):.0f} "
      f"max {end.max():.0f} us  ->  every wavefront is live for {np.median(end - start) / span:.3f} of the window (median)")
print(f"steps per wavefront: median {np.median(steps):.0f} (min {steps.min():.0f}, max {steps.max():.0f});  shader cycles per step: median {np.median(cycles / np.maximum(steps, 1)):.0f};  "
      f"share of cycles inside the evaluations {ev.sum() / cycles.sum():.3f}")
print(f"lane groups without work, summed over all steps: {idle.sum() / (steps.sum() * groups):.4f} of the group-steps "
      f"(waiting for the sector's previous frame, or - at the very end - out of tickets)")
print(f"rounds in which a whole wavefront slept (every group waiting): {sleeps.sum():.0f} in total, {sleeps.sum() / W:.1f} per wavefront, "
      f"max {sleeps.max():.0f};  clock {np.median(cycles / ((end - start) * 1e-6)) / 1e9:.2f} GHz")
k = 12
print(f"\n   t us   wavefronts live   (the one-pair launch of the same grid: all slots busy to 80 us, <= 2 per SIMD from 140 us, one from 175 us of 240 - profiles/r03_wave_timeline.txt)")
for x in np.linspace(0, span, k, endpoint=False):
    print(f"{x:8.0f}   {int(((start <= x) & (end > x)).sum()):6d}")
