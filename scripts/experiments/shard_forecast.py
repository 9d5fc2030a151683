#!/usr/bin/env python3
"""What ONE rank of an N-rank sharded sequence has to solve, measured on one GPU: the contiguous block [S/N] of the sector grid
 through frame-pipelined windows of 16 pairs (bench.py's sharded_configs.*_sequence), 64 pairs.  t(1) / t(1/N) is the
speed-up the solve alone allows at N devices - before the frame broadcast and the record all-gather.
   scripts/experiments/shard_forecast.py C4|C2|C4B [pairs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import correlation_amd as ca  # noqa: E402
from correlation_amd.workload import C2, C4, C4B, shard_range  # noqa: E402

wl = {"C2": C2, "C4": C4, "C4B": C4B}[sys.argv[1] if len(sys.argv) > 1 else "C4"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(os.environ.get("LK_K", 16))
frames = np.stack(ca.speckle.speckle_sequence(wl.size, wl.size, n + 1, velocity=(0.8, -0.4), dilation=1e-4, seed=7, device="cuda"))
c = (wl.size / 2 - 0.5, wl.size / 2 - 0.5)
zero = np.zeros(6, np.float32)
print(wl.name)
base = None
for world in [int(x) for x in os.environ.get("LK_WORLDS", "1,2,4,8").split(",")]:
    rank = world // 2          # a block from the middle of the grid
    first, count = shard_range(wl.hs * wl.vs, rank, world)
    e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
    if os.environ.get("LK_MODE") == "batch_invariant":
        e.set_batch_invariant(True)
    elif os.environ.get("LK_MODE") == "reference_order":
        e.set_reference_order(1)
    e.set_undeformed_image(frames[0])
    e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs, first, count)
    e.commit_sectors()
    e.sequence_reserve(2 * K)
    best = None
    for rep in range(3):
        ms = 0.0
        for w0 in range(0, n, K):
            k = min(K, n - w0)
            for i in range(k):
                e.sequence_set_frame((w0 // K % 2) * K + i, frames[w0 + i + 1])
            e.adjust_initial_guess(w0, True, zero, c)
            e.synchronize()
            e.correlate_sequence_async(k, first_slot=(w0 // K % 2) * K, constant_velocity=True, host_records=False)
            e.wait_sequence(False)
            ms += e.stats()["solve_ms"]
        best = ms if best is None else min(best, ms)
    base = base or best
    print(f" 1/{world} of the grid ({count} sectors, pipelined {e.sequence_is_pipelined}): {best / n:.4f} ms per pair  ->  solve-only speed-up {base / best:.2f} at {world} devices")
    e.close()
