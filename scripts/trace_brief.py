#!/usr/bin/env python3
"""Short summary of a wavefront trace (scripts/trace_solve.py): durations, cycles per step, SIMD sharing."""
import sys
from collections import Counter

import numpy as np

z = np.load(sys.argv[1])
t = z["trace"]
t = t[t[:, 3] > 0]
hw, xcc = t[:, 4].astype(np.int64), t[:, 5].astype(np.int64) & 0xF
key = list(zip(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF, (hw >> 4) & 3))
c = Counter(key)
per = np.array([c[k] for k in key])
dur = (t[:, 1] - t[:, 0]) / 100.0
steps = t[:, 3].astype(np.float64)
print(f"waves {len(t)}  solve_ms {float(z['solve_ms']):.3f}  kernel span {(t[:, 1].max() - t[:, 0].min()) / 100.0:.1f} us")
print(f"SIMDs used {len(c)}; waves per SIMD {dict(Counter(c.values()))}")
for n in sorted(set(per)):
    m = per == n
    print(f" {n} wave(s) on the SIMD: {m.sum():5d} waves, median {np.median(dur[m]):7.1f} us, steps {np.median(steps[m]):.0f}, "
          f"cycles/step {np.median(t[m, 6] / steps[m]):.0f} (evaluation {np.median(t[m, 2] / steps[m]):.0f}, solve {np.median(t[m, 7] / steps[m]):.0f})")
