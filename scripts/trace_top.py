#!/usr/bin/env python3
"""The longest wavefronts of a trace (scripts/trace_solve.py): duration, steps, cycles per step and their split."""
import sys

import numpy as np

z = np.load(sys.argv[1])
t = z["trace"].astype(np.float64)
t = t[t[:, 3] > 0]
dur = (t[:, 1] - t[:, 0]) / 100.0
order = np.argsort(-dur)[: int(sys.argv[2]) if len(sys.argv) > 2 else 10]
print("  wave   dur us  steps  cycles/step   evaluation   solve     w4      w5   (w4 / w5: HW_ID / XCC_ID, or with LK_TRACE_FINE fetch / post cycles per step)")
for i in order:
    st = t[i, 3]
    print(f" {i:5d} {dur[i]:8.1f} {int(st):6d} {t[i, 6] / st:11.0f} {t[i, 2] / st:12.0f} {t[i, 7] / st:7.0f} {t[i, 4] / st:7.0f} {t[i, 5] / st:7.0f}")
st = t[:, 3]
print(f"all waves: median dur {np.median(dur):.1f} us, steps {np.median(st):.0f}, cycles/step {np.median(t[:, 6] / st):.0f}, "
      f"evaluation {np.median(t[:, 2] / st):.0f}, solve {np.median(t[:, 7] / st):.0f}")
