#!/usr/bin/env python3
"""Cycle split per step of a -DLK_TRACE -DLK_TRACE_FINE trace: fetch block / evaluation / after the evaluation (of which solve) / rest."""
import sys

import numpy as np

z = np.load(sys.argv[1])
t = z["trace"].astype(np.float64)
t = t[t[:, 3] > 0]
st = t[:, 3]
tot, ev, so, fe, po = (np.median(t[:, i] / st) for i in (6, 2, 7, 4, 5))
print(f"waves {len(t)}, steps median {np.median(st):.0f} max {st.max():.0f}; cycles/step {tot:.0f} = fetch {fe:.0f} + evaluation {ev:.0f} "
      f"+ after-evaluation {po:.0f} (solve {so:.0f}) + rest {tot - fe - ev - po:.0f}")
