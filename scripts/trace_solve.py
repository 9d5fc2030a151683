#!/usr/bin/env python3
"""Per-wavefront timeline of one C2 solve launch (tuning build with -DLK_TRACE, see tune_build.sh):
LK_ENGINE_LIB=build/tune/liblk_trace.so python scripts/trace_solve.py [out.npz]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402
from correlation_amd import workload  # noqa: E402

wl = getattr(workload, os.environ.get("LK_TRACE_CONFIG", "C2"))   # (C4 with -D'LK_TRACE_PICK(G,S)=(G==1)': the one-lane kernel)

und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7)
e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
e.set_reference_order(int(os.environ.get("LK_REF_ORDER", 0)))   # (with -D'LK_TRACE_PICK(G,S)=((S)&&(G)==16)': the reference-order instance)
e.set_undeformed_image(und)
e.set_deformed_image(dfm)
grid = int(os.environ.get("LK_GRID", 0))   # another number of sectors of the same size (e.g. 32: wavefronts alone on their SIMDs)
if grid:
    pitch = (wl.x_end - wl.x_begin + 1) / wl.hs
    e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_begin + grid * pitch - 1, wl.x_begin + grid * pitch - 1, grid, grid)
else:
    e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
e.commit_sectors()
g = np.zeros(6, np.float32)
for _ in range(3):
    r = e.correlate_all(g)
st = e.stats()
lib = C.CDLL(ca.LIB_PATH)
buf = np.zeros(8 * 16384, np.uint64)
assert lib.lk_debug_trace(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
t = buf.reshape(-1, 8)
n = int(t[0, 3] >> 32)
t = t[:n].copy()
t[:, 3] &= 0xFFFFFFFF   # (steps; the upper half carried the grid size)
np.savez_compressed(sys.argv[1] if len(sys.argv) > 1 else "/tmp/trace.npz", trace=t, solve_ms=st["solve_ms"])
print("saved; analyse with scripts/trace_report.py")
