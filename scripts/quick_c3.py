#!/usr/bin/env python3
"""Config 3 (4096^2 pair, 8 x 32 annular sectors + one 64-vertex blob, affine, three levels): repeated solves of one pair.
python scripts/quick_c3.py [reps]   (under `rocprofv3 --kernel-trace` + scripts/kernel_timeline.py: the launches of one solve)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import correlation_amd as ca  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
truth = (1.1, 0.6, 0.0008, 0.0004, -0.0004, 0.0012)
und, dfm = ca.speckle.speckle_pair(4096, 4096, p=truth, seed=11, device="cuda")
e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY)
e.set_reference_order(int(os.environ.get("LK_REF_ORDER", 0)))
e.set_undeformed_image(und)
e.set_deformed_image(dfm)
rs, as_, ri, ro = int(os.environ.get("LK_C3_RS", 8)), int(os.environ.get("LK_C3_AS", 32)), 600.0, 1800.0   # (LK_C3_RS x LK_C3_AS annular sectors)
dr, da = np.float32((ro - ri) / rs), np.float32(2 * np.pi) / np.float32(as_)
only = os.environ.get("LK_C3_ONLY", "")   # "annulus" / "blob": one of the two sector kinds alone
n_ann = 0
if only != "blob":
    e.set_sectors_annular(0, np.float32([[np.float32(ri + i * dr), dr, np.float32(j) * da, da, 2048.0, 2048.0]
                                         for i in range(rs) for j in range(as_)]), as_)
    n_ann = rs * as_
if only != "annulus":
    t = 2 * np.pi * np.arange(64) / 64
    rad = np.where(np.arange(64) % 2 == 0, 1500.0, 900.0)
    e.resetPolygon_blob(n_ann, np.stack([2048 + rad * np.cos(t), 2048 + rad * np.sin(t)], 1).astype(np.float32))
e.commit_sectors()
g = np.zeros(6, np.float32)
r = e.correlate_all(g)
ms = []
for _ in range(reps):
    e.correlate_all(g)
    ms.append(e.stats()["solve_ms"])
st = e.stats()
if os.environ.get("LK_TRACE_OUT"):   # a -DLK_TRACE tuning build (-D'LK_TRACE_PICK(G,S)=((G)==512)'): the per-workgroup trace of the last launch
    import ctypes as C
    lib = C.CDLL(ca.LIB_PATH)
    buf = np.zeros(8 * 16384, np.uint64)
    assert lib.lk_debug_trace(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    t = buf.reshape(-1, 8)
    n = int(t[0, 3] >> 32)
    t = t[:n].copy()
    t[:, 3] &= 0xFFFFFFFF
    np.savez_compressed(os.environ["LK_TRACE_OUT"], trace=t, solve_ms=st["solve_ms"], n_points=r["n_points"])
print(f" sectors {len(r)} samples {int(r['n_points'].sum())} largest {int(r['n_points'].max())}  solve_ms min {min(ms):.4f} median {np.median(ms):.4f}  "
      f"evals/sector {st['evaluations'] / st['sectors']:.2f}  alg GB/s {st['algorithmic_bytes'] / (np.median(ms) * 1e-3) / 1e9:.1f}  errfree {(r['error_code'] == 0).mean():.4f}")
