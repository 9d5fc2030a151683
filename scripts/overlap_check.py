#!/usr/bin/env python3
"""Ready-list overlap (LK_OVERLAP, read once per process): records of config 4 with the overlap on must equal the
records with it off in batch-invariant mode (the same fixed 16-lane arithmetic), and repeat from solve to solve.
python scripts/overlap_check.py [C4|C4B|C5]   (spawns one process per mode)"""
import os
import subprocess
import sys

import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import correlation_amd as ca
    from correlation_amd import workload
    wl = getattr(workload, sys.argv[3] if len(sys.argv) > 3 else "C4")
    und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7)
    e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
    e.set_batch_invariant(True)
    e.set_undeformed_image(und)
    e.set_deformed_image(dfm)
    e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
    e.commit_sectors()
    r1 = e.correlate_all(np.zeros(6, np.float32))
    st1 = e.stats()
    r2 = e.correlate_all(np.zeros(6, np.float32))
    st2 = e.stats()
    print(f"LK_OVERLAP={os.environ.get('LK_OVERLAP')}: repeatable {r1.tobytes() == r2.tobytes()}, solve_ms {st1['solve_ms']:.3f} {st2['solve_ms']:.3f}, "
          f"evaluations {st1['evaluations']} {st2['evaluations']}, error-free {(r1['error_code'] == 0).mean():.4f}")
    r1.tofile(sys.argv[2])
    e.sector_stats().astype(np.uint32).tofile(sys.argv[2] + ".stats")
    sys.exit(0)
out = {}
for ov in ("0", "1"):
    path = f"/tmp/overlap_{ov}.bin"
    subprocess.run([sys.executable, __file__, "child", path] + sys.argv[1:], env=dict(os.environ, LK_OVERLAP=ov), check=True)
    out[ov] = np.fromfile(path, np.uint8).reshape(-1, 48)
d = (out["0"] != out["1"]).any(1)
print("records differing between overlap off and on:", int(d.sum()), "of", len(d), "first:", np.flatnonzero(d)[:10])
