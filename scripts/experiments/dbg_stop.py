import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import correlation_amd as ca
from correlation_amd import tracker as tk
frames = ca.speckle.speckle_sequence(320, 288, 4, velocity=(0.8, -0.45), dilation=5e-4, seed=21)
os.environ["LK_SEQ_SYNC"] = "1"
for mode in (tk.ERRMODE_CONTINUE, tk.ERRMODE_STOP_ALL):
    e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY)
    e.set_batch_invariant(True)
    t = tk.SequenceTracker(ca.FM_UVUXUYVXVY, tk.DOMAIN_RECT, tk.DEF_EULERIAN, tk.REF_FIRST, mode, [0.4, -0.2, 1e-3, 5e-4, -5e-4, 2e-3], lib=e.lib)
    t.set_rect_domain(30.0, 34.0, 289.0, 251.0, 160.0, 144.0, 9, 7)
    done = tk.run_sequence(e, t, frames, ["a", "b", "c", "d"])
    r = t.results()
    print(mode, "done", done, "error_status", r["error_status"][:12], "codes", np.unique(r["error_code"], return_counts=True))
