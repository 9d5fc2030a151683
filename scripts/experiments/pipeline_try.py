#!/usr/bin/env python3
"""Two C2 pairs in flight: engines A and B on their own streams, steps alternate between them,
so the straggler tail of one solve overlaps the head of the next.  Prints ms per step for one
engine and for two."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import correlation_amd as ca  # noqa: E402
from correlation_amd.workload import C2 as wl  # noqa: E402

dev = torch.device("cuda", 0)
und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7)
d_und, d_def = torch.from_numpy(und).to(dev), torch.from_numpy(dfm).to(dev)
engines = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
    st = torch.cuda.Stream(dev)
    e.set_stream(st.cuda_stream)
    e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
    e.commit_sectors()
    e.set_timing(False)
    if os.environ.get("LK_TRY_BI") == "1":
        e.set_batch_invariant(True)
    g = torch.zeros((e.n_sectors, 6), dtype=torch.float32, device=dev)
    r = torch.zeros((e.n_sectors, 48), dtype=torch.uint8, device=dev)
    engines.append((e, st, g, r))
torch.cuda.synchronize()


def run(n_engines, steps):
    for k in range(steps):
        e, st, g, r = engines[k % n_engines]
        e.set_image_pair_device(d_und.data_ptr(), d_def.data_ptr(), wl.size, wl.size)
        e.correlate_all_device(g.data_ptr(), r.data_ptr())
    torch.cuda.synchronize()


for n in ([len(engines)] if os.environ.get("LK_TRY_ONLY") else range(1, len(engines) + 1)):
    run(n, 40)
    t0 = time.perf_counter()
    run(n, 400)
    dt = time.perf_counter() - t0
    print(f"{n} pair(s) in flight: {dt / 400 * 1e3:.4f} ms per step")
ref = engines[0][3].cpu().numpy()
for e, st, g, r in engines[1:]:
    same = np.array_equal(ref.view(ca.RESULT_DTYPE)["iterations"], r.cpu().numpy().view(ca.RESULT_DTYPE)["iterations"])
    print("iteration counts equal to engine 0:", same)
