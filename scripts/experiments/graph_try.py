import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import correlation_amd as ca
from correlation_amd.workload import C2 as wl
und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7)
dev = torch.device("cuda", 0)
d_und, d_def = torch.from_numpy(und).to(dev), torch.from_numpy(dfm).to(dev)
e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
e.set_stream(stream.cuda_stream)
e.set_timing(False)
e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
e.commit_sectors()
S = e.n_sectors
d_guess = torch.zeros((S, 6), dtype=torch.float32, device=dev)
d_res = torch.empty((S, 48), dtype=torch.uint8, device=dev)
def step():
    e.set_image_device(ca.IMG_UND, d_und.data_ptr(), wl.size, wl.size)
    e.set_image_device(ca.IMG_DEF, d_def.data_ptr(), wl.size, wl.size)
    e.correlate_all_device(d_guess.data_ptr(), d_res.data_ptr())
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize()
print("direct ms/step", (time.perf_counter() - t0) * 10)
ref = d_res.clone()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=stream):
        step()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): g.replay()
    torch.cuda.synchronize()
    print("graph ms/step", (time.perf_counter() - t0) * 10, "same results", bool((d_res == ref).all()))
except Exception as ex:
    print("capture failed:", repr(ex))
