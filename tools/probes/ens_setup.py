import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ctypes as C
import torch
from gort_amd import api
from gort_amd.ensemble import draw_c5_members
t0 = time.perf_counter(); canopies, leaf = draw_c5_members(1000); t1 = time.perf_counter()
print("draw + gort_canopy_init x1000 (host): %.1f ms" % ((t1 - t0) * 1e3))
wl = np.arange(400.0, 2501.0)
e = api.Engine()
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    arr = (api.Canopy * len(canopies))(*canopies)
    larr = (api.LeafSoil * len(leaf))(*leaf)
    t1 = time.perf_counter()
    api._check(api.lib().gort_engine_set_members_leaf(e.h, arr, larr, len(canopies), 1, api._ptr(wl), wl.size))
    t2 = time.perf_counter()
    e.synchronize()
    t3 = time.perf_counter()
    print("rep %d: ctypes arrays %.2f ms | C call %.2f ms | sync %.2f ms | total %.2f ms" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
