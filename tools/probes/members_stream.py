#!/usr/bin/env python3
"""gort_rsurf_members_stream_dev - the observation operator of an ensemble filter: the same angle lines for every member -
across band counts: tools/probes/members_stream.py [MEMBERS [LINES [BANDS ...]]]
(under rocprofv3: the interpreter binary itself after `--`, see tools/bench_lines.py)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gort_amd import api
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(12345)
canopies, leaf = [], []
for _ in range(M):
    hb, br, pcc, lai = rng.uniform(1, 3), rng.uniform(1, 3.5), rng.uniform(0.2, 0.8), rng.uniform(0.5, 6)
    canopies.append(api.make_canopy(newstyle=(float(np.float32(hb)), float(np.float32(br)), float(np.float32(pcc))), lai=float(np.float32(lai))))
    leaf.append(api.leaf_soil(prospect=dict(N=rng.uniform(1, 2.5), Cab=rng.uniform(10, 60), Cw=rng.uniform(0.005, 0.03), Cm=rng.uniform(0.002, 0.015)),
                              rsl=(rng.uniform(0.05, 0.4), 0.1, 0.03726, -0.002426)))
ang = torch.as_tensor(np.stack([rng.uniform(0, 70, n), rng.uniform(0, 360, n), rng.uniform(10, 70, n), rng.uniform(0, 360, n)], 1), device="cuda")
e = api.Engine()
for nw in ([int(v) for v in sys.argv[3:]] or (7, 16, 17, 32, 100, 200, 640, 2101)):
    wl = np.linspace(400.0, 2500.0, nw)
    e.set_members_leaf(canopies, leaf, wl, compute_gaps=True); e.synchronize()
    out = torch.empty((M, n, nw), dtype=torch.float64, device="cuda")
    f = lambda: api._check(api.lib().gort_rsurf_members_stream_dev(e.h, api._ptr(ang), n, 0, M, api._ptr(out)))
    for _ in range(3):
        f(); e.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); f(); e.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print("%d members x %d lines x %4d bands: %9.1f us  %.3e samples/s  %6.0f GB/s" % (M, n, nw, t * 1e6, M * n * nw / t, M * n * nw * 8 / t / 1e9), flush=True)
