#!/usr/bin/env python3
"""Entry points at shapes nobody benchmarks: -energy streams of few bands, streams with component spectra (-prnspec) and with the
proportions (-prnprop); a million lines, device-resident, best of 5.   tools/probes/odd_shapes.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gort_amd import api
n = 1000000
rng = np.random.default_rng(0)
e = api.Engine(); e.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
def best(f, reps=5):
    for _ in range(2):
        f(); e.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); e.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)
a91 = torch.as_tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.integers(0, 91, n).astype(float), np.zeros(n)], 1), device="cuda")
aown = torch.as_tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), rng.uniform(0, 360, n)], 1), device="cuda")
for nw in (7, 100):
    e.set_spectra(*api.spectra(np.linspace(400.0, 2500.0, nw)))
    en = torch.empty((n, nw, 3), dtype=torch.float64, device="cuda")
    t = best(lambda: e.energy_stream_dev(a91, en))
    print("-energy, 1M lines x %3d bands, 91 sun directions, dense: %9.1f us  (%6.0f GB/s written)" % (nw, t * 1e6, n * nw * 24 / t / 1e9), flush=True)
    if nw == 7:
        t = best(lambda: e.energy_stream_dev(aown, en), reps=3)
        print("-energy, 1M lines x %3d bands, every line its own sun:    %9.1f us" % (nw, t * 1e6), flush=True)
    del en
    out = torch.empty((n, nw), dtype=torch.float64, device="cuda")
    sc = torch.empty((n, nw, 4), dtype=torch.float64, device="cuda")
    K = torch.empty((n, 4), dtype=torch.float64, device="cuda")
    t0 = best(lambda: e.rsurf_stream_dev(aown, out))
    t1 = best(lambda: e.rsurf_stream_dev(aown, out, None, K))
    t2 = best(lambda: e.rsurf_stream_dev(aown, out, sc, K))
    print("stream, 1M lines x %3d bands: %8.1f us; with the proportions %8.1f us; with component spectra too %8.1f us (%5.0f GB/s written)"
          % (nw, t0 * 1e6, t1 * 1e6, t2 * 1e6, n * nw * 40 / t2 / 1e9), flush=True)
    del out, sc, K
