#!/usr/bin/env python3
"""Hemisphere LUT (91 x 91 x 361, single canopy, device-resident, best of 7) across band counts: below 128 bands the geometry
kernel writes the samples itself (gort_geometry.hip, the fused forms), from 128 the LUT kernel expands records."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gort_amd import api
eng = api.Engine(); eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
g = api.hemisphere_grid(); rows = g.nsza * g.nvza
for nw in [int(x) for x in sys.argv[1:]] or (1, 2, 4, 7, 8, 9, 13, 16, 24, 32, 48, 64, 65, 100, 127, 128, 200):      # argv: band counts (one, under a profiler)
    eng.set_spectra(*api.spectra(np.linspace(400.0, 2500.0, nw)))
    lut = torch.empty((rows * g.nphi, nw), dtype=torch.float64, device="cuda")
    for _ in range(3):
        eng.rsurf_grid_dev(g, 0, rows, lut); eng.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); eng.rsurf_grid_dev(g, 0, rows, lut); eng.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print("hemisphere x %4d bands: %9.1f us  %.3e samples/s  %6.0f GB/s" % (nw, t * 1e6, rows * g.nphi * nw / t, rows * g.nphi * nw * 8 / t / 1e9), flush=True)
    del lut
