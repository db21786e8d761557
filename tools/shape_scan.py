#!/usr/bin/env python3
"""The stream expansion across band counts: which kernel a (lines, bands) shape takes and what it delivers - the fused launch
(<= 16 bands), the fused line kernel (17 ... 255 bands), the flat-panel kernel (>= 256), the narrow kernels below their thresholds.
Run from the repo root on a GPU box; profiles/r04/shape_scan.log (round 3: profiles/r03/stream_mid_bands.log, without the 0.25 s of load in front of every shape that round 4 added: its numbers read ~20 % high)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from gort_amd import api
c = api.gap_probabilities(api.make_canopy(lai=4.0))
eng = api.Engine(); eng.set_canopy(c)
rng = np.random.default_rng(0)
# the geometry stage alone (the viewed proportions K of a million lines, no band): what a call costs before its first sample
n = 1000000
a = torch.tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.integers(0, 90, n).astype(float), np.zeros(n)], 1), device="cuda")
Kt = torch.empty((n, 4), dtype=torch.float64, device="cuda")
t_up = time.perf_counter()
while time.perf_counter() - t_up < 0.25:
    eng.rsurf_stream_dev(a, None, K_t=Kt); eng.synchronize()
ts = []
for _ in range(15):
    t0 = time.perf_counter(); eng.rsurf_stream_dev(a, None, K_t=Kt); eng.synchronize(); ts.append(time.perf_counter() - t0)
geom = float(np.median(ts))
print("geometry stage alone, %d lines (K only): call %.1f us" % (n, geom * 1e6), flush=True)
for n, nw in ((1000000, 1), (1000000, 2), (1000000, 4), (1000000, 7), (1000000, 9), (1000000, 13), (1000000, 16), (1000000, 17), (1000000, 32), (1000000, 64), (1000000, 65), (1000000, 80), (1000000, 96), (1000000, 100), (1000000, 127), (1000000, 128), (1000000, 129), (1000000, 160), (1000000, 190), (1000000, 255), (1000000, 256), (500000, 300),
              (100000, 300), (1000, 2101), (1900, 2101), (2000, 2101), (10000, 1000), (4000000, 32)):
    wl = np.linspace(400.0, 2500.0, nw)
    eng.set_spectra(*api.spectra(wl))
    a = torch.tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.integers(0, 90, n).astype(float), np.zeros(n)], 1), device="cuda")
    out = torch.empty((n, nw), dtype=torch.float64, device="cuda")
    t_up = time.perf_counter()                 # clocks up: the device needs tens of milliseconds of load (a cold first shape reads 25 % slow)
    while time.perf_counter() - t_up < 0.25:
        eng.rsurf_stream_dev(a, out)
        eng.synchronize()
    ex, wall = [], []
    eng.time_streams(False)                    # the call as a user has it: no events around the expansion stage
    for _ in range(15):
        t0 = time.perf_counter(); eng.rsurf_stream_dev(a, out); eng.synchronize(); wall.append(time.perf_counter() - t0)
    eng.time_streams(True)                     # the stage alone, by the engine's events (they cost the call 6 us)
    for _ in range(15):
        eng.rsurf_stream_dev(a, out); eng.synchronize(); ex.append(eng.last_stream_ms() * 1e-3)
    e, w = float(np.median(ex)), float(np.median(wall))
    b = n * nw * 8 + n * 32
    # expansion-equivalent: the samples' bytes over what the call takes beyond the geometry stage of as many lines
    eq = n * nw * 8 / max(w - geom * n / 1000000, 1e-9) / 8e12
    print("%8d lines x %4d bands (%s): expansion stage %8.1f us, call %8.1f us = %.3e samples/s, %5.0f GB/s (%.3f of 8 TB/s; expansion-equivalent %.3f)"
          % (n, nw, eng.stream_form(), e * 1e6, w * 1e6, n * nw / w, b / w / 1e9, b / w / 8e12, eq), flush=True)
