#!/usr/bin/env python3
"""Torch-free timing of the LUT step (tuning aid): the slab is allocated with gort_dev_malloc
(ROCm runtime of /opt/rocm, not the one bundled with the torch wheel)."""
import os, sys, time
sys.modules["torch"] = None          # make `import torch` fail: keep torch's HIP runtime out of this process
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gort_amd import api

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
wl = np.arange(400.0, 2501.0)
c = api.gap_probabilities(api.make_canopy(lai=4.0))
eng = api.Engine(); eng.set_canopy(c); eng.set_spectra(*api.spectra(wl))
g = api.hemisphere_grid(); rows = g.nsza * g.nvza
n = rows * g.nphi * wl.size
lut = api.DeviceBuffer(n * 8)
for _ in range(2): eng.rsurf_grid_dev(g, 0, rows, lut)
eng.synchronize(); eng.last_expand_ms()
t0 = time.perf_counter()
for _ in range(steps): eng.rsurf_grid_dev(g, 0, rows, lut)
eng.synchronize(); dt = (time.perf_counter() - t0) / steps
k = eng.last_expand_ms()
print("native: %.3f ms/step, expand kernel %.3f ms = %.1f GB/s, %.3e samples/s" % (dt * 1e3, k, n * 8 / k / 1e6, n / dt))
