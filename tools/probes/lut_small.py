import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gort_amd import api
c = api.gap_probabilities(api.make_canopy(lai=4.0))
eng = api.Engine(); eng.set_canopy(c)
wl = np.arange(400.0, 2501.0)
eng.set_spectra(*api.spectra(wl))
g = api.hemisphere_grid()
for rows in (182, 364, 1036, 8281):
    lut = torch.empty((rows * g.nphi, wl.size), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        eng.rsurf_grid_dev(g, 0, rows, lut)
    eng.synchronize(); eng.last_expand_ms()
    for _ in range(20):
        eng.rsurf_grid_dev(g, 0, rows, lut)
    eng.synchronize()
    ms = eng.last_expand_ms()
    b = rows * g.nphi * wl.size * 8
    print("LUT kernel rows %5d  %.3f GB  %.1f us  %.0f GB/s  weights %s" % (rows, b / 1e9, ms * 1e3, b / ms / 1e6, eng.xcd_weights()))
    del lut
