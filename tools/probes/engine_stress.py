#!/usr/bin/env python3
"""Create, use and destroy engines in a loop (streams, events, device buffers must all come back)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from gort_amd import api  # noqa: E402

wl = np.linspace(400.0, 2500.0, 150)
c = api.gap_probabilities(api.make_canopy(lai=4.0))
sp = api.spectra(wl)
g = api.hemisphere_grid(5, 6, 361)
lut = torch.empty((30 * 361, wl.size), dtype=torch.float64, device="cuda")
ang = np.array([[10.0, 0.0, 30.0, 20.0]] * 64)
free0 = None
for i in range(300):
    e = api.Engine()
    e.set_canopy(c)
    e.set_spectra(*sp)
    for _ in range(3):
        e.rsurf_grid_dev(g, 0, 30, lut)
    e.rsurf_stream(ang)
    e.energy_stream(ang[:2])
    e.close()
    if i == 20:
        free0 = torch.cuda.mem_get_info()[0]
free1 = torch.cuda.mem_get_info()[0]
print("free after 20 engines %.1f MiB, after 300 engines %.1f MiB, drift %.1f MiB" % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
assert free0 - free1 < (64 << 20)
print("ok")
