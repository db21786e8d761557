#!/usr/bin/env python3
"""XCD duty weights A/B on the SAME physical slabs: several output slabs held at once, every weight setting
timed on each of them (median of synchronised steps), plus what the engine's own calibration picks there.
PROBE_N slabs (default 4), PROBE_NSZA (default 91)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from gort_amd import api  # noqa: E402


def med(fn, n=12):
    ms = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ms.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ms[3:]))


def main():
    nsza = int(os.environ.get("PROBE_NSZA", "91"))
    nslab = int(os.environ.get("PROBE_N", "4"))
    wl = np.arange(400.0, 2501.0, 1.0)
    canopy = api.gap_probabilities(api.make_canopy(lai=4.0))
    eng = api.Engine()
    eng.set_canopy(canopy)
    eng.set_spectra(*api.spectra(wl))
    grid = api.hemisphere_grid(nsza=nsza)
    rows = grid.nsza * grid.nvza
    gb = rows * grid.nphi * wl.size * 8 / 1e9
    slabs = [torch.empty((rows * grid.nphi, wl.size), dtype=torch.float64, device="cuda") for _ in range(nslab)]
    settings = [("equal", [32] * 8)] + [("even 32 odd %d" % o, [32, o] * 4) for o in (30, 28, 27, 26, 25, 24)] + \
               [("even 28 odd 32", [28, 32] * 4)]
    print("%.2f GB slabs; step = geometry + sun table + expansion, synchronised; ms (GB/s)" % gb)
    for i, lut in enumerate(slabs):
        def step():
            eng.rsurf_grid_dev(grid, 0, rows, lut)
            eng.synchronize()
        line = "slab %d:" % i
        for name, w in settings:
            eng.set_xcd_weights(w)
            line += "  %s %.2f" % (name, med(step))
        eng.set_xcd_weights(None)
        step()
        w, _ = eng.xcd_weights()
        line += "  | calibrated %s %.2f" % (w, med(step))
        print(line, flush=True)
    eng.close()


if __name__ == "__main__":
    main()
