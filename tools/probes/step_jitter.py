#!/usr/bin/env python3
"""Per-step wall time of the LUT step (synchronised after every step) for several slab sizes, to see
box-to-box and step-to-step variation of the expansion kernel.  python3 tools/step_jitter.py [nsza ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from gort_amd import api  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [91, 46, 23, 12, 91]
    wl = np.arange(400.0, 2501.0, 1.0)
    canopy = api.gap_probabilities(api.make_canopy(lai=4.0))
    rs, rl, tl = api.spectra(wl)
    eng = api.Engine()
    eng.set_canopy(canopy)
    eng.set_spectra(rs, rl, tl)
    for nsza in sizes:
        grid = api.hemisphere_grid(nsza=nsza)
        rows = grid.nsza * grid.nvza
        lut = torch.empty((rows * grid.nphi, wl.size), dtype=torch.float64, device="cuda")
        ms = []
        for _ in range(40):
            t0 = time.perf_counter()
            eng.rsurf_grid_dev(grid, 0, rows, lut)
            eng.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        gb = rows * grid.nphi * wl.size * 8 / 1e9
        a = np.array(ms[5:])
        print("nsza=%3d %.2f GB  step ms: min %.3f med %.3f max %.3f  -> %.0f GB/s at the median; first five %s"
              % (nsza, gb, a.min(), np.median(a), a.max(), gb / np.median(a) * 1e3,
                 " ".join("%.2f" % x for x in ms[:5])), flush=True)
        del lut
        torch.cuda.empty_cache()
    eng.close()


if __name__ == "__main__":
    main()
