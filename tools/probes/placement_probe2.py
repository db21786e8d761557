#!/usr/bin/env python3
"""Is the slow/fast state a property of the allocation?  Holds several output slabs at once and times the LUT
step and a plain torch fill into each, twice round.  PROBE_N (slabs, default 4), PROBE_NSZA (default 91)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from gort_amd import api  # noqa: E402


def med(fn, n=16):
    ms = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ms.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ms[4:]))


def main():
    nsza = int(os.environ.get("PROBE_NSZA", "91"))
    nslab = int(os.environ.get("PROBE_N", "4"))
    wl = np.arange(400.0, 2501.0, 1.0)
    canopy = api.gap_probabilities(api.make_canopy(lai=4.0))
    rs, rl, tl = api.spectra(wl)
    eng = api.Engine()
    eng.set_canopy(canopy)
    eng.set_spectra(rs, rl, tl)
    grid = api.hemisphere_grid(nsza=nsza)
    rows = grid.nsza * grid.nvza
    gb = rows * grid.nphi * wl.size * 8 / 1e9
    print("xcd mapping:", eng.xcd_mapping(), flush=True)
    slabs = []
    for i in range(nslab):
        slabs.append(torch.empty((rows * grid.nphi, wl.size), dtype=torch.float64, device="cuda"))
        print("slab %d at 0x%x" % (i, slabs[-1].data_ptr()), flush=True)
    eng.rsurf_grid_dev(grid, 0, rows, slabs[0])
    eng.synchronize()
    print("xcd weights:", eng.xcd_weights(), flush=True)
    for rnd in range(int(os.environ.get("PROBE_ROUNDS", "2"))):
        for i, lut in enumerate(slabs):
            def step():
                eng.rsurf_grid_dev(grid, 0, rows, lut)
                eng.synchronize()

            def fill():
                lut.fill_(1.0)
                torch.cuda.synchronize()
            t = med(step)
            f = med(fill) if os.environ.get("PROBE_FILL", "1") != "0" else float("nan")
            print("round %d slab %d: LUT step %.3f ms (%.0f GB/s)   torch fill_ %.3f ms (%.0f GB/s)"
                  % (rnd, i, t, gb / t * 1e3, f, gb / f * 1e3), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
