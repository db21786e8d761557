#!/usr/bin/env python3
"""Which part of a slab is slow?  Runs the nsza=23 step (12.7 GB) into the four quarters of one 50 GB
allocation, and the plain nsza=91 step into the whole of it, for PROBE_N allocations held at once."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from gort_amd import api  # noqa: E402


def med(fn, n=14):
    ms = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ms.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ms[4:]))


def main():
    nslab = int(os.environ.get("PROBE_N", "3"))
    wl = np.arange(400.0, 2501.0, 1.0)
    canopy = api.gap_probabilities(api.make_canopy(lai=4.0))
    rs, rl, tl = api.spectra(wl)
    eng = api.Engine()
    eng.set_canopy(canopy)
    eng.set_spectra(rs, rl, tl)
    big_grid = api.hemisphere_grid(nsza=91)
    q_grid = api.hemisphere_grid(nsza=22)
    big_rows, q_rows = 91 * big_grid.nvza, 22 * q_grid.nvza
    line = big_grid.nphi * wl.size
    slabs = [torch.empty((big_rows * big_grid.nphi, wl.size), dtype=torch.float64, device="cuda") for _ in range(nslab)]
    for i, lut in enumerate(slabs):
        def whole():
            eng.rsurf_grid_dev(big_grid, 0, big_rows, lut)
            eng.synchronize()
        t = med(whole)
        msg = "slab %d at 0x%x: whole %.3f ms (%.0f GB/s); quarters:" % (i, lut.data_ptr(), t, big_rows * line * 8 / t / 1e6)
        flat = lut.view(-1)
        for q in range(4):
            part = flat[q * q_rows * line + q * 1000:][:q_rows * line].view(q_rows * q_grid.nphi, wl.size)

            def quarter():
                eng.rsurf_grid_dev(q_grid, 0, q_rows, part)
                eng.synchronize()
            tq = med(quarter)
            msg += "  %.3f ms (%.0f GB/s)" % (tq, q_rows * line * 8 / tq / 1e6)
        print(msg, flush=True)
    eng.close()


if __name__ == "__main__":
    main()
