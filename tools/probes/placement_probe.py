#!/usr/bin/env python3
"""Does the LUT step's duration depend on WHERE the output slab lies?  Times the full metric grid into views
of one big allocation at several byte offsets, then into fresh allocations (address printed).
python3 tools/placement_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from gort_amd import api  # noqa: E402


def timed(eng, grid, rows, lut, n=25):
    ms = []
    for _ in range(n):
        t0 = time.perf_counter()
        eng.rsurf_grid_dev(grid, 0, rows, lut)
        eng.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ms[5:]))


def main():
    nsza = int(os.environ.get("PROBE_NSZA", "91"))
    wl = np.arange(400.0, 2501.0, 1.0)
    canopy = api.gap_probabilities(api.make_canopy(lai=4.0))
    rs, rl, tl = api.spectra(wl)
    eng = api.Engine()
    eng.set_canopy(canopy)
    eng.set_spectra(rs, rl, tl)
    grid = api.hemisphere_grid(nsza=nsza)
    rows = grid.nsza * grid.nvza
    n = rows * grid.nphi * wl.size
    gb = n * 8 / 1e9
    slack = 1 << 31
    big = torch.empty(n * 8 + slack, dtype=torch.uint8, device="cuda")
    print("big allocation at 0x%x (%.2f GB + 2 GiB slack)" % (big.data_ptr(), gb), flush=True)
    for off in (0, 8, 1 << 10, 1 << 12, 1 << 16, 1 << 21, 3 << 20, 1 << 24, 1 << 28, 1 << 30, (1 << 30) + (1 << 21)):
        view = big[off:off + n * 8].view(torch.float64).view(rows * grid.nphi, wl.size)
        t = timed(eng, grid, rows, view)
        print("offset %12d (0x%x): %.3f ms  %.0f GB/s" % (off, view.data_ptr(), t, gb / t * 1e3), flush=True)
    del big, view
    torch.cuda.empty_cache()
    keep = []
    for i in range(4):
        lut = torch.empty((rows * grid.nphi, wl.size), dtype=torch.float64, device="cuda")
        t = timed(eng, grid, rows, lut)
        print("fresh allocation %d at 0x%x: %.3f ms  %.0f GB/s" % (i, lut.data_ptr(), t, gb / t * 1e3), flush=True)
        if i % 2 == 0:
            keep.append(torch.empty(1 << 28, dtype=torch.uint8, device="cuda"))   # shift the next one
        del lut
        torch.cuda.empty_cache()
    eng.close()


if __name__ == "__main__":
    main()
