#!/usr/bin/env python3
"""Does the LUT step's time follow how far below the top of device memory the slab lies?  A spacer of X GiB is
allocated first (device memory is handed out top-down), then the slab; both are freed again before the next X.
PROBE_NSZA (default 91)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from gort_amd import api  # noqa: E402


def main():
    nsza = int(os.environ.get("PROBE_NSZA", "91"))
    wl = np.arange(400.0, 2501.0, 1.0)
    eng = api.Engine()
    eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
    eng.set_spectra(*api.spectra(wl))
    grid = api.hemisphere_grid(nsza=nsza)
    rows = grid.nsza * grid.nvza
    n = rows * grid.nphi * wl.size
    free, total = torch.cuda.mem_get_info()
    print("device memory: %.1f GiB free of %.1f GiB; slab %.2f GiB" % (free / 2**30, total / 2**30, 8 * n / 2**30))
    for rnd in range(int(os.environ.get("PROBE_ROUNDS", "2"))):
        xs = [float(v) for v in os.environ["PROBE_X"].split(",")] if os.environ.get("PROBE_X") else (0, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 32, 40, 48, 64, 96, 128, 160, 192)
        for x in xs:
            xb = int(x * 2**30)
            if xb + 8 * n + (4 << 30) > free:
                continue
            spacer = torch.empty(xb, dtype=torch.uint8, device="cuda") if xb else None
            lut = torch.empty(n, dtype=torch.float64, device="cuda")
            for _ in range(4):
                eng.rsurf_grid_dev(grid, 0, rows, lut)
            eng.synchronize()
            eng.last_expand_ms()
            for _ in range(10):
                eng.rsurf_grid_dev(grid, 0, rows, lut)
            eng.synchronize()
            k = eng.last_expand_ms()
            print("round %d  spacer %5.1f GiB  slab at 0x%x: kernel %.3f ms  %.0f GB/s" % (rnd, x, lut.data_ptr(), k, 8 * n / k / 1e6), flush=True)
            del lut, spacer
            torch.cuda.empty_cache()
    eng.close()


if __name__ == "__main__":
    main()
