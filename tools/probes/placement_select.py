"""Would choosing among several 50 GB allocations pay at N = 1, and does the allocator's probe (the LUT kernel's bare
store pattern, gort_engine_probe_store_pattern) rank them as the kernel does?  Four allocations alive together, each:
probe rate, then the kernel's mean over 20 launches, twice round.  Usage: python tools/probes/placement_select.py"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from gort_amd import api  # noqa: E402


def main():
    wl = np.arange(400.0, 2501.0)
    eng = api.Engine()
    eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
    eng.set_spectra(*api.spectra(wl))
    grid = api.hemisphere_grid()
    rows = grid.nsza * grid.nvza
    n = rows * grid.nphi * wl.size

    def kernel_ms(buf):
        for _ in range(3):
            eng.rsurf_grid_dev(grid, 0, rows, buf.at(0))
        eng.synchronize()
        eng.last_expand_ms()
        for _ in range(20):
            eng.rsurf_grid_dev(grid, 0, rows, buf.at(0))
        eng.synchronize()
        return eng.last_expand_ms()

    first = eng.lut_alloc(n, max_draws=1)
    print("first allocation of the process: kernel %.3f ms" % kernel_ms(first), flush=True)
    first.free()
    bufs = [eng.lut_alloc(n, max_draws=1) for _ in range(4)]
    for rnd in range(2):
        for i, b in enumerate(bufs):
            g = eng.probe_store_pattern(b.at(0), b.nbytes)
            print("round %d  allocation %d (%#x): probe %6.0f GB/s   kernel %.3f ms = %.0f GB/s" % (rnd, i, b.ptr, g, kernel_ms(b), n * 8 / kernel_ms(b) / 1e6), flush=True)


if __name__ == "__main__":
    main()
