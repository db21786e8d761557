"""Is the spread of the headline kernel between allocations of one process (bench.py: first_draw 6.72 ms, record 7.12 ms
on one box) a property of WHERE the 50 GB lie or of WHEN they are written?  The same 23 launches on: a first allocation
three times over, a second allocation made after freeing the first, the two after that alive together, alternating.
Usage: python tools/probes/placement_vs_time.py [rounds]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from gort_amd import api  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    wl = np.arange(400.0, 2501.0)
    eng = api.Engine()
    eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
    eng.set_spectra(*api.spectra(wl))
    grid = api.hemisphere_grid()
    rows = grid.nsza * grid.nvza
    n = rows * grid.nphi * wl.size
    t_start = time.perf_counter()

    def run(buf, tag):
        for _ in range(3):
            eng.rsurf_grid_dev(grid, 0, rows, buf.at(0))
        eng.synchronize()
        eng.last_expand_ms()
        for _ in range(20):
            eng.rsurf_grid_dev(grid, 0, rows, buf.at(0))
        eng.synchronize()
        print("t=%6.2f s  %-28s kernel %.3f ms  (ptr %#x)" % (time.perf_counter() - t_start, tag, eng.last_expand_ms(), buf.ptr), flush=True)

    a = eng.lut_alloc(n, max_draws=1)
    for i in range(rounds):
        run(a, "A (first allocation) #%d" % i)
    a.free()
    b = eng.lut_alloc(n, max_draws=1)
    for i in range(rounds):
        run(b, "B (after A was freed) #%d" % i)
    c = eng.lut_alloc(n, max_draws=1)
    d = eng.lut_alloc(n, max_draws=1)
    for i in range(rounds):
        run(c, "C (beside B) #%d" % i)
        run(d, "D (beside B, C) #%d" % i)
        run(b, "B again #%d" % i)
    time.sleep(5.0)
    run(b, "B after 5 s idle")
    run(c, "C")
    run(d, "D")
    # does the allocator's probe (the kernel's bare store pattern) rank the allocations as the kernel does?
    e = eng.lut_alloc(n, max_draws=1)
    for name, buf in (("B", b), ("C", c), ("D", d), ("E", e)):
        g = eng.probe_store_pattern(buf.at(0), buf.nbytes)
        run(buf, "%s: pattern probe %.0f GB/s" % (name, g))


if __name__ == "__main__":
    main()
