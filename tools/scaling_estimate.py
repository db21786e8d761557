#!/usr/bin/env python3
"""What one rank of an N-way strong-scaling run does, measured on ONE GPU: the metric grid's rows are cut with
gort_amd.shard.row_slab exactly as bench.py does, the slabs of rank 0, N/2 and N-1 are stepped back to back INTO THEIR
WINDOW OF THE GATHERABLE BUFFER (gort_lut_alloc, window = the slab: bench.py's layout), once on a plain first
allocation (max_draws 1) and once on the allocator's pick (max_draws 5, bench.py's default), and the implied efficiency t(1) / (N t(N)) is
printed for both.  No communication is involved in the timed step of bench.py, so this is the whole per-rank cost bar
the RCCL barrier.  An ESTIMATE from one GPU, not a scaling measurement."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gort_amd import api  # noqa: E402
from gort_amd.shard import gatherable_rows, row_slab  # noqa: E402


DRAWS = (1, 5)            # a plain first allocation, and bench.py's default


def main():
    steps = int(os.environ.get("STEPS", "40"))
    wl = np.arange(400.0, 2501.0, 1.0)
    canopy = api.gap_probabilities(api.make_canopy(lai=4.0))
    rs, rl, tl = api.spectra(wl)
    eng = api.Engine()
    eng.set_canopy(canopy)
    eng.set_spectra(rs, rl, tl)
    grid = api.hemisphere_grid()
    rows = grid.nsza * grid.nvza
    t1 = {}
    row_elems = grid.nphi * wl.size
    for world in (1, 2, 4, 8):
        worst = {d: 0.0 for d in DRAWS}
        for rank in sorted({0, world // 2, world - 1}):
            r0, r1 = row_slab(rank, world, rows)
            win = (r0 * row_elems, (r1 - r0) * row_elems)
            for draws in DRAWS:
                buf = eng.lut_alloc(gatherable_rows(world, rows) * row_elems, window=win, max_draws=draws)
                ptr = buf.at(win[0])
                for _ in range(5):
                    eng.rsurf_grid_dev(grid, r0, r1, ptr)
                eng.synchronize()
                eng.last_expand_ms()
                t0 = time.perf_counter()
                for _ in range(steps):
                    eng.rsurf_grid_dev(grid, r0, r1, ptr)
                eng.synchronize()
                ms = (time.perf_counter() - t0) * 1e3 / steps
                k = eng.last_expand_ms()
                gb = win[1] * 8 / 1e9
                pl = buf.placement
                pg = pl["probe_gbs"]
                print("N=%d rank %d rows [%d,%d) %.2f GB, max_draws %d: %.3f ms/step, expand kernel %.3f ms (%.0f GB/s), other %.3f ms; "
                      "%d candidates, picked #%d at %.0f GB/s (median %.0f, first %.0f), rescans %d"
                      % (world, rank, r0, r1, gb, draws, ms, k, gb / k * 1e3, ms - k, pl["draws"], pl["picked"], pg[pl["picked"]],
                         float(np.median(pg)), pg[0], pl["rescans"]), flush=True)
                worst[draws] = max(worst[draws], ms)
                buf.free()
        for draws in DRAWS:
            if world == 1:
                t1[draws] = worst[draws]
            print("  -> N=%d, max_draws %d: slowest rank %.3f ms/step, implied strong-scaling efficiency %.3f"
                  % (world, draws, worst[draws], t1[draws] / (world * worst[draws])), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
