#!/usr/bin/env python3
"""What one rank of an N-way strong-scaling run does, measured on ONE GPU: the metric grid's rows are cut with
gort_amd.shard.row_slab exactly as bench.py does, rank 0's and the last rank's slabs are stepped back to back
(no per-step sync, as in bench.py) and the implied efficiency t(1) / (N t(N)) is printed.  No communication is
involved in the timed step of bench.py, so this is the whole per-rank cost bar the RCCL barrier."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gort_amd import api  # noqa: E402
from gort_amd.shard import pick_fastest_slab, row_slab  # noqa: E402


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
    t1 = None
    for world in (1, 2, 4, 8):
        worst = 0.0
        for rank in sorted({0, world // 2, world - 1}):
            r0, r1 = row_slab(rank, world, rows)
            cands = int(os.environ.get("CANDIDATES", "3"))
            lut, cand_ms = pick_fastest_slab(eng, grid, r0, r1, wl.size, candidates=cands)
            for _ in range(5):
                eng.rsurf_grid_dev(grid, r0, r1, lut)
            eng.synchronize()
            eng.last_expand_ms()
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.rsurf_grid_dev(grid, r0, r1, lut)
            eng.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / steps
            k = eng.last_expand_ms()
            gb = (r1 - r0) * grid.nphi * wl.size * 8 / 1e9
            print("N=%d rank %d rows [%d,%d) %.2f GB: %.3f ms/step, expand kernel %.3f ms (%.0f GB/s), other %.3f ms; candidates %s"
                  % (world, rank, r0, r1, gb, ms, k, gb / k * 1e3, ms - k, " | ".join("%s -> %.3f" % (" ".join("%.3f" % x for x in a["probe_ms"]), a["verified_ms"]) for a in cand_ms)), flush=True)
            worst = max(worst, ms)
            del lut
            torch.cuda.empty_cache()
        if world == 1:
            t1 = worst
        print("  -> N=%d: slowest rank %.3f ms/step, implied strong-scaling efficiency %.3f" % (world, worst, t1 / (world * worst)),
              flush=True)
    eng.close()


if __name__ == "__main__":
    main()
