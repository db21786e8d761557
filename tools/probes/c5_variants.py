#!/usr/bin/env python3
"""BASELINE config 5 on one GPU, what each piece of its cycle costs and what overlaps: tools/probes/c5_variants.py [MEMBERS]
   variants: members marshalled inside / before the clock; the albedo table behind the LUT chunks / queued first on a stream of
   its own; the expansion timers on / off; chunks of 25 / 50 / 100 members."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gort_amd import api
from gort_amd.ensemble import c5_grid, draw_c5_members
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
wl = np.arange(400.0, 2501.0)
canopies, leaf = draw_c5_members(N)
g = c5_grid()
per_member = g.nvza * g.nphi * wl.size
sun = torch.tensor([[0.0, 0.0, 30.0, 0.0]], dtype=torch.float64, device="cuda")


eng = api.Engine()
eng.reserve_members(N, wl.size)
lut = eng.lut_alloc(250 * per_member, max_draws=3)          # ONE buffer for every variant: where it lies decides up to 10 %
energy = torch.empty((N, 1, wl.size, 3), dtype=torch.float64, device="cuda")
arrs = api.member_arrays(canopies, leaf)


def cycle(prebuilt, beside, timers, chunk):
    eng.time_expand(timers)
    eng.energy_beside_grids(beside)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if prebuilt:
        eng.set_members_leaf(arrs[0], arrs[1], wl, compute_gaps=True)
    else:
        eng.set_members_leaf(canopies, leaf, wl, compute_gaps=True)
    t_call = time.perf_counter() - t0
    eng.synchronize()
    t_setup = time.perf_counter() - t0
    if beside:
        eng.energy_members_dev(sun, 0, N, energy)
    t1 = time.perf_counter()
    for a in range(0, N, chunk):
        eng.rsurf_members_grid_dev(g, a, min(N, a + chunk), lut)
    if not beside:
        eng.synchronize()
        t_lut = time.perf_counter() - t1
        eng.energy_members_dev(sun, 0, N, energy)
    eng.synchronize()
    t_total = time.perf_counter() - t0
    if beside:
        t_lut = time.perf_counter() - t1
    eng.last_expand_ms()
    return (t_total, t_call, t_setup, t_lut), float(energy[:, 0, 700, 0].sum())


print("%d members x hemisphere at one sun zenith x 2101 bands + albedo table, one engine, one LUT buffer; ms: total | setter call | setup synchronised | LUT chunks (+ table where beside)" % N)
variants = ((0, 0, 1, 25), (1, 0, 1, 25), (1, 0, 1, 50), (1, 0, 1, 100), (1, 0, 1, 125), (1, 0, 1, 200), (1, 0, 1, 250), (1, 0, 0, 100), (1, 1, 1, 100))
cycle(*variants[0])
ref = None
for rnd in range(3):
    for v in variants:
        (t_total, t_call, t_setup, t_lut), chk = cycle(*v)
        ref = chk if ref is None else ref
        prebuilt, beside, timers, chunk = v
        print("arrays %s, table %s, timers %s, chunk %3d: %7.2f | %5.2f | %5.2f | %6.2f   %.3e samples/s   table %s" %
              ("prebuilt" if prebuilt else "marshal.", "beside" if beside else "behind", "on " if timers else "off", chunk, t_total * 1e3, t_call * 1e3,
               t_setup * 1e3, t_lut * 1e3, N * per_member / t_total, "same" if chk == ref else "DIFFERS"), flush=True)
