#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU (the 8-GPU form shards members): N members with their own crown
geometry, LAI and leaf/soil parameters; per member sun zenith 30 deg, view zenith 0..90, relative azimuth
0..360 (32 851 tuples) x 2101 bands.  Everything on the device: gap probabilities (one workgroup per
member), PROSPECT-D + Price (thread per member x band), LUT expansion in chunks of members that fit HBM.
Prints timings; not the headline metric (that is bench.py).
    tools/bench_ensemble.py [members [chunk [bands]]]      bands = 7: the MODIS land bands (the ensemble the reference's
README.md:8-9 names; fused node kernel, every member in one launch), else that many bands across 400..2500 nm."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gort_amd import api

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
CHUNK = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(12345)                      # SURVEY.md 8(d) C5 draw
canopies, leaf = [], []
for _ in range(N):
    hb, br, pcc, lai = rng.uniform(1, 3), rng.uniform(1, 3.5), rng.uniform(0.2, 0.8), rng.uniform(0.5, 6)
    cab, cw, cm, Nn, rsl1 = rng.uniform(10, 60), rng.uniform(0.005, 0.03), rng.uniform(0.002, 0.015), rng.uniform(1, 2.5), rng.uniform(0.05, 0.4)
    canopies.append(api.make_canopy(newstyle=(float(np.float32(hb)), float(np.float32(br)), float(np.float32(pcc))), lai=float(np.float32(lai))))
    leaf.append(api.leaf_soil(prospect=dict(N=Nn, Cab=cab, Cw=cw, Cm=cm), rsl=(rsl1, 0.1, 0.03726, -0.002426)))
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 2101
wl = np.arange(400.0, 2501.0) if NW == 2101 else (np.array([469.0, 555.0, 645.0, 858.5, 1240.0, 1640.0, 2130.0]) if NW == 7
                                                  else np.linspace(400.0, 2500.0, NW))
g = api.Grid(); g.sza0, g.dsza, g.nsza = 30.0, 1.0, 1; g.vza0, g.dvza, g.nvza = 0.0, 1.0, 91; g.phi0, g.dphi, g.nphi = 0.0, 1.0, 361
per_member = g.nvza * g.nphi * wl.size
e = api.Engine()
torch.cuda.synchronize()
t0 = time.perf_counter()
e.set_members_leaf(canopies, leaf, wl, compute_gaps=True); e.synchronize()
t_setup = time.perf_counter() - t0
lut = torch.empty((CHUNK, g.nvza * g.nphi, wl.size), dtype=torch.float64, device="cuda")
for m0 in range(0, min(N, 2 * CHUNK), CHUNK):           # warm-up
    e.rsurf_members_grid_dev(g, m0, min(N, m0 + CHUNK), lut)
e.synchronize(); e.last_expand_ms()
t0 = time.perf_counter()
for m0 in range(0, N, CHUNK):
    e.rsurf_members_grid_dev(g, m0, min(N, m0 + CHUNK), lut)
e.synchronize()
t_lut = time.perf_counter() - t0
k = e.last_expand_ms()
if k <= 0:                                               # below 128 bands there is no LUT expansion kernel to time
    print("members %d, chunk %d, %d bands: setup %.1f ms; LUT %.3f ms for %.3e samples = %.3e samples/s (%.2f GB written, %.0f GB/s)"
          % (N, CHUNK, wl.size, t_setup * 1e3, t_lut * 1e3, N * per_member, N * per_member / t_lut, N * per_member * 8 / 1e9,
             N * per_member * 8 / t_lut / 1e9))
    print("finite below the horizon:", bool(torch.isfinite(lut[:, : 90 * 361]).all()))
    sys.exit(0)
print("members %d, chunk %d: setup (H2D + gap kernel + spectra kernel + band tables) %.1f ms; LUT %.1f ms for %.3e samples"
      " = %.3e samples/s (%.1f GB written, expand kernel mean %.3f ms per chunk = %.0f GB/s)"
      % (N, CHUNK, t_setup * 1e3, t_lut * 1e3, N * per_member, N * per_member / t_lut, N * per_member * 8 / 1e9, k,
         min(CHUNK, N) * per_member * 8 / k / 1e6))
ok = bool(torch.isfinite(lut[:, : 90 * 361]).all())
print("finite below the horizon:", ok)
