#!/usr/bin/env python3
"""What the sun-direction table of an `-energy` stream costs (gort_energy.hip: key, rep, scan, place, index kernels): the indexed
form on N lines x 16 bands, at most 64 rows evaluated, for streams of 1, 91, 4096 and N distinct sun directions.
tools/probes/energy_table_cost.py [LINES]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gort_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nw = 16
eng = api.Engine(); eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0))); eng.set_spectra(*api.spectra(np.linspace(400.0, 2500.0, nw)))
rng = np.random.default_rng(0)
rows = torch.empty((64, nw, 3), dtype=torch.float64, device="cuda")
idx = torch.empty((n,), dtype=torch.int32, device="cuda")
cnt = torch.zeros((1,), dtype=torch.int32, device="cuda")
for name, sza in (("1 sun direction", np.full(n, 30.0)), ("91", rng.integers(0, 91, n).astype(float)),
                  ("91 in runs of 4096", ((np.arange(n) // 4096) % 91).astype(float)), ("4096", rng.integers(0, 4096, n) * (89.0 / 4096)),
                  ("every line its own", rng.uniform(0, 89, n))):
    a = torch.as_tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), sza, np.zeros(n)], 1), device="cuda")
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); eng.energy_stream_indexed_dev(a, rows, idx, cnt); eng.synchronize(); ts.append(time.perf_counter() - t0)
    print("%-20s %8d lines: %9.1f us per call, %d rows" % (name, n, float(np.median(ts[2:])) * 1e6, int(cnt.item())), flush=True)
