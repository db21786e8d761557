import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gort_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = np.arange(400.0, 2501.0)
c = api.gap_probabilities(api.make_canopy(lai=4.0))
eng = api.Engine(); eng.set_canopy(c); eng.set_spectra(*api.spectra(wl))
rng = np.random.default_rng(0)
cases = {"91 random": rng.integers(0, 90, n).astype(float), "all distinct": rng.uniform(0, 89, n), "1 zenith": np.full(n, 30.0),
         "runs of 4": ((np.arange(n) // 4) % 91).astype(float)}
out = torch.empty((n, wl.size), dtype=torch.float64, device="cuda")
tens = {k: torch.tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), v, np.zeros(n)], 1), device="cuda") for k, v in cases.items()}
res = {(k, g): [] for k in cases for g in (0, 1)}
for rnd in range(12):
    for k in cases:
        for g in (1, 0):
            eng.set_stream_grouping(2 if g else 0)
            for _ in range(3):
                eng.rsurf_stream_dev(tens[k], out); eng.synchronize()
                res[(k, g)].append(eng.last_stream_ms() * 1e3)
for (k, g), v in res.items():
    v = np.array(v[3:])
    print("%-14s grouping=%d  stage median %6.1f us  min %6.1f  max %6.1f  (n=%d)" % (k, g, np.median(v), v.min(), v.max(), v.size))
