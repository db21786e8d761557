#!/usr/bin/env python3
"""`-energy` on a stream (gortt.c:321-325): N random lines with 91 distinct sun zeniths x 2101 bands, device-resident,
with the rows of equal sun directions shared (the default) and with every line evaluated (GORT_ENERGY_DEDUP=0, a switch
of the measuring build gort_amd/libgort_amd_ab.so, which this tool loads).
Prints ms per call and the output rate (24 B per (line, band): albedo, vegetation and soil absorption)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
# GORT_ENERGY_DEDUP is an A/B switch: it exists in the measuring build only (python -m gort_amd.build --ab)
os.environ.setdefault("GORT_AMD_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gort_amd", "libgort_amd_ab.so"))
import torch
from gort_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 2101
wl = np.arange(400.0, 2501.0) if nw == 2101 else np.linspace(400.0, 2500.0, nw)
rng = np.random.default_rng(0)
c = api.gap_probabilities(api.make_canopy(lai=4.0))
for name, sza in (("91 sun zeniths", rng.integers(0, 91, n).astype(float)), ("every line its own", rng.uniform(0, 89, n))):
    ang = torch.as_tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), sza, np.zeros(n)], 1), device="cuda")
    out = torch.empty((n, nw, 3), dtype=torch.float64, device="cuda")
    for dedup in ("1", "0"):
        if dedup == "0" and n > 200000 and name != "91 sun zeniths":
            continue                                       # the same work as the case above
        os.environ["GORT_ENERGY_DEDUP"] = dedup
        eng = api.Engine(); eng.set_canopy(c); eng.set_spectra(*api.spectra(wl))
        torch.cuda.synchronize()
        eng.energy_stream_dev(ang, out); eng.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); eng.energy_stream_dev(ang, out); eng.synchronize(); ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        print("%-20s %8d lines x %d bands  %-12s %9.2f ms  %7.0f GB/s written (%.3f of 8 TB/s)"
              % (name, n, nw, "shared rows" if dedup == "1" else "every line", t * 1e3, n * nw * 24 / t / 1e9, n * nw * 24 / t / 8e12), flush=True)
        eng.close()
    del out
