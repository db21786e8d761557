#!/usr/bin/env python3
"""The shared-sun-zenith stream's expansion (gort_stream_suns.hip) against the flat-panel kernel, variants interleaved in ONE
process on ONE output buffer (boxes and allocations differ by 10-15 %): tools/probes/suns_variants.py [LINES [BANDS]]
Needs the measuring build (python -m gort_amd.build --ab)."""
import os, sys, time
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
os.environ["GORT_AMD_LIB"] = os.path.join(R, "gort_amd", "libgort_amd_ab.so")
import torch
from gort_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 2101
wl = np.arange(400.0, 2501.0) if nw == 2101 else np.linspace(400.0, 2500.0, nw)
eng = api.Engine(); eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0))); eng.set_spectra(*api.spectra(wl))
rng = np.random.default_rng(0)
cases = {"91 sun zeniths": rng.integers(0, 90, n).astype(float), "1 sun zenith": np.full(n, 30.0),
         "91 in runs of 4096": ((np.arange(n) // 4096) % 91).astype(float)}
out = torch.empty((n, nw), dtype=torch.float64, device="cuda")
variants = [("flat panels", 0, {}), ("suns, record per lane, 64 lines", 2, {"GORT_SUNS_FETCH": "lanes", "GORT_SUNS_SEG": "64"}),
            ("suns, record per lane, 32 lines", 2, {"GORT_SUNS_FETCH": "lanes", "GORT_SUNS_SEG": "32"}),
            ("suns, scalar record, 32 lines", 2, {"GORT_SUNS_FETCH": "scalar", "GORT_SUNS_SEG": "32"}),
            ("suns, scalar record, 64 lines", 2, {"GORT_SUNS_FETCH": "scalar", "GORT_SUNS_SEG": "64"})]
eng.time_streams(True)
for name, sza in cases.items():
    a = torch.tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), sza, np.zeros(n)], 1), device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        eng.rsurf_stream_dev(a, out); eng.synchronize()
    for rnd in range(2):
        for label, mode, env in variants:
            for k in ("GORT_SUNS_FETCH", "GORT_SUNS_SEG"):
                os.environ.pop(k, None)
            os.environ.update(env)
            eng.set_stream_sun_sharing(mode)
            ex = []
            for _ in range(6):
                eng.rsurf_stream_dev(a, out); eng.synchronize(); ex.append(eng.last_stream_ms() * 1e3)
            e = float(np.median(ex[1:]))
            print("%-20s round %d  %-34s %-5s expansion %8.1f us  %.3f of 8 TB/s" % (name, rnd, label, eng.stream_form(), e, n * nw * 8 / e / 8e6), flush=True)
