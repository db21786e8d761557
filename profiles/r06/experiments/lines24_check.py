#!/usr/bin/env python3
"""stream_lines24_kernel (rings of 24 doubles, GORT_LINES_RING=24, measuring build) against stream_lines_kernel, bit for bit: band
counts around every edge of the half-block / class / ring arithmetic, ragged last waves, outputs at every 8-byte offset of a
cache line, the viewed proportions beside.  Run on a GPU box from the repo root with GORT_AMD_LIB=gort_amd/libgort_amd_ab.so."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from gort_amd import api

c = api.gap_probabilities(api.make_canopy(lai=4.0))
eng = api.Engine(); eng.set_canopy(c)
rng = np.random.default_rng(5)
bad = 0
cases = 0
for nw in (17, 18, 19, 23, 24, 25, 31, 32, 33, 39, 40, 41, 47, 48, 49, 63, 64, 65, 96, 100, 127, 128, 129, 190, 255, 257, 300, 511, 600):
    eng.set_spectra(*api.spectra(np.linspace(400.0, 2500.0, nw)))
    base = (1 << 18) // nw + 1
    for nA in (base, base + 37, ((base + 63) // 64) * 64, base + 64 * 3 + 1):
        ang = np.stack([rng.uniform(-89, 89, nA), rng.uniform(0, 360, nA), rng.uniform(0, 89, nA), rng.uniform(0, 360, nA)], 1)
        ang[::97, 0] = 91.0                                   # NaN rows
        a = torch.tensor(ang, device="cuda")
        for off in (0, 1, 3, 8, 13, 15):
            buf0 = torch.full((nA * nw + 64,), -7.0, dtype=torch.float64, device="cuda")
            buf1 = torch.full((nA * nw + 64,), -7.0, dtype=torch.float64, device="cuda")
            K0 = torch.zeros((nA, 4), dtype=torch.float64, device="cuda"); K1 = torch.zeros_like(K0)
            os.environ.pop("GORT_LINES_RING", None)
            eng.rsurf_stream_dev(a, buf0[off:off + nA * nw].view(nA, nw), K_t=K0); eng.synchronize()
            f0 = eng.stream_form()
            os.environ["GORT_LINES_RING"] = "24"
            eng.rsurf_stream_dev(a, buf1[off:off + nA * nw].view(nA, nw), K_t=K1); eng.synchronize()
            os.environ.pop("GORT_LINES_RING", None)
            cases += 1
            same = torch.equal(buf0.view(torch.int64), buf1.view(torch.int64)) and torch.equal(K0.view(torch.int64), K1.view(torch.int64))
            if not same:
                bad += 1
                d = (buf0.view(torch.int64) != buf1.view(torch.int64)).nonzero().flatten()
                print("MISMATCH nw=%d nA=%d off=%d form=%s: %d elements differ, first at %s (row %d band %d)"
                      % (nw, nA, off, f0, d.numel(), d[:4].tolist(), (int(d[0]) - off) // nw if d.numel() else -1, (int(d[0]) - off) % nw if d.numel() else -1), flush=True)
                if bad > 12:
                    sys.exit(1)
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
