#!/usr/bin/env python3
"""The stream family's sample against the oracle on the CPU, by kernel form: random lines (uniform angles; a share of them with the
view near the sun and with zeniths near the horizon) x 100 bands (line kernel), x 2101 bands (flat panels from 2000 lines, narrow
kernels below) and x 7 bands (fused with the geometry).  Worst and 99.9th-percentile relative error of the reflectance.
GORT_AMD_LIB selects the library (tools/probes/sample_form_ab.sh: the tree before the 22 + 1 sample against the tree with it).
Run on a GPU box from the repo root."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from gort_amd import api
from oracle import oracle as O

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
print("library:", os.path.basename(api.LIB_PATH))


def lines(n):
    a = np.stack([rng.uniform(-89, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), rng.uniform(0, 360, n)], 1)
    k = n // 8
    a[:k, 0] = a[:k, 2] + rng.uniform(-0.5, 0.5, k)          # near the hot spot
    a[:k, 1] = a[:k, 3] + rng.uniform(-0.5, 0.5, k)
    a[k:2 * k, 2] = 90 - 10 ** rng.uniform(-3, 0.5, k)       # sun near the horizon
    a[2 * k:3 * k, 0] = 90 - 10 ** rng.uniform(-3, 0.5, k)   # view near the horizon
    return a


for kw in (dict(lai=4.0), dict(newstyle=(2.0, 2.0, 0.6), lai=1.7)):
    c = api.gap_probabilities(api.make_canopy(**kw))
    eng = api.Engine(); eng.set_canopy(c)
    oc = O.make_canopy(favd=c.favd, r=c.r, b=c.b, h1=c.h1, h2=c.h2, lam=c.lambda_, gaps=False)
    O.set_gap_tables(oc, np.array(c.p_n0), np.array(c.epgap), c.k_open, c.k_openep)
    for n, nw in ((40000, 7), (40000, 100), (6000, 2101), (1500, 2101)):
        wl = np.linspace(400.0, 2500.0, nw)
        eng.set_spectra(*api.spectra(wl))
        rs, rl, tl = O.spectra(wl)
        ang = lines(n)
        r, _, _ = eng.rsurf_stream(ang, want_K=False)
        form = eng.stream_form()
        ro, _, _ = O.rsurf_stream(oc, ang, rs, rl, tl)
        assert np.array_equal(np.isnan(r), np.isnan(ro))
        ok = np.isfinite(ro)
        er = np.abs(r[ok] - ro[ok]) / np.maximum(np.abs(ro[ok]), 1e-12)
        print("%-36s %6d lines x %4d bands (%-6s): max rel %.2e  99.9 %% %.2e  median %.2e  [%d samples]"
              % (str(kw), n, nw, form, er.max(), np.quantile(er, 0.999), np.median(er), er.size), flush=True)
    eng.close()
