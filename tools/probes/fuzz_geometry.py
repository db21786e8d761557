"""A million random lines through the stream entry point (4 bands, K included) against the oracle on the CPU: the worst
relative error of the reflectance and the worst absolute error of the viewed proportions, by regime - uniform angles, both
zeniths near the horizon (89 ... 89.9999 deg), near nadir, near the hot spot (view within 1e-3 ... 1 deg of the sun),
near the line kernel's hand-over of arithmetics (cos zenith around 1e-6).  Run on a GPU box from the repo root."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gort_amd import api
from oracle import oracle as O

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
wl = np.array([450.0, 650.0, 865.0, 1640.0])
regimes = {}
u = lambda lo, hi: rng.uniform(lo, hi, n)
regimes["uniform"] = np.stack([u(-89, 89), u(-400, 400), u(0, 89), u(-400, 400)], 1)
regimes["horizon"] = np.stack([90 - 10 ** u(-4, 0), u(0, 360), 90 - 10 ** u(-4, 0), u(0, 360)], 1)
regimes["nadir"] = np.stack([10 ** u(-9, 0), u(0, 360), 10 ** u(-9, 0), u(0, 360)], 1)
sz, sa = u(0, 85), u(0, 360)
d = 10 ** u(-3, 0)
regimes["hot spot"] = np.stack([sz + d * rng.choice([-1, 1], n), sa + d * u(-1, 1), sz, sa], 1)
za = np.degrees(np.arccos(10 ** u(-7, -5)))
regimes["arithmetic hand-over"] = np.stack([za, u(0, 360), u(0, 89), u(0, 360)], 1)
for kw in (dict(lai=4.0), dict(newstyle=(2.0, 2.0, 0.6), lai=3.3)):
    c = api.gap_probabilities(api.make_canopy(**kw))
    eng = api.Engine(); eng.set_canopy(c); eng.set_spectra(*api.spectra(wl))
    oc = O.make_canopy(favd=c.favd, r=c.r, b=c.b, h1=c.h1, h2=c.h2, lam=c.lambda_, gaps=False)
    O.set_gap_tables(oc, np.array(c.p_n0), np.array(c.epgap), c.k_open, c.k_openep)
    rs, rl, tl = O.spectra(wl)
    for name, ang in regimes.items():
        r, _, K = eng.rsurf_stream(ang)
        ro, _, Ko = O.rsurf_stream(oc, ang, rs, rl, tl)
        ok = np.isfinite(ro).all(axis=1)
        assert np.array_equal(np.isnan(r), np.isnan(ro)), name
        er = np.abs(r[ok] - ro[ok]) / np.maximum(np.abs(ro[ok]), 1e-12)
        eK = np.abs(K[ok] - Ko[ok])
        i = np.unravel_index(np.argmax(er), er.shape)[0]
        print("%-22s %-40s rsurf max rel %.2e (line %s)  K max abs %.2e  [%d finite lines]"
              % (name, str(kw), er.max(), np.array2string(ang[ok][i], precision=6), eK.max(), ok.sum()), flush=True)
    eng.close()
