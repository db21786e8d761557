import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from gort_amd import api
g = np.load('/root/repo/tests/golden/prospect_fuzz.npz')
wl = np.arange(400.0, 2501.0)
names = ("N", "Cab", "Car", "Anth", "Cbrown", "Cw", "Cm")
leaves = [api.leaf_soil(prospect=dict(zip(names, (float(x) for x in p)))) for p in g["params"]]
members = [api.gap_probabilities(api.make_canopy(lai=4.0))] * len(leaves)
eng = api.Engine(); eng.set_members_leaf(members, leaves, wl)
for m, (params, want) in enumerate(zip(g["params"], g["RT"])):
    _, rs, rl, tl = eng.get_member(m)
    got = np.stack([rl[g["bands"]], tl[g["bands"]]])
    ok = np.isfinite(want)
    if not ok.any(): continue
    e = np.abs(got[ok] - want[ok]) / np.maximum(np.abs(want[ok]), 1e-12)
    RT = api.prospect_d(*params); host = np.stack([RT[:2101][g["bands"]], RT[2101:][g["bands"]]])
    eh = np.abs(host[ok] - want[ok]) / np.maximum(np.abs(want[ok]), 1e-12)
    if e.max() > 1e-12:
        i = np.argmax(np.abs(got - want) / np.maximum(np.abs(want), 1e-12) * ok)
        print(m, "kind", m % 8, "dev err %.2e host err %.2e" % (e.max(), eh.max()), "params", np.round(params, 4), "at", np.unravel_index(i, got.shape), got.flat[i], want.flat[i])
