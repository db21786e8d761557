import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gort_amd import api
c = api.gap_probabilities(api.make_canopy(lai=4.0))
eng = api.Engine(); eng.set_canopy(c)
wl = np.arange(400.0, 2501.0)
eng.set_spectra(*api.spectra(wl))
rng = np.random.default_rng(91)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 70001
pool = np.concatenate([np.arange(0.0, 90.0), -np.arange(1.0, 45.0)])
ang = np.stack([rng.uniform(-89, 89, n), rng.uniform(-400, 400, n), rng.choice(pool, n), rng.uniform(-400, 400, n)], 1)
a = torch.as_tensor(ang, device="cuda")
res = {}
for grouping in (True, False):
    out = torch.full((n, wl.size), -7.0, dtype=torch.float64, device="cuda")
    eng.set_stream_grouping(grouping)
    eng.rsurf_stream_dev(a, out)
    print("grouping", grouping, "form", eng.stream_form(), "ms", eng.last_stream_ms())
    bad = (out == -7.0)
    nb = int(bad.sum())
    print("  unwritten:", nb)
    if nb:
        idx = torch.nonzero(bad.flatten()).flatten().cpu().numpy()
        print("  first", idx[:10], "last", idx[-10:], "rows", np.unique(idx // wl.size)[:20])
    res[grouping] = out
d = (res[True].view(torch.int64) != res[False].view(torch.int64))
print("differing:", int(d.sum()))
if int(d.sum()):
    idx = torch.nonzero(d.flatten()).flatten().cpu().numpy()
    print(" first", idx[:10], "rows", np.unique(idx // wl.size)[:20], "bands", np.unique(idx % wl.size)[:40])
    i = idx[0]
    print(res[True].flatten()[i].item(), res[False].flatten()[i].item())
