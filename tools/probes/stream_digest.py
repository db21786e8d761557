"""sha256 over the outputs of wide streams (5 band counts x 4 output alignments, NaN lines included): two builds of the
stream kernel that print the same digest wrote the same bits.  Run from the repo root on a GPU box."""
import sys, os, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gort_amd import api
c = api.gap_probabilities(api.make_canopy(lai=4.0))
eng = api.Engine(); eng.set_canopy(c)
rng = np.random.default_rng(5)
h = hashlib.sha256()
for nw, n in ((2101, 70001), (2101, 120000), (129, 40000), (1999, 9000), (143, 35000), (33, 150000), (100, 60000), (300, 20000), (256, 30000), (640, 9000)):
    wl = np.arange(400.0, 2501.0) if nw == 2101 else np.linspace(400.0, 2500.0, nw)
    eng.set_spectra(*api.spectra(wl))
    ang = np.stack([rng.uniform(-89, 89, n), rng.uniform(-400, 400, n), rng.choice(np.arange(0.0, 90.0), n), rng.uniform(-400, 400, n)], 1)
    ang[7, 2] = 95.0; ang[11, 0] = np.nan
    a = torch.as_tensor(ang, device="cuda")
    for off in (0, 1, 5, 16):
        buf = torch.full((n * nw + 32,), -7.0, dtype=torch.float64, device="cuda")
        out = buf[off:off + n * nw].view(n, nw)
        torch.cuda.synchronize()
        eng.rsurf_stream_dev(a, out); eng.synchronize()
        assert eng.stream_form() == ("lines" if nw <= 255 or (nw <= 600 and nw % 128) else "flat")
        assert float(buf[:off].min() if off else -7.0) == -7.0 and float(buf[off + n * nw:].max()) == -7.0
        assert not bool((out == -7.0).any())
        h.update(out.cpu().numpy().tobytes())
print("digest", h.hexdigest())
