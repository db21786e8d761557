#!/usr/bin/env python3
"""Timings of the BASELINE.json configs other than the headline grid (those are parity-test cases, not
bench lines; this records what bounds each of them).  Device-resident inputs and outputs, HIP-side time
measured as wall time around stream synchronisation, best of 5."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gort_amd import api


def best(fn, eng, reps=5):
    # clocks up first: 0.2 s of the very call that is measured (a cold device runs the first calls ~10 % slower)
    t_up = time.perf_counter()
    while time.perf_counter() - t_up < 0.2:
        fn(); eng.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); eng.synchronize(); t.append(time.perf_counter() - t0)
    return min(t)


c = api.gap_probabilities(api.make_canopy(lai=4.0))
eng = api.Engine(); eng.set_canopy(c)

# C2: principal plane, 181 view zeniths x 1 band (stream entry point, device buffers)
eng.set_spectra(*api.spectra([800.0]))
ang = torch.tensor([[float(v), 0.0, 30.0, 0.0] for v in range(-90, 91)], dtype=torch.float64, device="cuda")
out = torch.empty((181, 1), dtype=torch.float64, device="cuda")
t = best(lambda: eng.rsurf_stream_dev(ang, out), eng)
print("C2  181 tuples x 1 band          : %8.1f us  %.3e samples/s  (launch latency bound: one fused kernel)" % (t * 1e6, 181 / t))

# C3: full hemisphere x 1 band (LUT entry point, few-band path)
g = api.hemisphere_grid(); rows = g.nsza * g.nvza
lut = torch.empty((rows * g.nphi, 1), dtype=torch.float64, device="cuda")
t = best(lambda: eng.rsurf_grid_dev(g, 0, rows, lut), eng)
print("C3  2 989 441 tuples x 1 band    : %8.1f us  %.3e samples/s  (fp64 transcendental bound: geometry kernel)" % (t * 1e6, rows * g.nphi / t))

# C4: spectral albedo + fAPAR, 91 sun zeniths x 2101 bands (energy entry point)
wl = np.arange(400.0, 2501.0)
eng.set_spectra(*api.spectra(wl))
sza = torch.tensor([[0.0, 0.0, float(s), 0.0] for s in range(91)], dtype=torch.float64, device="cuda")
en = torch.empty((91, wl.size, 3), dtype=torch.float64, device="cuda")
t = best(lambda: eng.energy_stream_dev(sza, en), eng)
print("C4  91 sun zeniths x 2101 bands  : %8.1f us  = %.3e BRDF evaluations/s equivalent (91 x 512 nodes x 2101 bands); "
      "the reference needs 512 rsurf calls per (sun zenith, band)" % (t * 1e6, 91 * 512 * wl.size / t))

# stream entry point at full spectrum: 65 536 random lines x 2101 bands, (a) 91 distinct sun zeniths, (b) every line
# its own sun zenith (the same kernel either way: per-line sun terms)
rng = np.random.default_rng(0)
n = 65536
o2 = torch.empty((n, wl.size), dtype=torch.float64, device="cuda")
for label, sza in (("91 sun zeniths", rng.integers(0, 90, n).astype(float)), ("all distinct  ", rng.uniform(0, 89, n))):
    a = torch.tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), sza, np.zeros(n)], 1), device="cuda")
    torch.cuda.synchronize()
    t = best(lambda: eng.rsurf_stream_dev(a, o2), eng)
    print("stream 65 536 lines x 2101, %s: %8.1f us  %.3e samples/s  %.0f GB/s written  (%s form)"
          % (label, t * 1e6, n * wl.size / t, n * wl.size * 8 / t / 1e9, eng.stream_form()))
# host in / host out (PCIe inclusive), what the CLI pays before formatting: into a pinned buffer (one DMA) and into an
# ordinary, already touched numpy array (pinned staging + threaded copy-out); tools/bench_host_path.py has the copy rate
ah = a.cpu().numpy()
pin = api.PinnedArray((n, wl.size))
t = best(lambda: eng.rsurf_stream(ah, want_K=True, out=pin.array), eng, reps=3)
print("same through host buffers, pinned output   : %8.1f ms  %.3e samples/s" % (t * 1e3, n * wl.size / t))
out = np.zeros((n, wl.size))
t = best(lambda: eng.rsurf_stream(ah, want_K=True, out=out), eng, reps=3)
print("same through host buffers, pageable output : %8.1f ms  %.3e samples/s" % (t * 1e3, n * wl.size / t))
