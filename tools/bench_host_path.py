#!/usr/bin/env python3
"""Host-buffer entry point gort_rsurf_stream (what the CLI pays before formatting): 65 536 random lines x 2101
bands = 1.1 GB of results over PCIe.  Against the box's own pinned device->host copy rate."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gort_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
wl = np.arange(400.0, 2501.0)
c = api.gap_probabilities(api.make_canopy(lai=4.0))
eng = api.Engine(); eng.set_canopy(c); eng.set_spectra(*api.spectra(wl))
rng = np.random.default_rng(0)
ang = np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.integers(0, 90, n).astype(float), np.zeros(n)], 1)
nbytes = n * wl.size * 8
pin = api.PinnedArray((n, wl.size))
dev = api.DeviceBuffer(nbytes)
import ctypes as C
def d2h():
    api._check(api.lib().gort_memcpy_d2h(C.c_void_p(pin.ptr), C.c_void_p(dev.ptr), nbytes))
d2h()
t = []
for _ in range(5):
    t0 = time.perf_counter(); d2h(); t.append(time.perf_counter() - t0)
raw = nbytes / min(t) / 1e9
print("pinned device->host copy of %.2f GB: %.1f ms = %.1f GB/s (the box's PCIe rate)" % (nbytes / 1e9, min(t) * 1e3, raw))
def best(fn, reps=5):
    fn()
    tt = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); tt.append(time.perf_counter() - t0)
    return min(tt)
t = best(lambda: eng.rsurf_stream(ang, want_K=True, out=pin.array))
print("gort_rsurf_stream -> pinned buffer  : %7.1f ms  %.3e samples/s  %.1f GB/s = %.2f of the copy rate" % (t * 1e3, n * wl.size / t, nbytes / t / 1e9, nbytes / t / 1e9 / raw))
out = np.empty((n, wl.size))
t = best(lambda: eng.rsurf_stream(ang, want_K=True, out=out))
print("gort_rsurf_stream -> pageable buffer: %7.1f ms  %.3e samples/s  %.1f GB/s = %.2f of the copy rate (pinned staging, 3 chunks in flight, threaded copy-out)" % (t * 1e3, n * wl.size / t, nbytes / t / 1e9, nbytes / t / 1e9 / raw))
