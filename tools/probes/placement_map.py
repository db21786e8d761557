#!/usr/bin/env python3
"""Round 3: where in device memory does the LUT kernel write fast?  (a) 16 separate allocations of one N=8 window
(6.3 GB), alive together; (b) ONE allocation of 16 windows, every window inside it; (c) the same for N=4 windows
(12.6 GB, 8 of them); for a few of the regions also the real kernel's time.  Store-pattern GB/s per region."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ctypes as C
import numpy as np
from gort_amd import api

eng = api.Engine()
eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
wl = np.arange(400.0, 2501.0)
eng.set_spectra(*api.spectra(wl))
grid = api.hemisphere_grid()
row_elems = grid.nphi * wl.size


def kernel_ms(ptr, rows):
    for _ in range(3):
        eng.rsurf_grid_dev(grid, 0, rows, ptr)
    eng.synchronize(); eng.last_expand_ms()
    for _ in range(20):
        eng.rsurf_grid_dev(grid, 0, rows, ptr)
    eng.synchronize()
    return eng.last_expand_ms()


for world in (8, 4):
    rows = -(-8281 // world)
    wbytes = rows * row_elems * 8
    n = 16 if world == 8 else 8
    print("== windows of N=%d: %d rows = %.2f GB" % (world, rows, wbytes / 1e9), flush=True)
    bufs = [api.DeviceBuffer(wbytes) for _ in range(n)]
    rates = [eng.probe_store_pattern(b, wbytes) for b in bufs]
    print("separate allocations   :", " ".join("%.0f" % r for r in rates), flush=True)
    for i in (int(np.argmax(rates)), int(np.argmin(rates))):
        print("   kernel on #%d (%.0f GB/s probe): %.3f ms = %.0f GB/s" % (i, rates[i], kernel_ms(bufs[i], rows), wbytes / kernel_ms(bufs[i], rows) / 1e6), flush=True)
    addrs = [b.ptr for b in bufs]
    print("   device addresses (GB):", " ".join("%.1f" % ((a - min(addrs)) / 1e9) for a in addrs))
    for b in bufs:
        b.free()
    big = api.DeviceBuffer(n * wbytes)
    rates = [eng.probe_store_pattern(C.c_void_p(big.ptr + i * wbytes), wbytes) for i in range(n)]
    print("windows of ONE allocation:", " ".join("%.0f" % r for r in rates), flush=True)
    for i in (int(np.argmax(rates)), int(np.argmin(rates))):
        p = C.c_void_p(big.ptr + i * wbytes)
        print("   kernel on window %d (%.0f GB/s probe): %.3f ms = %.0f GB/s" % (i, rates[i], kernel_ms(p, rows), wbytes / kernel_ms(p, rows) / 1e6), flush=True)
    # the same windows, shifted by half a window
    rates = [eng.probe_store_pattern(C.c_void_p(big.ptr + i * wbytes + wbytes // 2 // 4096 * 4096), wbytes) for i in range(n - 1)]
    print("   ... shifted by half   :", " ".join("%.0f" % r for r in rates), flush=True)
    big.free()
eng.close()
