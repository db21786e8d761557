#!/usr/bin/env python3
"""Round 3: the write rate of an N=8 / N=4 / N=2 window (6.3 / 12.6 / 25 GB) as a function of WHERE inside one big
allocation it lies, in 1-GiB steps (store-pattern probe, GB/s), and the absolute device address.
  placement_scan.py [trials] [worlds, e.g. 8,4,2] [max offset GiB]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ctypes as C
import numpy as np
from gort_amd import api

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 2
worlds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,4,2").split(",")]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 100
eng = api.Engine()
eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
eng.set_spectra(*api.spectra(np.arange(400.0, 2501.0)))
row_elems = 361 * 2101
G = 1 << 30
print("GORT_XCD_ROTATE=%s" % os.environ.get("GORT_XCD_ROTATE", "0"))
for trial in range(trials):
    big = api.DeviceBuffer((top + 12) * G)
    print("== allocation %d at device address 0x%x = %.3f GiB, mod 32 GiB = %.3f GiB" % (trial, big.ptr, big.ptr / G, (big.ptr % (32 * G)) / G), flush=True)
    for world in worlds:
        wbytes = -(-8281 // world) * row_elems * 8
        out = []
        for off in range(0, top - int(wbytes / G), 1 if world == 8 else 2):
            r = eng.probe_store_pattern(C.c_void_p(big.ptr + off * G), wbytes)
            out.append("%d:%.0f" % (off, r))
        print("N=%d window %.2f GiB, offset GiB : GB/s  " % (world, wbytes / G) + " ".join(out), flush=True)
    hold = api.DeviceBuffer(5 * G)       # shift the next allocation
    big.free()
eng.close()
