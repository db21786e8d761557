#!/usr/bin/env python3
"""Where the ONE chunk buffer of BASELINE config 5's LUTs lies decides more than how the chunks are cut: the same 40 chunks of 25
members into (a) a plain allocation, (b) the best of three whole-buffer draws, (c) a window placed by gort_lut_alloc's scan
(a buffer of twice the chunk, window = its first half).  tools/probes/c5_placement.py [MEMBERS]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from gort_amd import api
from gort_amd.ensemble import c5_grid, draw_c5_members
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
CHUNK = 25
wl = np.arange(400.0, 2501.0)
canopies, leaf = draw_c5_members(N)
g = c5_grid()
per_member = g.nvza * g.nphi * wl.size
eng = api.Engine()
eng.reserve_members(N, wl.size)
eng.set_members_leaf(*api.member_arrays(canopies, leaf), wl, compute_gaps=True)
eng.synchronize()


def run(lut):
    best = None
    for rep in range(3):
        eng.synchronize(); eng.last_expand_ms()
        t0 = time.perf_counter()
        for a in range(0, N, CHUNK):
            eng.rsurf_members_grid_dev(g, a, min(N, a + CHUNK), lut)
        eng.synchronize()
        t = time.perf_counter() - t0
        k = eng.last_expand_ms()
        if best is None or t < best[0]:
            best = (t, k)
    return best


for rnd in range(2):
    for what in ("plain", "best of 3", "scanned window", "plain", "scanned window"):
        if what == "plain":
            lut = eng.lut_alloc(CHUNK * per_member, max_draws=1)
        elif what == "best of 3":
            lut = eng.lut_alloc(CHUNK * per_member, max_draws=3)
        else:
            lut = eng.lut_alloc(2 * CHUNK * per_member, window=(0, CHUNK * per_member), max_draws=5)
        t, k = run(lut)
        print("%-15s: 40 chunks %.2f ms, expansion kernel %.3f ms per chunk = %.0f GB/s, %.3e samples/s; placement %s" %
              (what, t * 1e3, k, CHUNK * per_member * 8 / k / 1e6, N * per_member / t, {kk: lut.placement[kk] for kk in ("draws", "picked", "shifted", "slack_bytes")}), flush=True)
        lut.free()
