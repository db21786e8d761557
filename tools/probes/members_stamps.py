#!/usr/bin/env python3
"""Where the waves of the members stream (stream_lines_kernel<.., MEMBERS>) lie in time, per XCD: tools/probes/members_stamps.py LINES BANDS
on the stamps build (python -m gort_amd.build --stamps).  Prints per XCD the number of waves, when its first wave started and its last
one ended, the mean life of a wave and of its phases (geometry / band blocks / seams), and the waves resident over time."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
os.environ["GORT_AMD_LIB"] = os.path.join(ROOT, "gort_amd", "libgort_amd_stamps.so")
import torch
from gort_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 2101
M = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
rng = np.random.default_rng(12345)
canopies, leaf = [], []
for _ in range(M):
    hb, br, pcc, lai = rng.uniform(1, 3), rng.uniform(1, 3.5), rng.uniform(0.2, 0.8), rng.uniform(0.5, 6)
    canopies.append(api.make_canopy(newstyle=(float(np.float32(hb)), float(np.float32(br)), float(np.float32(pcc))), lai=float(np.float32(lai))))
    leaf.append(api.leaf_soil(prospect=dict(N=rng.uniform(1, 2.5), Cab=rng.uniform(10, 60), Cw=rng.uniform(0.005, 0.03), Cm=rng.uniform(0.002, 0.015)),
                              rsl=(rng.uniform(0.05, 0.4), 0.1, 0.03726, -0.002426)))
ang = torch.as_tensor(np.stack([rng.uniform(0, 70, n), rng.uniform(0, 360, n), rng.uniform(10, 70, n), rng.uniform(0, 360, n)], 1), device="cuda")
e = api.Engine()
e.set_members_leaf(canopies, leaf, np.linspace(400.0, 2500.0, nw), compute_gaps=True); e.synchronize()
out = torch.empty((M, n, nw), dtype=torch.float64, device="cuda")
f = lambda: api._check(api.lib().gort_rsurf_members_stream_dev(e.h, api._ptr(ang), n, 0, M, api._ptr(out)))
fn = api.lib().gort_debug_stamps_lines
fn.argtypes = [C.c_void_p, C.c_int]
for _ in range(5):
    f(); e.synchronize()
fn(None, 1)
t0 = time.perf_counter(); f(); e.synchronize(); dt = time.perf_counter() - t0
buf = np.zeros((16384, 8), dtype=np.int64)
assert fn(buf.ctypes.data, 1) == 0
b = buf[buf[:, 0] != 0]
t = (b[:, :4] - b[:, 0].min()) / 100.0          # us (100 MHz)
xcc = (b[:, 7] >> 32) & 15
hw = b[:, 7] & 0xffffffff
cu = (hw >> 8) & 15; se = (hw >> 13) & 7; sh = (hw >> 12) & 1
print("%d members x %d lines x %d bands: call %.1f us, %d waves stamped, launch span %.1f us" % (M, n, nw, dt * 1e6, len(b), t[:, 3].max()))
print("xcd  waves  first-start  last-end   life  geometry  bands  seams   CUs")
for x in sorted(set(xcc)):
    m = xcc == x
    tt = t[m]
    print("%3d %6d %10.1f %10.1f %7.1f %8.1f %7.1f %6.1f  %4d" % (x, m.sum(), tt[:, 0].min(), tt[:, 3].max(), (tt[:, 3] - tt[:, 0]).mean(), (tt[:, 1] - tt[:, 0]).mean(),
          (tt[:, 2] - tt[:, 1]).mean(), (tt[:, 3] - tt[:, 2]).mean(), len(set(zip(se[m], sh[m], cu[m])))))
# resident waves over time, all XCDs, ten slices
T = t[:, 3].max()
edges = np.linspace(0, T, 11)
res = [((t[:, 0] < (a + c) / 2) & (t[:, 3] > (a + c) / 2)).sum() for a, c in zip(edges[:-1], edges[1:])]
print("resident waves at the middle of ten slices of the launch:", res)
order = np.argsort(b[:, 0])
print("unit ids of the first 24 waves to start:", list(np.nonzero(buf[:, 0] != 0)[0][order[:24]]))
