#!/usr/bin/env python3
"""Where does the time of a workgroup go?  Runs BASELINE configs 2, 3, 4, an `-energy` stream and a 100-band stream on the
measuring build of the library (python -m gort_amd.build --stamps -> gort_amd/libgort_amd_stamps.so, csrc/gort_stamps.h:
the kernels record the 100 MHz wall clock at their phase boundaries, per workgroup or wave, with the XCC and HW_ID of the
writing wave) and prints the phases and how the units of a launch lie in time.

    python tools/stamps.py [c2] [c3] [c4] [energy] [lines] [narrow]        (default: all; output kept in profiles/r04/stamps.log)

The product library has none of this compiled in; the numbers of a stamped kernel are a few per cent above the product's."""
import collections
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMPS_LIB = os.path.join(ROOT, "gort_amd", "libgort_amd_stamps.so")
if not os.path.exists(STAMPS_LIB):                         # (after a change of the sources: python -m gort_amd.build --stamps)
    from gort_amd import build
    build.build_stamps()
os.environ["GORT_AMD_LIB"] = STAMPS_LIB
import torch  # noqa: E402
from gort_amd import api  # noqa: E402

UNITS, SLOTS = 16384, 8
read_units_mask = None


def read(name, clear=True):
    fn = getattr(api.lib(), "gort_debug_stamps_" + name)
    fn.argtypes = [C.c_void_p, C.c_int]
    buf = np.zeros((UNITS, SLOTS), dtype=np.int64)
    if fn(buf.ctypes.data, 1 if clear else 0) != 0:
        raise RuntimeError("gort_debug_stamps_%s failed" % name)
    return buf


def clear(name):
    fn = getattr(api.lib(), "gort_debug_stamps_" + name)
    fn.argtypes = [C.c_void_p, C.c_int]
    fn(None, 1)


def warm(fn, eng, seconds=0.25):
    t = time.perf_counter()
    while time.perf_counter() - t < seconds:
        fn(); eng.synchronize()


def once(fn, eng, name):
    """The call warmed up, the buffer cleared, ONE call, its stamps: (microseconds of the call, stamps of the units that wrote)."""
    warm(fn, eng)
    clear(name)
    t0 = time.perf_counter(); fn(); eng.synchronize(); dt = time.perf_counter() - t0
    global read_units_mask
    buf = read(name)
    read_units_mask = buf[:, 0] != 0
    return dt * 1e6, buf[read_units_mask]


def phases(b, names):
    """b: stamps of the units (only the first len(names) + 1 columns count); prints per phase median, p10, p90 in us."""
    t = (b[:, :len(names) + 1] - b[:, 0].min()) / 100.0
    d = np.diff(t, axis=1)
    for k, n in enumerate(names):
        print("   %-28s median %6.2f   p10 %6.2f   p90 %6.2f   max %6.2f us" % (n, np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90), d[:, k].max()))
    print("   %-28s median %6.2f   p10 %6.2f   p90 %6.2f   max %6.2f us; units start within %.2f us, the last ends at %.2f us"
          % ("unit's life", np.median(t[:, -1] - t[:, 0]), np.percentile(t[:, -1] - t[:, 0], 10), np.percentile(t[:, -1] - t[:, 0], 90),
             (t[:, -1] - t[:, 0]).max(), t[:, 0].max(), t[:, -1].max()))
    return t


def placement(b):
    hw = b[:, SLOTS - 1] & 0xffffffff
    xcc = (b[:, SLOTS - 1] >> 32) & 15
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 3
    return xcc * 64 + se * 16 + sh * 8 + cu, simd


def end_clusters(t, width=1.5):
    e = np.sort(t[:, -1])
    groups, start = [], 0
    for i in range(1, len(e) + 1):
        if i == len(e) or e[i] - e[i - 1] > width:
            groups.append((e[start:i].mean(), i - start)); start = i
    big = [g for g in groups if g[1] >= max(3, len(e) // 100)]
    return ", ".join("%d units at %.1f us" % (n, m) for m, n in big)


def run_c2():
    eng = api.Engine(); eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0))); eng.set_spectra(*api.spectra([800.0]))
    ang = torch.tensor([[float(v), 0.0, 30.0, 0.0] for v in range(-90, 91)], dtype=torch.float64, device="cuda")
    out = torch.empty((181, 1), dtype=torch.float64, device="cuda")
    K = torch.empty((181, 4), dtype=torch.float64, device="cuda")
    for label, fn in (("reflectances only", lambda: eng.rsurf_stream_dev(ang, out)), ("with the viewed proportions", lambda: eng.rsurf_stream_dev(ang, out, None, K))):
        us, b = once(fn, eng, "geometry")
        print("== C2, principal plane -90 ... 90 x 1 band, %s: call %.1f us, %d waves (geometry_stream_kernel<fused>)" % (label, us, len(b)))
        t = (b[:, :4] - b[:, 0].min()) / 100.0
        for w in range(len(b)):
            print("   wave %d: angle line %.2f, geometry %.2f, sample + store %.2f, ends at %.2f us" % (w, t[w, 1] - t[w, 0], t[w, 2] - t[w, 1], t[w, 3] - t[w, 2], t[w, 3]))
    eng.close()


def run_c3():
    eng = api.Engine(); eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0))); eng.set_spectra(*api.spectra([800.0]))
    g = api.hemisphere_grid(); rows = g.nsza * g.nvza
    lut = torch.empty((rows * g.nphi, 1), dtype=torch.float64, device="cuda")
    for by_rows in ("0", "1"):
        os.environ["GORT_GRID_BY_ROWS"] = by_rows
        us, b = once(lambda: eng.rsurf_grid_dev(g, 0, rows, lut), eng, "geometry")
        os.environ.pop("GORT_GRID_BY_ROWS")
        print("== C3, hemisphere 91 x 91 x 361 x 1 band, partitioned by %s: call %.1f us, %d waves stamped (geometry_grid_kernel)" % ("rows" if by_rows == "1" else "nodes", us, len(b)))
        t = phases(b, ["row terms (to the barrier)", "azimuth nodes"])
        cu, simd = placement(b)
        per_cu = collections.Counter(collections.Counter(cu.tolist()).values())
        print("   waves per CU: %s;  ends in clusters: %s" % (sorted(per_cu.items()), end_clusters(t)))
        unit = np.flatnonzero(read_units_mask)                     # unit = 4 x workgroup + wave of the workgroup
        for k in range(4):
            m = unit % 4 == k
            if m.any():
                print("   wave %d of its workgroup (%4d): on SIMD %s; nodes begin at %.2f, end at median %.2f (p90 %.2f) us"
                      % (k, m.sum(), sorted(collections.Counter(simd[m].tolist()).items()), np.median(t[m, 1]), np.median(t[m, 2]), np.percentile(t[m, 2], 90)))
    eng.close()


def run_c4():
    eng = api.Engine(); eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
    wl = np.arange(400.0, 2501.0); eng.set_spectra(*api.spectra(wl))
    sza = torch.tensor([[0.0, 0.0, float(s), 0.0] for s in range(91)], dtype=torch.float64, device="cuda")
    en = torch.empty((91, wl.size, 3), dtype=torch.float64, device="cuda")
    us, b = once(lambda: eng.energy_stream_dev(sza, en), eng, "energy")
    print("== C4, 91 sun zeniths x 2101 bands (energy_kernel, the workgroups of the first band range): call %.1f us, %d lines" % (us, len(b)))
    phases(b, ["row terms (16 lanes)", "node geometry", "five sums", "band passes"])
    eng.close()


def run_energy():
    eng = api.Engine(); eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
    wl = np.arange(400.0, 2501.0); eng.set_spectra(*api.spectra(wl))
    n = 65536
    rng = np.random.default_rng(3)
    a = torch.tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89.9, n), rng.uniform(0, 360, n)], 1), device="cuda")
    en = torch.empty((n, wl.size, 3), dtype=torch.float64, device="cuda")
    for batch in ("1", "0"):
        os.environ["GORT_ENERGY_BATCH"] = batch
        us, b = once(lambda: eng.energy_stream_dev(a, en), eng, "energy")
        os.environ.pop("GORT_ENERGY_BATCH")
        what = "four lines per workgroup pass (energy_list_batched_kernel; unit = batch)" if batch == "1" else "one line after the other (energy_list_kernel; unit = line)"
        print("== -energy, %d lines with their own suns x 2101 bands, %s: call %.2f ms, %d units stamped" % (n, what, us / 1e3, len(b)))
        phases(b, ["row terms", "node geometry", "five sums", "band passes"])
    eng.close()


def run_lines():
    n = 1000000
    rng = np.random.default_rng(0)
    a = torch.tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), rng.uniform(0, 360, n)], 1), device="cuda")
    for nw in (32, 100):
        eng = api.Engine(); eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
        eng.set_spectra(*api.spectra(np.linspace(400.0, 2500.0, nw)))
        out = torch.empty((n, nw), dtype=torch.float64, device="cuda")
        us, b = once(lambda: eng.rsurf_stream_dev(a, out), eng, "lines")
        print("== stream, %d lines x %d bands (stream_lines_kernel): call %.1f us, %d waves" % (n, nw, us, len(b)))
        t = phases(b, ["geometry + line terms", "band blocks + cache lines", "seams"])
        life = t[:, -1] - t[:, 0]
        cu, simd = placement(b)
        mid = 0.5 * t[:, -1].max()
        alive = (t[:, 0] <= mid) & (t[:, -1] > mid)
        per_simd = collections.Counter(collections.Counter((cu[alive] * 4 + simd[alive]).tolist()).values())
        print("   the lives fill %.0f %% of the wave slots over the kernel's span; mid-kernel %d waves alive, per SIMD: %s"
              % (100.0 * life.sum() / (len(set(cu.tolist())) * 9 * t[:, -1].max()), alive.sum(), sorted(per_simd.items())))
        eng.close()
        del out


def run_narrow():
    n = 1000000
    rng = np.random.default_rng(0)
    a = torch.tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), rng.uniform(0, 360, n)], 1), device="cuda")
    for nw in (7, 16):
        eng = api.Engine(); eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
        eng.set_spectra(*api.spectra(np.linspace(400.0, 2500.0, nw)))
        out = torch.empty((n, nw), dtype=torch.float64, device="cuda")
        us, b = once(lambda: eng.rsurf_stream_dev(a, out), eng, "geometry")
        print("== stream, %d lines x %d bands (geometry_stream_kernel<fused>, unit = wave): call %.1f us, %d waves" % (n, nw, us, len(b)))
        t = phases(b, ["angle line", "geometry", "samples + stores"])
        life = t[:, -1] - t[:, 0]
        cu, simd = placement(b)
        print("   the lives fill %.0f %% of the wave slots (16 per CU) over the kernel's span" % (100.0 * life.sum() / (len(set(cu.tolist())) * 16 * t[:, -1].max())))
        eng.close()
        del out


if __name__ == "__main__":
    want = [w for w in sys.argv[1:] if not w.startswith("-")] or ["c2", "c3", "c4", "energy", "lines", "narrow"]
    for w in want:
        {"c2": run_c2, "c3": run_c3, "c4": run_c4, "energy": run_energy, "lines": run_lines, "narrow": run_narrow}[w]()
