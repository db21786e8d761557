#!/usr/bin/env python3
"""The arbitrary-angle stream (gortt.c:232-329) on device-resident buffers: N random lines x 2101 bands, with
91 distinct sun zeniths, every line its own sun zenith, one sun zenith, ... (the flat-panel kernel, include/gort_amd_tuning.h).
Prints, per case, the time of the expansion stage (HIP events on the engine's stream), the whole call (geometry included, wall clock around a stream synchronisation), the samples/s
and the fraction of the 8 TB/s HBM peak at 8 B per sample + 32 B per line."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gort_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
only = sys.argv[3] if len(sys.argv) > 3 else None
wl = np.arange(400.0, 2501.0)
c = api.gap_probabilities(api.make_canopy(lai=4.0))
eng = api.Engine(); eng.set_canopy(c); eng.set_spectra(*api.spectra(wl))
rng = np.random.default_rng(0)
cases = {
    "91 sun zeniths": rng.integers(0, 90, n).astype(float),
    "all distinct": rng.uniform(0, 89, n),
    "1 sun zenith": np.full(n, 30.0),
    "91 in runs of 32": ((np.arange(n) // 32) % 91).astype(float),
    "91 in runs of 4": ((np.arange(n) // 4) % 91).astype(float),
}
if os.environ.get("BENCH_STREAM_PLACED"):
    # the output through the allocator of the C ABI: placement of the 17.6 GB measured (the buffer is declared twice the
    # size, so that the window - the output - is placed by the scan of gort_lut_alloc)
    _buf = eng.lut_alloc(2 * n * wl.size, window=(0, n * wl.size), max_draws=5)
    pl = _buf.placement
    print("output placed by gort_lut_alloc: %d candidates, picked #%d at %.0f GB/s (median %.0f, first %.0f), rescans %d"
          % (pl["draws"], pl["picked"], pl["probe_gbs"][pl["picked"]], float(np.median(pl["probe_gbs"])), pl["probe_gbs"][0], pl["rescans"]), flush=True)
    out = _buf.tensor()[: n * wl.size].view(n, wl.size)
else:
    out = torch.empty((n, wl.size), dtype=torch.float64, device="cuda")
# the device's clocks take tens of milliseconds of load to come up (a pure-FMA probe runs 61 -> 67 -> 71 TFLOP/s over its
# first three 8-ms launches): whichever case is measured first would look ~10 % slower than the others
_a = torch.tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), np.zeros(n)], 1), device="cuda")
_t = time.perf_counter()
while time.perf_counter() - _t < 0.3:
    eng.rsurf_stream_dev(_a, out)
    eng.synchronize()
for name, sza in cases.items():
    if only and not any(o in name for o in only.split(",")):
        continue
    a = torch.tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), sza, np.zeros(n)], 1), device="cuda")
    for grouping in (1,):

        _t = time.perf_counter()                                 # every case starts from a busy device (the clocks sag within
        while time.perf_counter() - _t < 0.1:                    # the milliseconds it takes to build the case's angles)
            eng.rsurf_stream_dev(a, out)
            eng.synchronize()
        ex, wall = [], []
        eng.time_streams(False)                                  # the call as a user has it: no events around the expansion stage
        for _ in range(reps):
            t0 = time.perf_counter()
            eng.rsurf_stream_dev(a, out)
            eng.synchronize()
            wall.append(time.perf_counter() - t0)
        eng.time_streams(True)                                   # the stage alone, by the engine's events (they cost the call 6 us)
        for _ in range(reps):
            eng.rsurf_stream_dev(a, out)
            eng.synchronize()
            ex.append(eng.last_stream_ms() * 1e-3)
        form = eng.stream_form()
        byts = n * wl.size * 8 + n * 32
        e, w = float(np.median(ex)), float(np.median(wall))
        print("%-16s form=%d %-6s expansion %7.1f us (%5.0f GB/s, %.3f of 8 TB/s) | call %7.1f us  %.3e samples/s (%.3f)"
              % (name, grouping, form, e * 1e6, byts / e / 1e9, byts / e / 8e12, w * 1e6, n * wl.size / w, byts / w / 8e12), flush=True)
        if os.environ.get("BENCH_STREAM_JSON"):              # the same in the shape of bench.py's line, one object per case and form
            import json
            with open(os.environ["BENCH_STREAM_JSON"], "a") as f:
                f.write(json.dumps({
                    "metric": "BRDF samples/sec, arbitrary-angle stream", "value": n * wl.size / w, "unit": "samples/s", "n_gpus": 1,
                    "dtype": "f64", "data": "synthetic", "config": {"workload": "%d random lines x %d bands, sun zeniths: %s" % (n, wl.size, name),
                                                                    "form": form},
                    "ms_per_call": w * 1e3,
                    "roofline": {"bound": "hbm", "kernel": "expansion stage (HIP events on the engine's stream)", "achieved": byts / e / 1e9,
                                 "peak": 8000.0, "unit": "GB/s", "frac": byts / e / 8e12, "kernel_ms": e * 1e3,
                                 "algorithmic_bytes_per_launch": byts, "traffic": None}}) + "\n")
