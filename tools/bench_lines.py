#!/usr/bin/env python3
"""One stream shape, device-resident: tools/bench_lines.py LINES BANDS [REPS].  For profiles of the kernel a band count
takes: rocprofv3 --kernel-trace --stats / --pmc ... -- "$PY" tools/bench_lines.py 1000000 100 5 with
PY=$(python3 -c 'import sys; print(sys.executable)') - the interpreter BINARY itself after `--`: a `python3` shim, env or bash
there would be an exec hop behind a profiler that has already initialised the GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gort_amd import api

n, nw = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 15
c = api.gap_probabilities(api.make_canopy(lai=4.0))
eng = api.Engine(); eng.set_canopy(c)
eng.set_spectra(*api.spectra(np.linspace(400.0, 2500.0, nw)))
rng = np.random.default_rng(0)
a = torch.tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), rng.uniform(0, 360, n)], 1), device="cuda")
out = torch.empty((n, nw), dtype=torch.float64, device="cuda")
t_up = time.perf_counter()
while time.perf_counter() - t_up < 0.25:       # clocks up
    eng.rsurf_stream_dev(a, out)
    eng.synchronize()
ex, wall = [], []
for _ in range(reps):                                    # the call as a user has it: no events around the expansion stage
    t0 = time.perf_counter(); eng.rsurf_stream_dev(a, out); eng.synchronize(); wall.append(time.perf_counter() - t0)
eng.time_streams(True)                                   # the stage alone, by the engine's events (they cost the call 6 us)
for _ in range(reps):
    eng.rsurf_stream_dev(a, out); eng.synchronize(); ex.append(eng.last_stream_ms() * 1e-3)
e, w = float(np.median(ex)), float(np.median(wall))
b = n * nw * 8 + n * 32
print("%8d lines x %4d bands (%s): expansion stage %8.1f us, call %8.1f us = %.3e samples/s, %5.0f GB/s (%.3f of 8 TB/s)"
      % (n, nw, eng.stream_form(), e * 1e6, w * 1e6, n * nw / w, b / w / 1e9, b / w / 8e12), flush=True)
