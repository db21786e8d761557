#!/usr/bin/env python3
"""Headline benchmark: BRDF samples/s on the synthetic full-hemisphere x full-spectrum grid.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric; SURVEY.md 8d "metric grid"): sun zenith 0..90, view
zenith 0..90, relative azimuth 0..360 in integer degrees x 400..2500 nm @ 1 nm =
2 989 441 angle tuples x 2101 bands = 6.28e9 (theta_v, theta_s, dphi, lambda) samples,
50.2 GB of fp64 output, canopy `-LAI 4.0`, default PROSPECT-D / Price parameters.

One step = one full evaluation of the grid into HBM: the per-angle geometry kernel, the
(sun zenith, band) table kernel and the LUT expansion kernel, all inside the timed
region.  Inputs (gap tables, spectra) are resident in HBM before timing starts; the
output stays in HBM.  With N ranks the 8281 (sun zenith, view zenith) rows are split into
N contiguous slabs, total work fixed (strong scaling), and every rank computes into ITS WINDOW of one gatherable
LUT buffer; no collective inside the step.  After the timed steps the in-place RCCL all-gather that reassembles the
LUT on every GPU runs once and is reported (`allgather`: ms, GB/s received per GPU against the xGMI bound, parity of
rows that came from other ranks) - see DESIGN.md "Multi-GPU" for why it is not part of the step.  A `config5` block
(the 1000-member ensemble, members sharded, its energy-table all-gather timed) follows.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_SAMPLE = 8.0     # one fp64 value written per sample (grid mode)


def cpu_baseline(wl, budget_s=20.0, force_port=False, procs=1):
    """Reference `gortt` (oracle/_ref, built from /root/reference in the build container) timed on this
    box's host CPU.  The program is single-threaded: `procs` > 1 runs that many copies at once, each on its own
    angle stream (how a user would use all cores).  Falls back to the oracle port (1 thread)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "gortt") if not force_port else "/nonexistent"
    cores = 1
    try:
        model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    if os.path.exists(ref) and os.access(ref, os.X_OK):
        try:
            lut = subprocess.run([ref, "-LAI", "4.0", "-W"], capture_output=True, timeout=120).stdout
            lut_path = "/tmp/gort_bench_lut_%d.dat" % os.getpid()
            open(lut_path, "wb").write(lut)
            nw = 180                                   # header line limit of the reference (999 chars)
            w = wl[:: max(1, len(wl) // nw)][:nw]
            rng = np.random.default_rng(1)

            def stream(n):
                a = np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), np.zeros(n)], 1)
                head = "%d %d %s\n" % (n, len(w), " ".join("%d" % x for x in w))
                return (head + "".join("%.4f %.4f %.4f %.4f\n" % tuple(r) for r in a)).encode()

            n = 2000
            t0 = time.perf_counter()
            subprocess.run([ref, "-LAI", "4.0", "-P", lut_path], input=stream(n), stdout=subprocess.DEVNULL, timeout=300)
            dt = time.perf_counter() - t0
            n = int(min(max(n * budget_s / max(dt, 1e-3), n), 400000))
            files = []
            for p in range(procs):
                path = "/tmp/gort_bench_stream_%d_%d.txt" % (os.getpid(), p)
                open(path, "wb").write(stream(n))
                files.append(path)
            t0 = time.perf_counter()
            running = [subprocess.Popen([ref, "-LAI", "4.0", "-P", lut_path], stdin=open(f, "rb"),
                                        stdout=subprocess.DEVNULL) for f in files]
            rcs = [p.wait(timeout=900) for p in running]
            dt = time.perf_counter() - t0
            for f in files + [lut_path]:
                os.unlink(f)
            if any(rcs):
                raise RuntimeError("reference exited with %r" % rcs)
            return {"value": procs * n * len(w) / dt, "unit": "samples/s", "cores": procs, "kind": "reference",
                    "sample": "reference gortt (-O3 build of /root/reference, gap LUT via -P, text I/O to /dev/null): "
                              "%d process(es) x %d random angle lines x %d bands in %.1f s; cpu: %s"
                              % (procs, n, len(w), dt, model)}
        except Exception as ex:                      # fall through to the port
            print("cpu_baseline: reference run failed (%s); using the oracle port" % ex, file=sys.stderr)
    from oracle import oracle as O
    O.AUTO_BUILD = False
    c = O.make_canopy(lai=4.0)
    rs, rl, tl = O.spectra(wl)
    rng = np.random.default_rng(1)
    n = 400
    a = np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), np.zeros(n)], 1)
    t0 = time.perf_counter(); O.rsurf_stream(c, a, rs, rl, tl, want_K=False); dt = time.perf_counter() - t0
    n = int(min(max(n * budget_s / max(dt, 1e-3), n), 200000))
    a = np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), np.zeros(n)], 1)
    t0 = time.perf_counter(); O.rsurf_stream(c, a, rs, rl, tl, want_K=False); dt = time.perf_counter() - t0
    return {"value": n * len(wl) / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "oracle/gort_oracle.c (scalar C restatement, no I/O): %d random angle lines x %d bands in %.1f s; cpu: %s"
                      % (n, len(wl), dt, model)}


def _ref_grid_worker(args):
    """One process of cpu_baseline_grid: the reference's gortt_rsurf (oracle/_ref/libgortt_ref.so, built from
    /root/reference by oracle/Makefile) on a share of the metric grid's nodes, all 2101 bands, no text I/O."""
    so, nodes, wl = args[:3]
    keep = args[3] if len(args) > 3 else 0                  # return the rows of the first `keep` nodes (parity_reference)
    rows = []
    import ctypes as C
    L = C.CDLL(so)
    D = C.c_double
    argv = (C.c_char_p * 3)(b"gortt", b"-LAI", b"4.0")
    L.refshim_canopy(3, argv)
    nw = len(wl)
    w = np.ascontiguousarray(wl, dtype=np.float64)
    rs, rl, tl = np.zeros(nw), np.zeros(nw), np.zeros(nw)
    pd = lambda a: a.ctypes.data_as(C.POINTER(D))
    L.refshim_spectra(pd(w), nw, pd(rs), pd(rl), pd(tl))
    L.refshim_rsurf.argtypes = [D] * 5 + [C.POINTER(D)] * 4
    out, sc, K, pr = np.zeros(nw), np.zeros(4 * nw), np.zeros(4), np.zeros(4)
    po, ps, pk, pp = pd(out), pd(sc), pd(K), pd(pr)
    rad = np.pi / 180.0
    t0 = time.perf_counter()
    acc = 0.0
    for isza, ivza, iphi in nodes:
        # the normalisation of main() for the line "vza phi sza 0" (gortt.c:240-279)
        vza, vaa, sza, saa = ivza * rad, iphi * rad, isza * rad, 0.0
        raa = saa - vaa
        raa = abs(raa - 2 * np.pi * int(0.5 + raa / (2 * np.pi)))
        L.refshim_rsurf(vza, vaa, sza, saa, raa, po, ps, pk, pp)
        acc += out[0]
        if len(rows) < keep:
            rows.append(out.copy())
    return time.perf_counter() - t0, len(nodes), acc, rows


def cpu_baseline_grid(wl, procs, budget_s=12.0):
    """The SAME workload shape as the GPU line, on the host CPU, through the reference itself: random nodes of the
    metric grid x all 2101 bands via gortt_rsurf (no LUT file, no text), one process per core.  None where the
    reference build did not travel (oracle/_ref is built in the build container only)."""
    so = os.path.join(ROOT, "oracle", "_ref", "libgortt_ref.so")
    if not os.path.exists(so):
        return None
    import multiprocessing as mp
    rng = np.random.default_rng(2)
    try:
        model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"

    def draw(n):
        return [tuple(int(x) for x in r) for r in np.stack([rng.integers(0, 90, n), rng.integers(0, 90, n), rng.integers(0, 361, n)], 1)]
    ctx = mp.get_context("spawn")                      # never fork a process that has initialised the GPU
    with ctx.Pool(procs) as pool:
        first = draw(60)
        dt, n, _, rows = pool.apply(_ref_grid_worker, ((so, first, wl, 24),))
        per_node = dt / n
        n_each = int(min(max(budget_s / per_node, 100), 20000))
        t0 = time.perf_counter()
        res = pool.map(_ref_grid_worker, [(so, draw(n_each), wl) for _ in range(procs)])
        wall = time.perf_counter() - t0
    total = sum(r[1] for r in res) * len(wl)
    busy = max(r[0] for r in res)
    # the first 24 nodes' rows travel back for a parity check of the GPU LUT against the reference itself (main())
    return {"_check_nodes": first[:len(rows)], "_check_rows": np.array(rows),
            "value": total / busy, "unit": "samples/s", "cores": procs, "kind": "reference",
            "sample": "reference gortt_rsurf (oracle/_ref/libgortt_ref.so = /root/reference compiled -O3; direct gap "
                      "probabilities, no text I/O): %d processes x %d random nodes of the metric grid x %d bands, slowest "
                      "process %.1f s (wall %.1f s incl. each process's 0.4 s gap-probability setup); cpu: %s"
                      % (procs, n_each, len(wl), busy, wall, model)}


PARK_DEADLINE_S = 420.0      # N > 1: how long the other ranks wait for rank 0's traffic pass and cpu_baseline
XGMI_BOUND_GBS = 7 * 153.0   # receive bound of one GPU: seven xGMI links x ~153 GB/s (MI355X_MICROARCH.md)


def reference_build_id():
    """sha256 (first 16 hex digits) of the reference build this run can time and check against: oracle/_ref/libgortt_ref.so,
    compiled from /root/reference by oracle/Makefile in the build container (oracle/_ref.MANIFEST holds the hashes of
    record); 'absent' where it did not travel - cpu_baseline then falls back to the port and says so in `kind`."""
    so = os.path.join(ROOT, "oracle", "_ref", "libgortt_ref.so")
    if not os.path.exists(so):
        return "absent"
    import hashlib
    h = hashlib.sha256()
    with open(so, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def _single_process_env():
    """The environment of a child that runs alone on this rank's GPU: none of the launcher's rendezvous variables."""
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
            "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")
    return {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_")}


def measure_traffic(args, timeout_s=150, world=1):
    """HBM bytes per launch of the dominant kernel, measured in THIS run: two child passes of this script under
    `rocprofv3 --pmc` (WRITE_SIZE, then FETCH_SIZE: separate passes, as MI355X_MICROARCH.md prescribes), 3 steps each, no
    parity / baselines / config 5; bytes = (WRITE_SIZE + 2 x FETCH_SIZE) x 1024, averaged over the launches of
    expand_flat_kernel (gfx950's FETCH_SIZE counts half the bytes of wide reads).  Returns (bytes or None, how)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-parity",
             "--sustain-s", "0", "--no-config5", "--no-configs", "--lut-draws", "1", "--placement-evidence", "0", "--no-traffic", "--nsza", str(args.nsza), "--nw", str(args.nw)]
    if world > 1:
        child += ["--slab-of-world", str(world)]          # rank 0's launch: its slab, in its window of the gatherable buffer
    kb = {}
    for counter in ("WRITE_SIZE", "FETCH_SIZE"):
        d = tempfile.mkdtemp(prefix="gort_pmc_", dir="/tmp")
        try:
            env = dict(_single_process_env(), TMPDIR="/tmp")
            r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                               capture_output=True, timeout=timeout_s)
            vals = {}
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "expand_flat_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        vals[row["Dispatch_Id"]] = vals.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not vals:
                return None, "rocprofv3 --pmc %s: no counter rows (rc %d): %s" % (counter, r.returncode, r.stderr.decode(errors="replace")[-200:])
            kb[counter] = sum(vals.values()) / len(vals)
        except Exception as ex:                                   # noqa: BLE001  (a missing profile must not cost the bench line)
            return None, "rocprofv3 --pmc %s failed: %r" % (counter, ex)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (kb["WRITE_SIZE"] + 2.0 * kb["FETCH_SIZE"]) * 1024.0, \
        ("measured in this run: two child passes of bench.py under rocprofv3 --pmc (WRITE_SIZE %.0f KB, FETCH_SIZE %.0f KB per launch; "
         "bytes = (WRITE + 2 FETCH) x 1024)" % (kb["WRITE_SIZE"], kb["FETCH_SIZE"]))


VALU_FALLBACK = os.path.join("profiles", "r05", "valu_counts.json")
# which kernel carries a config's arithmetic: (substring of the kernel's name, least grid size in threads)
VALU_KERNELS = {"C3": ("geometry_grid_kernel", 0), "C4": ("energy_kernel<true>", 0),
                "stream_1M_x_7": ("geometry_stream_kernel<true>", 500000), "stream_1M_x_100": ("stream_lines_kernel", 500000)}


def measure_valu(timeout_s=240):
    """Vector instructions per launch of the kernels behind `configs`, counted in THIS run: one child pass of this script
    (`--configs-only`: the configs block alone, two calls per shape) under `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES`.
    Returns ({config: {"valu_insts_per_launch", "waves_per_launch", "launches"}} or None, how)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="gort_valu_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run([exe, "--pmc", "SQ_INSTS_VALU", "SQ_WAVES", "--output-format", "csv", "-d", d, "--", sys.executable,
                            os.path.join(ROOT, "bench.py"), "--configs-only"], cwd="/tmp", env=env, capture_output=True, timeout=timeout_s)
        per = {}                                          # (config, dispatch) -> {counter: value}
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                for cfg, (needle, min_grid) in VALU_KERNELS.items():
                    if needle in row["Kernel_Name"].replace(", ", ",").replace("(anonymous namespace)::", "") and \
                            float(row.get("Grid_Size") or 0) >= min_grid:
                        c = per.setdefault((cfg, row["Dispatch_Id"]), {})
                        c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        out = {}
        for cfg in VALU_KERNELS:
            rows = [v for (c, _), v in per.items() if c == cfg and "SQ_INSTS_VALU" in v]
            if rows:
                out[cfg] = {"valu_insts_per_launch": sum(v["SQ_INSTS_VALU"] for v in rows) / len(rows),
                            "waves_per_launch": sum(v.get("SQ_WAVES", 0.0) for v in rows) / len(rows), "launches": len(rows)}
        if not out:
            return None, "rocprofv3 --pmc SQ_INSTS_VALU: no counter rows (rc %d): %s" % (r.returncode, r.stderr.decode(errors="replace")[-200:])
        return out, "measured in this run"
    except Exception as ex:                                       # noqa: BLE001
        return None, "rocprofv3 --pmc SQ_INSTS_VALU failed: %r" % (ex,)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def apply_valu(configs, valu, how):
    """fp64-issue fractions of the `configs` block from instruction counts: vector instructions per launch x 4 cycles over
    1024 SIMDs x 2.4 GHz x the call's time.  `valu` from measure_valu(); the committed counts of an earlier run of the same
    kernels (VALU_FALLBACK) where the counter pass was not possible - each entry says which."""
    committed = {}
    try:
        committed = json.load(open(os.path.join(ROOT, VALU_FALLBACK)))
    except Exception:                                             # noqa: BLE001
        pass
    for cfg in VALU_KERNELS:
        if cfg not in configs or not isinstance(configs[cfg], dict):
            continue
        n, src = None, None
        if valu and cfg in valu:
            n, src = valu[cfg]["valu_insts_per_launch"], "measured in this run"
            configs[cfg]["valu_waves_per_launch"] = valu[cfg]["waves_per_launch"]
        elif cfg in committed:
            n, src = committed[cfg], "committed: %s (%s)" % (VALU_FALLBACK, how)
        configs[cfg]["valu_insts_per_launch"] = n
        configs[cfg]["valu_source"] = src
        key = "frac_fp64_valu" if cfg == "C4" else "frac"
        configs[cfg][key] = None if not n else n * 4.0 / (1024 * 2.4e9 * configs[cfg]["us"] * 1e-6)


def oracle_rows(wl, grid, rows_idx, phi_idx):
    """rsurf rows of grid nodes (global row = isza * nvza + ivza, azimuth index) from the CPU restatement."""
    from oracle import oracle as O
    O.AUTO_BUILD = False                  # the prebuilt checker or nothing: no compiler runs in a bench process
    oc = O.make_canopy(lai=4.0)
    ors, orl, otl = O.spectra(wl)
    ang = np.stack([(rows_idx % grid.nvza).astype(float), phi_idx.astype(float),
                    (rows_idx // grid.nvza).astype(float), np.zeros(rows_idx.size)], 1)
    ref, _, _ = O.rsurf_stream(oc, ang, ors, orl, otl, want_K=False)
    return ref


def compare(got, ref, what):
    m = np.isfinite(ref)
    return {"max_rel_err": float(np.max(np.abs(got[m] - ref[m]) / np.maximum(np.abs(ref[m]), 1e-12))) if m.any() else 0.0,
            "nan_pattern_equal": bool(np.array_equal(np.isnan(got), np.isnan(ref))), "samples_checked": int(ref.size),
            "tolerance": 1e-5, "rows": what}


def configs_block(quick=False):
    """The other BASELINE configs and the two stream shapes a user of the reference's CLI has, timed in this run (< 1 s):
    device-resident inputs and outputs, best of 5 wall-clock times around a stream synchronisation after 0.15 s of the same
    call (clocks up).  Each with what bounds it and the achieved fraction of that bound: HBM bytes at 8 B per sample + 32 B
    per line over 8 TB/s, or fp64 VALU issue (apply_valu(): the kernels' vector instructions, counted in this run by
    measure_valu()).  quick: two calls per shape and nothing else - the child pass under the counters."""
    import torch
    from gort_amd import api

    def best(fn, eng, reps=5):
        t_up = time.perf_counter()
        while not quick and time.perf_counter() - t_up < 0.15:
            fn(); eng.synchronize()
        ts = []
        for _ in range(2 if quick else reps):
            t0 = time.perf_counter(); fn(); eng.synchronize(); ts.append(time.perf_counter() - t0)
        return min(ts)

    eng = api.Engine()
    eng.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
    out = {}
    # C2: principal plane, 181 view zeniths x 1 band (one fused launch)
    eng.set_spectra(*api.spectra([800.0]))
    ang = torch.tensor([[float(v), 0.0, 30.0, 0.0] for v in range(-90, 91)], dtype=torch.float64, device="cuda")
    o = torch.empty((181, 1), dtype=torch.float64, device="cuda")
    t = best(lambda: eng.rsurf_stream_dev(ang, o), eng)
    out["C2"] = {"workload": "181 view zeniths x 1 band, stream entry point", "us": t * 1e6, "samples_per_s": 181 / t,
                 "bound": "latency", "note": "one launch of three waves: launch + one serial chain of the geometry"}
    # C3: hemisphere x 1 band through the LUT entry point (geometry fused with the samples)
    g = api.hemisphere_grid(); rows = g.nsza * g.nvza
    lut = torch.empty((rows * g.nphi, 1), dtype=torch.float64, device="cuda")
    t = best(lambda: eng.rsurf_grid_dev(g, 0, rows, lut), eng)
    out["C3"] = {"workload": "91x91x361 angles x 1 band, LUT entry point", "us": t * 1e6, "samples_per_s": rows * g.nphi / t,
                 "bound": "fp64_valu", "frac": None, "hbm_frac": rows * g.nphi * 8 / t / 8e12}
    del lut
    # the same hemisphere as a LUT of 7 bands (MODIS land bands: the ensemble use the reference's README names) and of 100 (the band
    # counts its command line can read): up to 64 bands the geometry kernel writes the samples itself, whole rows per store; from 65
    # compact records + expand_flat_few_kernel (the aligned chunks of the headline kernel)
    for nw, key in () if quick else ((7, "lut_hemisphere_x_7"), (100, "lut_hemisphere_x_100")):      # (not under the counters: C3's kernel again)
        eng.set_spectra(*api.spectra(np.linspace(400.0, 2500.0, nw)))
        lut = torch.empty((rows * g.nphi, nw), dtype=torch.float64, device="cuda")
        t = best(lambda: eng.rsurf_grid_dev(g, 0, rows, lut), eng)
        form = "geometry kernel writes the samples" if nw <= 64 else "records + expand_flat_few_kernel"
        out[key] = {"workload": "91x91x361 angles x %d bands, LUT entry point (%s)" % (nw, form), "us": t * 1e6,
                    "samples_per_s": rows * g.nphi * nw / t, "bound": "fp64_valu" if nw < 32 else "hbm",
                    "hbm_frac": rows * g.nphi * nw * 8 / t / 8e12}
        del lut
    # C4: albedo / fAPAR table, 91 sun zeniths x 2101 bands
    wl = np.arange(400.0, 2501.0)
    eng.set_spectra(*api.spectra(wl))
    sza = torch.tensor([[0.0, 0.0, float(s), 0.0] for s in range(91)], dtype=torch.float64, device="cuda")
    en = torch.empty((91, wl.size, 3), dtype=torch.float64, device="cuda")
    t = best(lambda: eng.energy_stream_dev(sza, en), eng)
    out["C4"] = {"workload": "91 sun zeniths x 2101 bands x (albedo, fAPAR, soil absorption)", "us": t * 1e6,
                 "brdf_evaluations_per_s": 91 * 512 * wl.size / t, "bound": "latency",
                 "note": "182 workgroups (two band ranges per sun zenith) on 256 CUs: row terms split over lanes, node geometry, quadrature and band passes per workgroup",
                 "frac_fp64_valu": None}
    # the streams a user of the reference's command line has: a million lines x 7 bands (MODIS-like) and x 100 bands
    rng = np.random.default_rng(0)
    n = 1000000
    a = torch.tensor(np.stack([rng.uniform(0, 89, n), rng.uniform(0, 360, n), rng.uniform(0, 89, n), rng.uniform(0, 360, n)], 1), device="cuda")
    for nw, key in ((7, "stream_1M_x_7"), (100, "stream_1M_x_100")):
        eng.set_spectra(*api.spectra(np.linspace(400.0, 2500.0, nw)))
        o = torch.empty((n, nw), dtype=torch.float64, device="cuda")
        t = best(lambda: eng.rsurf_stream_dev(a, o), eng)
        byts = n * nw * 8 + n * 32
        out[key] = {"workload": "1 000 000 random lines x %d bands, stream entry point (%s kernel)" % (nw, eng.stream_form()),
                    "us": t * 1e6, "samples_per_s": n * nw / t, "bound": "fp64_valu", "frac": None, "hbm_frac": byts / t / 8e12}
        del o
    eng.close()
    return out


def config5_block(args, rank, world, dist, barrier):
    """BASELINE config 5 as a block of its own: 1000 canopy-parameter members sharded over the ranks (row_slab), per
    member gap probabilities + PROSPECT-D/Price + band tables + the hemisphere x 2101-band LUT (chunks, HBM resident)
    + the albedo/fAPAR table, then the ONE exchange step of the path: the RCCL all-gather of energy[members][2101][3].
    Strong scaling (the ensemble is fixed); value = samples of all members / slowest rank's time incl. the gather."""
    import torch
    from gort_amd.ensemble import sharded_albedo_table
    wl = np.arange(400.0, 2501.0)
    n = args.c5_members
    table, t = sharded_albedo_table(n, wl, rank, world, lut_chunk=args.c5_chunk, gather_on_cpu=args.rehearse, barrier=barrier,
                                    lut_slack_gib=args.lut_slack_gib, warmup_cycles=args.c5_warmup)
    red_dev = "cpu" if args.rehearse else "cuda"
    keys = ("setup_s", "lut_s", "energy_s", "gather_s", "total_s")
    mine = torch.tensor([t[k] for k in keys], dtype=torch.float64, device=red_dev)
    per_rank = None
    if world > 1:
        worst = mine.clone()
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "members": t["members"], "setup_ms": t["setup_s"] * 1e3,
                                          "lut_ms": t["lut_s"] * 1e3, "lut_chunk_ms": t["lut_chunk_ms"],
                                          "energy_ms": t["energy_s"] * 1e3, "gather_ms": t["gather_s"] * 1e3})
    else:
        worst = mine
    worst = dict(zip(keys, (float(x) for x in worst)))
    if rank != 0:
        return None
    samples = n * 91 * 361 * wl.size if args.c5_chunk else 0
    out = {"workload": "EnKF ensemble: %d members (SURVEY 8d draw, seed 12345) x hemisphere 91x361 x %d bands + albedo/fAPAR table, "
                       "members sharded over %d rank(s)" % (n, wl.size, world),
           "members": n, "lut_chunk_members": args.c5_chunk, "warmup_cycles": t.get("warmup_cycles"), "samples": samples, "scaling": "strong",
           "value": samples / worst["total_s"] if samples else None, "unit": "samples/s",
           "ms": {k[:-2] + "_ms": worst[k] * 1e3 for k in keys}, "timing": "max over ranks of each stage; total = first setup call "
           "to gathered table on every rank",
           "allgather": {"what": "energy[members][2101][3] f64, in place (gort_amd.shard)", "ms": worst["gather_s"] * 1e3,
                         "bytes_received_per_gpu": t["gather_bytes_received"],
                         "gbs_received_per_gpu": t["gather_bytes_received"] / worst["gather_s"] / 1e9 if world > 1 else None,
                         "xgmi_bound_gbs": XGMI_BOUND_GBS, "backend": "gloo (rehearsal)" if args.rehearse else ("nccl" if world > 1 else None)},
           "lut_kernel_ms_rank0": t.get("lut_kernel_ms"), "lut_alloc_rank0": t.get("lut_alloc"),
           "per_rank": per_rank}
    # parity of the exchanged product: one member of this rank and one that arrived through the all-gather
    try:
        from gort_amd.ensemble import draw_c5_members
        from oracle import oracle as O
        O.AUTO_BUILD = False
        canopies, leaf = draw_c5_members(n)
        errs = {}
        for label, m in (("own_member", 0), ("foreign_member", n - 1)):
            c, ls = canopies[m], leaf[m]
            oc = O.make_canopy(favd=c.favd, r=c.r, b=c.b, h1=c.h1, h2=c.h2, lam=c.lambda_)
            rs, rl, tl = O.spectra(wl, prospect=dict(N=ls.N, Cab=ls.Cab, Car=ls.Car, Cw=ls.Cw, Cm=ls.Cm), rsl=tuple(ls.rsl))
            ref = O.energy_stream(oc, np.array([[0.0, 0.0, 30.0, 0.0]]), rs, rl, tl)[0]
            errs[label] = dict(compare(table[m], ref, "member %d" % m))
        out["parity"] = errs
    except Exception as ex:
        out["parity"] = {"error": repr(ex)}
    return out


def kernel_spread(per_draw):
    """(max - min) / min of the draws' kernel times; None without two draws."""
    if not per_draw or len(per_draw) < 2:
        return None
    k = [x["kernel_ms"] for x in per_draw]
    return (max(k) - min(k)) / min(k) if min(k) > 0 else None


def _ranks(v):
    order = sorted(range(len(v)), key=lambda i: v[i])
    r = [0.0] * len(v)
    i = 0
    while i < len(order):
        j = i
        while j + 1 < len(order) and v[order[j + 1]] == v[order[i]]:
            j += 1
        for k in range(i, j + 1):
            r[order[k]] = (i + j) / 2.0
        i = j + 1
    return r


def probe_kernel_rank_correlation(per_draw, min_spread=0.02):
    """Spearman's rho between the allocator's probe (GB/s) and the kernel's rate on the same plain allocations; None unless
    the kernel times spread over more than `min_spread` (an argmin coincidence is not evidence)."""
    spread = kernel_spread(per_draw)
    if spread is None or spread <= min_spread:
        return None
    a, b = _ranks([x["probe_gbs"] for x in per_draw]), _ranks([-x["kernel_ms"] for x in per_draw])
    ma, mb = sum(a) / len(a), sum(b) / len(b)
    va, vb = sum((x - ma) ** 2 for x in a), sum((y - mb) ** 2 for y in b)
    if va == 0 or vb == 0:
        return None
    return sum((x - ma) * (y - mb) for x, y in zip(a, b)) / (va * vb) ** 0.5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nsza", type=int, default=91, help="sun-zenith nodes (91 = the metric grid)")
    ap.add_argument("--nw", type=int, default=2101, help="bands (2101 = the metric grid; other values are tuning experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lut-draws", type=int, default=None,
                    help="max_draws of gort_lut_alloc, the C ABI's allocator for LUT buffers: > 1 = the placement of this rank's "
                         "window is measured (1 = plain allocation).  Default at N = 1: 3 (best of three 50 GB allocations by the "
                         "store-pattern probe, >= 64 GB of the device left free) where the `per_draw` record of the run shows placements "
                         "that differ by more than 2 %, else 1 (`placement_selection` says which, DESIGN.md 5.1); 5 at N > 1 (a rank's "
                         "slab is placed by a scan through slack)")
    ap.add_argument("--placement-evidence", type=int, default=3,
                    help="N = 1: before the headline's buffer is allocated, this many plain allocations alive together (the first "
                         "draw's buffer among them) are each probed with gort_lut_alloc's store-pattern probe and written by 5 + 10 "
                         "steps of the kernel: `per_draw`")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle spot check (profiler passes)")
    ap.add_argument("--sustain-s", type=float, default=3.0,
                    help="after the timed region, keep stepping for this many seconds and report the mean step ('sustained'); 0 = off")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank dry run on ONE GPU: every rank uses cuda:0, process group gloo (not a measurement)")
    ap.add_argument("--no-gather", action="store_true", help="world > 1: skip the all-gather of the LUT after the timed steps")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 (ensemble) block")
    ap.add_argument("--c5-members", type=int, default=1000, help="members of the config-5 ensemble (1000 = BASELINE config 5)")
    ap.add_argument("--c5-chunk", type=int, default=25,
                    help="members per LUT chunk of the config-5 block (0 = no LUTs); 25 keeps a chunk's records under the 64 MB "
                         "up to which the engine overlaps the next chunk's geometry with this chunk's expansion")
    ap.add_argument("--c5-warmup", type=int, default=1,
                    help="whole cycles of the config-5 block run untimed before the timed one (a filter cycles many times; the first "
                         "cycle of a process allocates the chunks' record buffers and calibrates the XCD weights of their size class)")
    ap.add_argument("--lut-slack-gib", type=int, default=48,
                    help="world > 1: cap (GiB) of the slack gort_lut_alloc keeps beside the buffer to place this rank's window "
                         "(GORT_LUT_SLACK_GIB; 0 = plain allocation).  The timed steps are also run with 0 and 16: `slack_sweep`")
    ap.add_argument("--collective-timeout", type=float, default=240.0,
                    help="seconds a collective behind the timed region may take before it is recorded as failed; the JSON line is "
                         "printed all the same and the run exits non-zero")
    ap.add_argument("--inject-gather-error", action="store_true", help="test hook: the LUT all-gather raises")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (C2, C3, C4 and two stream shapes, < 1 s)")
    ap.add_argument("--configs-only", action="store_true",
                    help="the configs block alone, two calls per shape, one JSON line (the child pass of measure_valu under rocprofv3 --pmc)")
    ap.add_argument("--no-valu", action="store_true",
                    help="N = 1: do not count the configs' vector instructions with a rocprofv3 --pmc child pass (~40 s); the committed counts are used")
    ap.add_argument("--no-traffic", action="store_true",
                    help="N = 1: do not measure the kernel's HBM traffic with two rocprofv3 --pmc child passes (~30 s)")
    ap.add_argument("--slab-of-world", type=int, default=1,
                    help="N = 1 only: compute rank 0's slab of a world of this many ranks, in its window of that world's gatherable "
                         "buffer (the child of the N > 1 run's traffic pass: one rank, rank 0's launch)")
    ap.add_argument("--traffic-gb", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc pass (GB)")
    args = ap.parse_args()

    import torch
    from gort_amd import api

    lut_draws_default = args.lut_draws is None
    if lut_draws_default:
        args.lut_draws = 3 if int(os.environ.get("WORLD_SIZE", "1")) == 1 else 5
    if args.configs_only:
        torch.cuda.set_device(0)
        api.set_device(0)
        print(json.dumps({"configs": configs_block(quick=True)}), flush=True)
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    # --rehearse: all ranks share GPU 0 and talk over gloo (a 1-GPU box cannot run RCCL between two ranks);
    # exercises sharding, window layout, gather and the max-over-ranks timing exactly as the real multi-GPU run does.
    dev_index = 0 if args.rehearse else local
    torch.cuda.set_device(dev_index)
    api.set_device(dev_index)        # the engine, its LUT buffers and streams live on THIS rank's GPU whatever torch has initialised so far
    red_dev = "cpu" if args.rehearse else "cuda"
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        # the library's own watchdog fires well after our deadline: a hung collective must cost its own record, not the line
        pg_timeout = datetime.timedelta(seconds=max(args.collective_timeout, PARK_DEADLINE_S) + 180.0)
        if args.rehearse:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=pg_timeout)

    def barrier():
        if world > 1:
            dist.barrier()

    def reduce_max(values):
        if world == 1:
            return [float(v) for v in values]
        t = torch.tensor(list(values), dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    # Everything behind the timed region that talks to other ranks runs under a deadline, on a helper thread: an RCCL
    # error raises, a hang times out - either way the failure is RECORDED, no further collective is attempted (they would
    # hang on the same fault), rank 0 still prints its line and every rank exits non-zero.  The one multi-GPU run the
    # driver gets must not end without a line because the 50 GB all-gather threw.
    errors = []

    def guarded(what, fn, default=None, deadline=None):
        if errors:
            return default                         # the process group is not trusted any more
        import threading
        box = {}

        def run():
            try:
                torch.cuda.set_device(dev_index)       # the current device is per thread: without this every rank's helper
                api.set_device(dev_index)              # thread would talk to GPU 0
                box["value"] = fn()
            except BaseException as ex:           # noqa: BLE001
                box["error"] = repr(ex)
        th = threading.Thread(target=run, daemon=True)
        th.start()
        th.join(deadline or args.collective_timeout)
        if th.is_alive():
            errors.append({"where": what, "error": "no completion within %.0f s" % (deadline or args.collective_timeout)})
            return default
        if "error" in box:
            errors.append({"where": what, "error": box["error"]})
            return default
        return box.get("value")

    # ---- untimed setup: canopy, gap probabilities (GPU), spectra, engine ----
    wl = np.arange(400.0, 2501.0, 1.0) if args.nw == 2101 else np.linspace(400.0, 2500.0, args.nw)
    nw = wl.size
    canopy = api.gap_probabilities(api.make_canopy(lai=4.0))
    rs, rl, tl = api.spectra(wl)
    eng = api.Engine()
    eng.set_canopy(canopy)
    eng.set_spectra(rs, rl, tl)
    grid = api.hemisphere_grid(nsza=args.nsza)
    rows = grid.nsza * grid.nvza
    from gort_amd.shard import all_gather_in_place, gatherable_rows, row_slab
    layout_world = args.slab_of_world if (world == 1 and args.slab_of_world > 1) else world
    r0, r1 = row_slab(rank, layout_world, rows)
    row_elems = grid.nphi * nw
    my_samples = (r1 - r0) * row_elems
    total_samples = rows * row_elems
    # The layout that is timed IS the layout a reassembled LUT needs: every rank allocates the whole LUT (+ < world rows
    # of padding) once and computes straight into its own window of it; the all-gather after the timed steps lands in
    # place.  At world 1 the window is the buffer.
    buf_rows = gatherable_rows(layout_world, rows)
    window = (r0 * row_elems, max(r1 - r0, 0) * row_elems)

    def timed_steps(lut_ptr, warmup, steps):
        """`warmup` untimed and `steps` timed steps into lut_ptr, bracketed as the contract says; (seconds, kernel ms)."""
        for _ in range(warmup):
            if r1 > r0:
                eng.rsurf_grid_dev(grid, r0, r1, lut_ptr)
        eng.synchronize()
        eng.last_expand_ms()                      # drop warm-up launches from the kernel average
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            if r1 > r0:
                eng.rsurf_grid_dev(grid, r0, r1, lut_ptr)
        eng.synchronize()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        k = eng.last_expand_ms()                  # mean duration of the LUT expansion kernel, HIP events on its stream
        return dt, (k if k > 0 else 0.0)          # a rank without rows has no launches

    # ---- (1) a plain first allocation, timed exactly like the record: what hipMalloc's first answer is worth ----
    first = eng.lut_alloc(buf_rows * row_elems, window=window, max_draws=1)
    fd_dt, fd_kernel = timed_steps(first.at(window[0]), args.warmup, args.steps)
    # ---- (1b) N = 1: does the allocator's probe rank placements the way the kernel does?  Plain
    #           allocations alive together - the first draw's buffer is draw 0 - each probed with gort_lut_alloc's own
    #           store-pattern probe and then written by 1 + 3 steps of the real kernel.  All of them are freed before the
    #           headline's buffer is allocated (by gort_lut_alloc, which holds at most --lut-draws candidates itself) ----
    per_draw, peak_used_gb = None, None
    if world == 1 and args.placement_evidence > 0 and r1 > r0:
        held, per_draw = [first], []
        win_bytes = window[1] * 8
        for i in range(min(args.placement_evidence, 4)):
            if i > 0:
                free_b, total_b = torch.cuda.mem_get_info()
                if free_b < buf_rows * row_elems * 8 + (64 << 30):          # >= 64 GB stay free beside the draws
                    break
                held.append(eng.lut_alloc(buf_rows * row_elems, window=window, max_draws=1))
            free_b, total_b = torch.cuda.mem_get_info()
            peak_used_gb = max(peak_used_gb or 0.0, (total_b - free_b) / 1e9)
        # every draw the same treatment, draw 0 included (it has first_draw's steps behind it, the others their first touch):
        # the probe, then 5 untimed + 10 timed steps, HIP events around the 10
        for b in held:
            probe = eng.probe_store_pattern(b.at(window[0]), win_bytes)
            _, k10 = timed_steps(b.at(window[0]), 5, 10)
            per_draw.append({"probe_gbs": probe, "kernel_ms": k10})
        for b in held[1:]:
            b.free()
    if world > 1:
        first.free()                                     # before the slack sweep's allocations, as ever
        first = None
    # ---- (2) the product allocator of the C ABI (gort_lut_alloc: best of <= --lut-draws placements by a store-pattern
    #          probe of the window this rank writes); the number of record is measured on its buffer ----
    #          At world > 1 the window is placed by a scan through slack that stays allocated: the
    #          same steps are also timed with the slack capped at 0 and 16 GiB, so that the record says what the slack buys.
    slack_sweep = None
    if world > 1 and args.lut_draws > 1:
        slack_sweep = {}
        for cap in sorted({0, 16} - {args.lut_slack_gib}):
            eng.set_lut_slack_gib(cap)
            b2 = eng.lut_alloc(buf_rows * row_elems, window=window, max_draws=args.lut_draws)
            s_dt, s_k = timed_steps(b2.at(window[0]), args.warmup, args.steps)
            slack_sweep[str(cap)] = {"dt": s_dt, "kernel_ms": s_k, "slack_gb": b2.placement["slack_bytes"] / 1e9}
            b2.free()
    eng.set_lut_slack_gib(args.lut_slack_gib)
    # N = 1: the selection among whole-buffer placements is used where this box's placements DIFFER (kernel times of the plain
    # draws above more than 2 % apart: seen on some boxes of the pool, 4-11 %) and retired where they do not (the driver's records
    # of rounds 4 and 5: the selected buffer within 0.5 % of the plain first one) - decided by this run's own record, and said
    # in the line (`placement_selection`).  --lut-draws given explicitly is taken as given.
    placement_selection, buf = None, None
    if world == 1 and lut_draws_default:
        spread = kernel_spread(per_draw)
        if spread is not None and spread <= 0.02:
            args.lut_draws = 1
            placement_selection = ("retired in this run: the kernel times of %d plain allocations lie within %.2f %% of each other; the "
                                   "headline runs on the plain first allocation (first_draw's buffer)" % (len(per_draw), spread * 100))
            buf = first                                  # a plain allocation is a plain allocation: the one already there
        else:
            placement_selection = "used: " + ("no per_draw evidence" if spread is None else
                                              "the kernel times of %d plain allocations spread over %.2f %%" % (len(per_draw), spread * 100))
    if buf is None:
        if first is not None:
            first.free()
        buf = eng.lut_alloc(buf_rows * row_elems, window=window, max_draws=args.lut_draws)
    lut_ptr = buf.at(window[0])
    free_b, total_b = torch.cuda.mem_get_info()
    hbm = {"lut_buffer_gb": buf_rows * row_elems * 8 / 1e9, "placement_slack_gb": buf.placement["slack_bytes"] / 1e9,
           "device_used_gb_with_lut": (total_b - free_b) / 1e9, "device_total_gb": total_b / 1e9}
    dt, kernel_ms = timed_steps(lut_ptr, args.warmup, args.steps)

    # ---- sustained rate: the same step back to back for >= --sustain-s seconds (clocks and power settled) ----
    dt, kernel_ms_max, fd_dt, fd_kernel_max = reduce_max([dt, kernel_ms, fd_dt, fd_kernel])
    if slack_sweep:
        for cap, rec in slack_sweep.items():
            r_dt, r_k = reduce_max([rec["dt"], rec["kernel_ms"]])
            slack_sweep[cap] = {"value": total_samples * args.steps / r_dt, "ms_per_step": r_dt / args.steps * 1e3,
                                "kernel_ms_slowest_rank": r_k, "slack_gb_this_rank": rec["slack_gb"]}
        slack_sweep[str(args.lut_slack_gib)] = "the number of record (value, ms_per_step, roofline)"
    sustained = None
    if args.sustain_s > 0:
        n_sus = max(args.steps, int(args.sustain_s / max(dt / args.steps, 1e-4)) + 1)     # from the reduced dt: equal on all ranks
        sus_dt, sus_kernel = timed_steps(lut_ptr, 0, n_sus)
        sus_dt, sus_kernel = reduce_max([sus_dt, sus_kernel])
        sustained = {"seconds": sus_dt, "steps": n_sus, "ms_per_step": sus_dt / n_sus * 1e3,
                     "value": total_samples * n_sus / sus_dt, "kernel_ms": sus_kernel}

    # ---- every rank's own numbers, gathered to rank 0 ----
    mine = {"rank": rank, "rows": [r0, r1], "samples": my_samples, "kernel_ms": kernel_ms,
            "first_draw_kernel_ms": fd_kernel, "lut_alloc": buf.placement, "hbm": hbm, "xcd_mapping": eng.xcd_mapping(),
            "xcd_weights_32nds": eng.xcd_weights()[0], "store_pattern_gbs": eng.store_pattern_gbs()}
    per_rank = [mine]
    if world > 1:
        def gather_records():
            got = [None] * world
            dist.all_gather_object(got, mine)
            return got
        per_rank = guarded("all_gather_object(per-rank records)", gather_records, default=[mine])

    # ---- RCCL pre-flight through the C ABI, before the 50 GB exchanges: gort_rccl_unique_id -> gort_rccl_comm_init_rank ->
    #      gort_lut_allgather of ONE double per rank, every rank's slot checked afterwards.  An exception is recorded beside the
    #      result (the ranks agree on the outcome before anybody goes on); a hang costs the run like any other collective.
    #      A rehearsal (all ranks on one GPU, gloo) cannot run RCCL between its ranks: there every rank drives the same calls
    #      on a communicator of one.
    rccl_preflight, preflight_comm = None, {}
    if world > 1 and not args.no_gather:
        from gort_amd.shard import rccl_comm_for_group

        def preflight():
            st = {}
            try:
                t0 = time.perf_counter()
                comm = api.RcclComm(1, api.rccl_unique_id(), 0) if args.rehearse else rccl_comm_for_group()
                st["init_ms"] = (time.perf_counter() - t0) * 1e3
                slots = eng.lut_alloc(comm.world, max_draws=1)
                mine_at = comm.rank
                slots.tensor((comm.world,)).fill_(-1.0)
                slots.tensor((comm.world,))[mine_at] = float(rank + 1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.lut_allgather(slots, 1, 1, comm)
                eng.synchronize()
                st["allgather_ms"] = (time.perf_counter() - t0) * 1e3
                got = slots.to_numpy(comm.world, 0)
                want = np.array([float(rank + 1)]) if args.rehearse else np.arange(1.0, world + 1.0)
                slots.free()
                if not np.array_equal(got, want):
                    raise RuntimeError("pre-flight all-gather delivered %r, expected %r" % (got.tolist(), want.tolist()))
                if args.rehearse:
                    comm.destroy()
                else:
                    preflight_comm["comm"] = comm           # kept for the LUT's all-gather through the C ABI
            except Exception as ex:                        # noqa: BLE001
                st["error"] = repr(ex)
            bad, init_ms, ag_ms = reduce_max([1.0 if "error" in st else 0.0, st.get("init_ms", 0.0), st.get("allgather_ms", 0.0)])
            st["failed_on_some_rank"] = bad != 0.0
            st["init_ms_slowest_rank"], st["allgather_ms_slowest_rank"] = init_ms, ag_ms
            return st
        rccl_preflight = guarded("RCCL pre-flight (gort_rccl_* + gort_lut_allgather of one double per rank)", preflight)
        if rccl_preflight is None:
            rccl_preflight = {"error": errors[-1] if errors else "skipped: an earlier collective failed"}
        rccl_preflight["what"] = ("every rank its own communicator of one (rehearsal on one GPU: RCCL cannot connect ranks that share a device)"
                                  if args.rehearse else "one communicator over all ranks (the unique id travels over the process group)")

    # ---- OUTSIDE the timed step: reassemble the LUT on every rank with one in-place RCCL all-gather ----
    allgather = None
    if world > 1 and not args.no_gather:
        full_t = buf.tensor((buf_rows, row_elems))

        def gather_lut():
            torch.cuda.synchronize(); barrier()
            tg = time.perf_counter()
            if args.inject_gather_error:
                raise RuntimeError("injected by --inject-gather-error")
            all_gather_in_place(full_t, rows)
            torch.cuda.synchronize(); barrier()
            return reduce_max([time.perf_counter() - tg])[0]
        ag_s = guarded("all-gather of the LUT", gather_lut)
        received = (buf_rows - buf_rows // world) * row_elems * 8            # bytes that arrive in this GPU's HBM
        what = "the whole LUT, in place: each rank's window is its send buffer (gort_amd.shard.all_gather_in_place)"
        backend = "gloo (rehearsal on one GPU: not a measurement)" if args.rehearse else "nccl (RCCL)"
        if ag_s is None:
            allgather = {"what": what, "error": errors[-1], "bytes_received_per_gpu": received, "backend": backend,
                         "inside_timed_region": False}
        else:
            allgather = {"what": what, "ms": ag_s * 1e3, "bytes_received_per_gpu": received, "gbs_received_per_gpu": received / ag_s / 1e9,
                         "xgmi_bound_gbs": XGMI_BOUND_GBS, "frac_of_xgmi_bound": received / ag_s / 1e9 / XGMI_BOUND_GBS,
                         "backend": backend, "inside_timed_region": False}

    # ---- the same exchange once more through the C ABI (gort_rccl_* + gort_lut_allgather: librccl's ncclAllGather on the
    #      engine's stream, the path of a host without torch).  Idempotent (the windows hold what they held); a failure
    #      that raises is recorded beside the number and does not count against the run, a hang does.
    allgather_c_abi = None
    if allgather is not None and "error" not in allgather and not args.rehearse:
        from gort_amd.shard import all_gather_in_place_c_abi, rccl_comm_for_group
        state = {}

        def gather_c_abi():
            try:
                comm = preflight_comm.pop("comm", None) or rccl_comm_for_group()
                torch.cuda.synchronize(); barrier()
                tg = time.perf_counter()
                all_gather_in_place_c_abi(eng, buf, rows, row_elems, comm)
                torch.cuda.synchronize(); barrier()
                state["s"] = time.perf_counter() - tg
                comm.destroy()
            except Exception as ex:                        # noqa: BLE001
                state["error"] = repr(ex)
            # the ranks agree on the outcome BEFORE anybody reduces a time: a failure on some ranks only (dlopen, communicator)
            # must not leave the others alone in a collective.  [failed anywhere?, slowest time] in one all-reduce.
            bad, worst = reduce_max([0.0 if "s" in state else 1.0, state.get("s", 0.0)])
            state["all_ok"], state["worst"] = bad == 0.0, worst
            return True
        done = guarded("all-gather of the LUT through the C ABI", gather_c_abi)
        if done and state.get("all_ok"):
            ag2 = state["worst"]
            received = (buf_rows - buf_rows // world) * row_elems * 8
            allgather_c_abi = {"what": "gort_lut_allgather (include/gort_amd.h): ncclAllGather of librccl, in place, on the engine's stream",
                               "ms": ag2 * 1e3, "gbs_received_per_gpu": received / ag2 / 1e9,
                               "frac_of_xgmi_bound": received / ag2 / 1e9 / XGMI_BOUND_GBS}
        else:
            allgather_c_abi = {"error": state.get("error") or (errors[-1] if errors else "failed on another rank")}

    # ---- parity spot checks (outside the timed region) against the CPU oracle: rows of this rank's window and, after
    #      the gather, rows that other ranks computed (rank 1's and the last rank's windows) ----
    parity, parity_foreign = None, None
    if rank == 0 and not args.no_parity:
        try:
            rng = np.random.default_rng(5)
            idx = np.sort(rng.choice((r1 - r0) * grid.nphi, size=64, replace=False))
            got = np.stack([buf.to_numpy(nw, window[0] + int(i) * nw) for i in idx])
            parity = compare(got, oracle_rows(wl, grid, r0 + idx // grid.nphi, idx % grid.nphi), "64 nodes of rank 0's window")
            if allgather is not None and "error" not in allgather:
                picks = []
                for other in sorted({1, world - 1}):
                    o0, o1 = row_slab(other, world, rows)
                    if o1 > o0:
                        picks += [(int(r), int(l)) for r, l in zip(rng.integers(o0, o1, 12), rng.integers(0, grid.nphi, 12))]
                gr, gl = np.array([p[0] for p in picks]), np.array([p[1] for p in picks])
                got = np.stack([buf.to_numpy(nw, (int(r) * grid.nphi + int(l)) * nw) for r, l in picks])
                parity_foreign = compare(got, oracle_rows(wl, grid, gr, gl), "%d nodes from the windows of ranks %s, read on rank 0 "
                                         "after the all-gather" % (len(picks), sorted({1, world - 1})))
        except Exception as ex:
            parity = {"error": repr(ex)}

    out = None
    if rank == 0:
        value = total_samples * args.steps / dt
        # the roofline of the dominant kernel is per launch: rank 0's launch and its HIP-event duration; the slowest
        # rank's duration is quoted beside it
        per_launch_bytes = my_samples * BYTES_PER_SAMPLE
        achieved = per_launch_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
        # HBM bytes per launch of the dominant kernel come from PMC counters, which a live bench run cannot collect
        # (rocprofv3 --pmc needs passes of its own): `traffic` is a measurement of THIS code only when it is handed in
        # with --traffic-gb by the profiling script; otherwise it is null and the last committed PMC figure is quoted
        # separately, labelled as replayed, with the commit it was taken at.
        traffic, traffic_src = (args.traffic_gb * 1e9, "measured: --traffic-gb from a rocprofv3 --pmc pass of this commit") \
            if args.traffic_gb else (None, None)
        replayed = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            if pmc.get("workload") == "%dx%dx%dx%d" % (grid.nsza, grid.nvza, grid.nphi, nw) and pmc.get("n_gpus") == world:
                replayed = {"traffic": pmc["traffic_bytes"], "label": "replayed", "commit": pmc.get("commit"),
                            "note": pmc["note"]}
        out = {
            "metric": "BRDF samples/sec ((theta_v,theta_s,dphi,lambda) tuples)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "full-hemisphere x full-spectrum LUT: %dx%dx%d angles x %d bands, -LAI 4.0"
                                   % (grid.nsza, grid.nvza, grid.nphi, nw),
                       "samples_per_step": total_samples, "output_gb_per_step": total_samples * 8 / 1e9,
                       "sharding": "rows of (sun zenith, view zenith) in %d contiguous slabs (ceil partition); every rank "
                                   "computes into its window of ONE gatherable LUT buffer (%d rows, %.1f GB per GPU)"
                                   % (world, buf_rows, buf_rows * row_elems * 8 / 1e9),
                       "allocation": ("gort_lut_alloc, max_draws %d: " % args.lut_draws) +
                                     ("a plain allocation" if args.lut_draws == 1 else
                                      "the placement of the rank's window is measured (the C ABI's allocator); first_draw = a plain "
                                      "allocation, same steps; per_draw = probe and kernel time on plain allocations side by side")},
            "per_draw": per_draw,
            # does the probe order the draws as the kernel does?  Spearman's rank correlation of the probe's GB/s with the kernel's
            # rate (-kernel_ms) over the draws; null where the kernel times lie within 2 % of each other (nothing to order: the
            # noise of ten steps is ~0.5 %).  With three draws the values are -1, -0.5, 0.5, 1.
            "per_draw_probe_orders_kernel": probe_kernel_rank_correlation(per_draw),
            "per_draw_kernel_spread": kernel_spread(per_draw),
            "placement_selection": placement_selection,
            "per_draw_what": None if per_draw is None else
                             "plain allocations alive together (peak %.0f GB of the device in use), draw 0 = first_draw's buffer: "
                             "gort_lut_alloc's store-pattern probe of each, then 5 untimed + 10 timed steps of the real kernel on it (HIP "
                             "events around the 10), every draw alike.  The headline's buffer is allocated afterwards by "
                             "gort_lut_alloc(max_draws %d); with more than one draw it ranks its own candidates by that probe" % (peak_used_gb, args.lut_draws),
            "first_draw": {"value": total_samples * args.steps / fd_dt, "ms_per_step": fd_dt / args.steps * 1e3,
                           "kernel_ms_slowest_rank": fd_kernel_max,
                           "what": "the same warm-up + steps on a plain first allocation (gort_lut_alloc with max_draws 1), max over ranks"},
            "roofline": {"bound": "hbm", "kernel": "expand_flat_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "kernel_ms": kernel_ms, "kernel_ms_slowest_rank": kernel_ms_max,
                         "algorithmic_bytes_per_launch": per_launch_bytes, "launch": "rank 0's slab",
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_replayed": replayed},
            "per_rank": per_rank,
            "slack_sweep": slack_sweep,
            "sustained": sustained,
            "parity": parity,
            "reference_build": reference_build_id(),
        }
        if rccl_preflight is not None:
            out["rccl_preflight"] = rccl_preflight
        if allgather is not None:
            out["allgather"] = allgather
            out["allgather_c_abi"] = allgather_c_abi
            out["parity_after_allgather"] = parity_foreign
        elif world > 1:
            out["allgather"] = None
    buf.free()

    # ---- BASELINE config 5 as a block of its own (after the LUT buffer is gone: its chunks want the HBM) ----
    if not args.no_config5:
        c5 = guarded("config 5 block", lambda: config5_block(args, rank, world, dist, barrier), deadline=3 * args.collective_timeout)
        if rank == 0:
            out["config5"] = c5 if c5 is not None else {"error": errors[-1] if errors else "skipped: an earlier collective failed"}

    if rank == 0 and world == 1 and not args.no_configs:
        try:
            out["configs"] = configs_block()
            valu, how = (None, "--no-valu") if args.no_valu else measure_valu()
            apply_valu(out["configs"], valu, how)
            out["configs"]["valu_counts"] = how
        except Exception as ex:                           # noqa: BLE001
            out["configs"] = {"error": repr(ex)}
    # ---- what makes the line stand on its own, at every N (the CPU number "in the same run"):
    #      on rank 0, with the engine still open - the HBM traffic of rank 0's launch from the PMC counters (child passes of
    #      this script, one rank, rank 0's slab in its window) and the reference on the host's cores.  At N > 1 the other
    #      ranks are parked at a barrier with a deadline of its own meanwhile.
    if rank == 0 and args.no_traffic and args.traffic_gb is None:
        out["roofline"]["traffic_source"] = "not measured: --no-traffic"
    if rank == 0 and not errors and not args.no_traffic and args.traffic_gb is None and out["roofline"].get("algorithmic_bytes_per_launch"):
        # (the LUT buffers are gone: the child passes need the memory); a failure leaves `traffic` null and says why
        t_bytes, how = measure_traffic(args, timeout_s=150 if world == 1 else 110, world=world)
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = t_bytes, how
        if t_bytes:
            out["roofline"]["traffic_over_algorithmic"] = t_bytes / out["roofline"]["algorithmic_bytes_per_launch"]
    if rank == 0:
        if not args.no_cpu_baseline and not errors:
            # the LUT of the timed region is gone (config 5 needed the memory): the check against the reference's own rows
            # recomputes the few rows it compares, with the same entry point
            # all host cores of the box's share (16 per GPU on this pool), one reference process per core
            ncores = max(1, min(len(os.sched_getaffinity(0)), 16))
            # (1) the SAME workload shape through the reference's own gortt_rsurf (grid nodes x 2101 bands, no text)
            same = cpu_baseline_grid(wl, ncores, budget_s=12.0)
            if same is not None:
                # the reference's own rows of 24 of its nodes, all 2101 bands, against what expand_flat_kernel writes
                cn, cr = np.array(same.pop("_check_nodes")), same.pop("_check_rows")
                inside = cn[:, 0] < grid.nsza                 # reduced grids (--nsza, tests) hold only the first sun zeniths
                cn, cr = cn[inside], cr[inside]
            if same is not None and len(cn):
                mine_rows = []
                one = eng.lut_alloc(row_elems, max_draws=1)
                for isza, ivza, iphi in cn:
                    row = int(isza) * grid.nvza + int(ivza)
                    eng.rsurf_grid_dev(grid, row, row + 1, one)
                    eng.synchronize()
                    mine_rows.append(one.to_numpy(nw, int(iphi) * nw))
                one.free()
                out["parity_reference"] = dict(compare(np.stack(mine_rows), cr, "%d nodes x %d bands" % cr.shape),
                                               against="the reference's gortt_rsurf itself (oracle/_ref/libgortt_ref.so)")
            # (2) the reference as a user runs it: the CLI with a -P LUT, random lines x 180 bands, text to /dev/null
            #     (N = 1 only, like (3): the other ranks of an N > 1 run are waiting)
            cli = cpu_baseline(wl, budget_s=10.0, procs=ncores) if (world == 1 or same is None) else None
            out["cpu_baseline"] = same if same is not None else cli
            if same is not None and cli is not None:
                out["cpu_baseline_cli"] = cli
            if world == 1:
                # (3) our own hoisted scalar-C restatement (no text I/O), one core: the strongest per-core CPU number we have
                out["cpu_baseline_port"] = cpu_baseline(wl, budget_s=6.0, force_port=True)
                out["gpu_over_cpu"] = {"vs_cpu_baseline_same_shape_all_cores": out["value"] / out["cpu_baseline"]["value"],
                                       "vs_port_one_core": out["value"] / out["cpu_baseline_port"]["value"]}
            else:
                out["gpu_over_cpu"] = {"vs_cpu_baseline_same_shape_all_cores_of_rank_0s_share": out["value"] / out["cpu_baseline"]["value"]}
    if world > 1 and not (args.no_cpu_baseline and args.no_traffic):
        # the other ranks wait here while rank 0 measures (two profiler passes and ~20 s of the reference: a minute or two)
        guarded("barrier behind rank 0's traffic pass and cpu_baseline", barrier, deadline=PARK_DEADLINE_S)
    if rank == 0:
        if errors:
            out["errors"] = errors
        print(json.dumps(out), flush=True)
    if errors:
        # a collective failed or hangs on a helper thread: the line is out (rank 0), nothing here can be torn down in order.
        # The launcher (torch.distributed.run) kills every rank as soon as ONE exits non-zero: the other ranks give rank 0
        # time (a collective deadline and a minute) to get its line out before they report the failure with their exit code.
        print("bench.py: rank %d: %r" % (rank, errors), file=sys.stderr, flush=True)
        sys.stdout.flush()
        if rank != 0:
            time.sleep(args.collective_timeout + 60.0)     # rank 0 may still be waiting out a deadline of its own; its exit ends this
        os._exit(4)
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    # a headline number without its parity check is not a result: fail the run
    if rank == 0 and not args.no_parity:
        checks = [("parity", parity)]
        if world > 1 and not args.no_gather:
            checks.append(("parity_after_allgather", parity_foreign))
        if "parity_reference" in out:
            checks.append(("parity_reference", out["parity_reference"]))
        if out.get("config5") and isinstance(out["config5"].get("parity"), dict):
            for k, v in out["config5"]["parity"].items():
                checks.append(("config5." + k, v) if isinstance(v, dict) else ("config5.parity", out["config5"]["parity"]))
        for name, pr in checks:
            if pr is None or "error" in pr or not pr["nan_pattern_equal"] or not pr["max_rel_err"] <= pr["tolerance"]:
                print("bench.py: parity check %s failed or missing: %r" % (name, pr), file=sys.stderr)
                sys.exit(3)


if __name__ == "__main__":
    main()
