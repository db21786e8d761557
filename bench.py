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
N contiguous slabs, total work fixed (strong scaling); no collective on the data path -
see DESIGN.md "Multi-GPU" for why the 50 GB LUT is not all-gathered inside the step.

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nsza", type=int, default=91, help="sun-zenith nodes (91 = the metric grid)")
    ap.add_argument("--nw", type=int, default=2101, help="bands (2101 = the metric grid; other values are tuning experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--slab-candidates", type=int, default=3,
                    help="untimed setup: allocate this many LUT slabs, keep the one the expansion kernel writes fastest, "
                         "verify it, redraw up to 3 times if it is a slow placement (1 = off)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle spot check (profiler passes)")
    ap.add_argument("--sustain-s", type=float, default=3.0,
                    help="after the timed region, keep stepping for this many seconds and report the mean step ('sustained'); 0 = off")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank dry run on ONE GPU: every rank uses cuda:0, process group gloo (not a measurement)")
    ap.add_argument("--gather", action="store_true",
                    help="after the timed steps, all-gather the full LUT on every rank (RCCL) and report allgather_ms")
    ap.add_argument("--traffic-gb", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc pass (GB)")
    args = ap.parse_args()

    import torch
    from gort_amd import api

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    # --rehearse: all ranks share GPU 0 and talk over gloo (a 1-GPU box cannot run RCCL between two ranks);
    # exercises sharding, slab sizing and the max-over-ranks timing exactly as the real multi-GPU run does.
    dev_index = 0 if args.rehearse else local
    torch.cuda.set_device(dev_index)
    red_dev = "cpu" if args.rehearse else "cuda"
    if world > 1:
        import torch.distributed as dist
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))

    def barrier():
        if world > 1:
            dist.barrier()

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
    from gort_amd.shard import all_gather_in_place, empty_gatherable, my_window, row_slab
    r0, r1 = row_slab(rank, world, rows)
    my_samples = (r1 - r0) * grid.nphi * nw
    total_samples = rows * grid.nphi * nw
    slab_ms = []
    if args.gather and world > 1:
        # the whole LUT once (+ < world rows of padding): this rank computes straight into its window of it and
        # the all-gather lands in place - no receive buffer, no second copy (gort_amd/shard.py)
        full_padded = empty_gatherable(rows, grid.nphi * nw, world, torch.float64, "cuda")
        lut = my_window(full_padded, rank, world, rows).view((r1 - r0) * grid.nphi, nw)
    else:
        # untimed setup: the slab is the fastest of a few allocations (physical placement decides +-5 % of the
        # expansion kernel's write rate and a multi-GPU step ends with its slowest rank)
        from gort_amd.shard import pick_fastest_slab
        lut, slab_ms = pick_fastest_slab(eng, grid, r0, r1, nw, candidates=args.slab_candidates)

    def step():
        if r1 > r0:
            eng.rsurf_grid_dev(grid, r0, r1, lut)

    for _ in range(args.warmup):
        step()
    eng.synchronize()
    eng.last_expand_ms()                      # drop warm-up launches from the kernel average
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.synchronize()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = eng.last_expand_ms()          # mean duration of the LUT expansion kernel, HIP events on its stream
    if kernel_ms < 0:
        kernel_ms = 0.0                       # a rank without rows

    # ---- sustained rate: the same step back to back for >= --sustain-s seconds (clocks and power settled) ----
    sustained = None
    if args.sustain_s > 0:
        n_sus = max(args.steps, int(args.sustain_s / max(dt / args.steps, 1e-4)) + 1)
        barrier()
        ts = time.perf_counter()
        for _ in range(n_sus):
            step()
        eng.synchronize()
        torch.cuda.synchronize()
        barrier()
        sus_dt = time.perf_counter() - ts
        sus_kernel = eng.last_expand_ms()
        if world > 1:
            t = torch.tensor([sus_dt, max(sus_kernel, 0.0)], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sus_dt, sus_kernel = float(t[0]), float(t[1])
        sustained = {"seconds": sus_dt, "steps": n_sus, "ms_per_step": sus_dt / n_sus * 1e3,
                     "value": total_samples * n_sus / sus_dt, "kernel_ms": sus_kernel}

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        k = torch.tensor([kernel_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        kernel_ms = float(k.item())

    # ---- optional, OUTSIDE the timed step: reassemble the LUT on every rank with one RCCL all-gather ----
    allgather_ms = None
    if args.gather and world > 1:
        torch.cuda.synchronize(); barrier()
        tg = time.perf_counter()
        try:
            full = all_gather_in_place(full_padded, rows)
            torch.cuda.synchronize(); barrier()
            allgather_ms = (time.perf_counter() - tg) * 1e3
            del full
        except torch.OutOfMemoryError:
            # only a rehearsal can end here: N full LUTs on ONE GPU, gathered through gloo's staging copies
            if not args.rehearse:
                raise
            print("bench.py: --rehearse --gather does not fit one GPU at this size; gather skipped", file=sys.stderr)
            allgather_ms = None
            barrier()                                       # the ranks that got through are waiting in theirs

    # ---- parity spot check (outside the timed region): sampled rows vs the CPU oracle ----
    parity = None
    if rank == 0 and not args.no_parity:
        try:
            from oracle import oracle as O
            O.AUTO_BUILD = False                  # the prebuilt checker or nothing: no compiler runs in a bench process
            oc = O.make_canopy(lai=4.0)
            ors, orl, otl = O.spectra(wl)
            rng = np.random.default_rng(5)
            idx = np.sort(rng.choice((r1 - r0) * grid.nphi, size=64, replace=False))
            got = lut[torch.as_tensor(idx, device="cuda")].cpu().numpy()
            row = r0 + idx // grid.nphi
            ang = np.stack([(row % grid.nvza).astype(float), (idx % grid.nphi).astype(float),
                            (row // grid.nvza).astype(float), np.zeros(idx.size)], 1)
            ref, _, _ = O.rsurf_stream(oc, ang, ors, orl, otl, want_K=False)
            nan_ok = bool(np.array_equal(np.isnan(got), np.isnan(ref)))
            m = np.isfinite(ref)
            parity = {"max_rel_err": float(np.max(np.abs(got[m] - ref[m]) / np.maximum(np.abs(ref[m]), 1e-12))),
                      "nan_pattern_equal": nan_ok, "samples_checked": int(ref.size), "tolerance": 1e-5}
        except Exception as ex:
            parity = {"error": str(ex)}

    if rank == 0:
        value = total_samples * args.steps / dt
        per_launch_bytes = my_samples * BYTES_PER_SAMPLE
        achieved = per_launch_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms and kernel_ms > 0 else None
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
                       "sharding": "rows of (sun zenith, view zenith) split in %d contiguous slabs" % world},
            "roofline": {"bound": "hbm", "kernel": "expand_flat_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": per_launch_bytes,
                         "xcd_mapping": eng.xcd_mapping(), "xcd_weights_32nds": eng.xcd_weights()[0],
                         "bare_store_pattern_gbs_equal_xcd_shares": eng.store_pattern_gbs(),
                         "slab_selection_rank0": slab_ms,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_replayed": replayed},
            "sustained": sustained,
            "parity": parity,
        }
        if allgather_ms is not None:
            out["allgather_ms"] = allgather_ms
        if world == 1 and not args.no_cpu_baseline:
            # all host cores of the box's share (16 per GPU on this pool), one reference process per core
            ncores = max(1, min(len(os.sched_getaffinity(0)), 16))
            # (1) the SAME workload shape through the reference's own gortt_rsurf (grid nodes x 2101 bands, no text)
            same = cpu_baseline_grid(wl, ncores, budget_s=12.0)
            if same is not None:
                # the reference's own rows of 24 of its nodes, all 2101 bands, against the LUT this run wrote
                cn, cr = np.array(same.pop("_check_nodes")), same.pop("_check_rows")
                inside = cn[:, 0] < grid.nsza                 # reduced grids (--nsza, tests) hold only the first sun zeniths
                cn, cr = cn[inside], cr[inside]
            if same is not None and len(cn):
                at = (cn[:, 0] * grid.nvza + cn[:, 1]) * grid.nphi + cn[:, 2]
                mine = lut[torch.as_tensor(at, device="cuda")].cpu().numpy()
                fin = np.isfinite(cr)
                out["parity_reference"] = {
                    "max_rel_err": float(np.max(np.abs(mine[fin] - cr[fin]) / np.maximum(np.abs(cr[fin]), 1e-12))),
                    "nan_pattern_equal": bool(np.array_equal(np.isnan(mine), np.isnan(cr))),
                    "samples_checked": int(cr.size), "tolerance": 1e-5,
                    "against": "the reference's gortt_rsurf itself (oracle/_ref/libgortt_ref.so), %d nodes x %d bands" % cr.shape}
            # (2) the reference as a user runs it: the CLI with a -P LUT, random lines x 180 bands, text to /dev/null
            cli = cpu_baseline(wl, budget_s=10.0, procs=ncores)
            out["cpu_baseline"] = same if same is not None else cli
            if same is not None:
                out["cpu_baseline_cli"] = cli
            # (3) our own hoisted scalar-C restatement (no text I/O), one core: the strongest per-core CPU number we have
            out["cpu_baseline_port"] = cpu_baseline(wl, budget_s=6.0, force_port=True)
            out["gpu_over_cpu"] = {"vs_cpu_baseline_same_shape_all_cores": value / out["cpu_baseline"]["value"],
                                   "vs_port_one_core": value / out["cpu_baseline_port"]["value"]}
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    # a headline number without its parity check is not a result: fail the run
    if rank == 0 and not args.no_parity:
        bad = parity is None or "error" in parity or not parity["nan_pattern_equal"] or not parity["max_rel_err"] <= parity["tolerance"]
        pr = out.get("parity_reference")
        if pr is not None and (not pr["nan_pattern_equal"] or not pr["max_rel_err"] <= pr["tolerance"]):
            bad = True
            parity = pr
        if bad:
            print("bench.py: parity check failed or missing: %r" % (parity,), file=sys.stderr)
            sys.exit(3)


if __name__ == "__main__":
    main()
