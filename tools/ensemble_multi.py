#!/usr/bin/env python3
"""BASELINE config 5 across ranks: N ensemble members sharded over the GPUs of a node, one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P \
           tools/ensemble_multi.py [--members 1000] [--chunk 100] [--rehearse]

Each rank takes a contiguous slab of members (gort_amd.shard.row_slab), runs for its slab - on the device -
the gap probabilities, PROSPECT-D/Price, the per-member hemisphere LUT (sun zenith 30 deg, 91 x 361 view
directions x 2101 bands; 552 MB per member, never leaves its GPU) and the per-member spectral albedo /
fAPAR table.  The only exchange step is ONE all-gather (RCCL over xGMI with the nccl backend) of the reduced
product: energy[member][2101][3], 50 KB per member, so every rank ends with the ensemble's albedo table.
--rehearse: all ranks share cuda:0 and use gloo (dry run on a 1-GPU box).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def draw_members(n):
    """SURVEY.md 8(d) C5 draw: numpy default_rng(12345); HB, BR, PCC, LAI through float32."""
    from gort_amd import api
    rng = np.random.default_rng(12345)
    canopies, leaf = [], []
    for _ in range(n):
        hb, br, pcc, lai = rng.uniform(1, 3), rng.uniform(1, 3.5), rng.uniform(0.2, 0.8), rng.uniform(0.5, 6)
        cab, cw, cm = rng.uniform(10, 60), rng.uniform(0.005, 0.03), rng.uniform(0.002, 0.015)
        N, rsl1 = rng.uniform(1, 2.5), rng.uniform(0.05, 0.4)
        canopies.append(api.make_canopy(newstyle=(float(np.float32(hb)), float(np.float32(br)), float(np.float32(pcc))),
                                        lai=float(np.float32(lai))))
        leaf.append(api.leaf_soil(prospect=dict(N=N, Cab=cab, Cw=cw, Cm=cm), rsl=(rsl1, 0.1, 0.03726, -0.002426)))
    return canopies, leaf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=1000)
    ap.add_argument("--chunk", type=int, default=100, help="members per LUT launch (552 MB each)")
    ap.add_argument("--rehearse", action="store_true")
    ap.add_argument("--no-lut", action="store_true", help="skip the per-member LUTs, only the reduced product")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gort_amd import api
    from gort_amd.shard import all_gather_lut, row_slab

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = 0 if args.rehearse else local
    torch.cuda.set_device(dev)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))

    canopies, leaf = draw_members(args.members)             # every rank draws the same ensemble
    m0, m1 = row_slab(rank, world, args.members)
    wl = np.arange(400.0, 2501.0)
    g = api.Grid()
    g.sza0, g.dsza, g.nsza = 30.0, 1.0, 1
    g.vza0, g.dvza, g.nvza = 0.0, 1.0, 91
    g.phi0, g.dphi, g.nphi = 0.0, 1.0, 361
    per_member = g.nvza * g.nphi * wl.size

    eng = api.Engine()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.set_members_leaf(canopies[m0:m1], leaf[m0:m1], wl, compute_gaps=True)
    eng.synchronize()
    t_setup = time.perf_counter() - t0

    t_lut = 0.0
    if not args.no_lut:
        chunk = min(args.chunk, m1 - m0)
        lut = torch.empty((chunk, g.nvza * g.nphi, wl.size), dtype=torch.float64, device="cuda")
        t0 = time.perf_counter()
        for a in range(0, m1 - m0, chunk):
            eng.rsurf_members_grid_dev(g, a, min(m1 - m0, a + chunk), lut)
        eng.synchronize()
        t_lut = time.perf_counter() - t0

    # reduced product of my slab, then the one exchange step
    sun = torch.tensor([[0.0, 0.0, 30.0, 0.0]], dtype=torch.float64, device="cuda")
    energy = torch.empty((m1 - m0, 1, wl.size, 3), dtype=torch.float64, device="cuda")
    eng.energy_members_dev(sun, 0, m1 - m0, energy)
    eng.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    slab = energy.view(m1 - m0, wl.size * 3)
    if world > 1:
        full = all_gather_lut(slab.cpu() if args.rehearse else slab, args.members)
    else:
        full = slab
    torch.cuda.synchronize()
    t_gather = time.perf_counter() - t0
    full = full.cpu().numpy().reshape(args.members, wl.size, 3)

    tm = torch.tensor([t_setup, t_lut, t_gather], dtype=torch.float64, device="cpu" if args.rehearse else "cuda")
    if world > 1:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    if rank == 0:
        t_setup, t_lut, t_gather = [float(x) for x in tm]
        closure = float(np.abs(full.sum(axis=2) - 1.0).max())      # albedo + favegt + fasoil = 1 for every member, band
        print("ranks %d, members %d: setup %.1f ms, LUTs %.1f ms (%.3e samples, %.3e samples/s aggregate), "
              "all-gather of the %d x 2101 x 3 albedo table %.2f ms; energy closure max|sum-1| = %.2e; "
              "albedo(800 nm) of members 0..3: %s"
              % (world, args.members, t_setup * 1e3, t_lut * 1e3, args.members * per_member,
                 (args.members * per_member / t_lut) if t_lut else float("nan"), args.members, t_gather * 1e3, closure,
                 np.array2string(full[:4, 400, 0], precision=6)))
        np.save(os.path.join(os.environ.get("GORT_OUT", "/tmp"), "ensemble_energy_w%d.npy" % world), full)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
