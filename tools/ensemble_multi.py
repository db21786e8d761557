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

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = 0 if args.rehearse else local
    torch.cuda.set_device(dev)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))

    from gort_amd.ensemble import c5_grid, sharded_albedo_table
    wl = np.arange(400.0, 2501.0)
    g = c5_grid()
    per_member = g.nvza * g.nphi * wl.size
    full, tt = sharded_albedo_table(args.members, wl, rank, world, lut_chunk=0 if args.no_lut else args.chunk,
                                    gather_on_cpu=args.rehearse)
    tm = torch.tensor([tt["setup_s"], tt["lut_s"], tt["gather_s"]], dtype=torch.float64, device="cpu" if args.rehearse else "cuda")
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
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
