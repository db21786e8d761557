"""Row sharding of the LUT across ranks and its reassembly.

The BRDF grid shards with no exchange step: rows (sun zenith x view zenith) are independent.
`row_slab` is the partition bench.py and any multi-GPU caller use; `all_gather_lut` reassembles
the full LUT on every rank with ONE all-gather (RCCL over xGMI when the process group is `nccl`,
gloo in the CPU tests).  torch.distributed is plumbing here; no numerics.
"""
import torch
import torch.distributed as dist


def row_slab(rank, world, rows):
    """Contiguous rows [begin, end) of rank `rank`; slabs differ by at most one row."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return rows * rank // world, rows * (rank + 1) // world


def all_gather_lut(slab, rows_total, group=None):
    """slab: this rank's [rows_local, row_elems] tensor (row_slab order).  Returns the full
    [rows_total, row_elems] tensor on every rank.  Slabs are padded to a common row count so that a single
    all_gather_into_tensor moves them; the padding rows are dropped while compacting."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    r0, r1 = row_slab(rank, world, rows_total)
    if slab.shape[0] != r1 - r0:
        raise ValueError("rank %d holds %d rows, expected %d" % (rank, slab.shape[0], r1 - r0))
    row_elems = slab.shape[1]
    max_rows = max(row_slab(r, world, rows_total)[1] - row_slab(r, world, rows_total)[0] for r in range(world))
    if slab.shape[0] == max_rows:
        send = slab.contiguous()
    else:
        send = torch.zeros((max_rows, row_elems), dtype=slab.dtype, device=slab.device)
        send[: slab.shape[0]] = slab
    recv = torch.empty((world, max_rows, row_elems), dtype=slab.dtype, device=slab.device)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    full = torch.empty((rows_total, row_elems), dtype=slab.dtype, device=slab.device)
    for r in range(world):
        a, b = row_slab(r, world, rows_total)
        full[a:b] = recv[r, : b - a]
    return full
