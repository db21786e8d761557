"""Row sharding of the LUT (or of an ensemble's members) across ranks and its reassembly in place.

The BRDF grid shards with no exchange step: rows (sun zenith x view zenith) are independent, and so are the
members of an ensemble.  `row_slab` is the partition bench.py and every multi-GPU caller use: a CEIL partition,
every rank but the last holds exactly `slab_rows()` rows.  That makes the full LUT gatherable IN PLACE: a rank
allocates the whole (padded) LUT once, computes its slab straight into its own window of it, and ONE
all_gather_into_tensor (RCCL over xGMI when the process group is `nccl`) fills the other windows - no receive
buffer, no second copy, no compaction pass; the padding (< world rows) sits behind the last row.
Memory per GPU: world * slab_rows rows <= rows + world - 1, i.e. 1.0009 x the LUT for the metric grid at 8 ranks
(8288 rows for 8281).  torch.distributed is plumbing here; no numerics, and no placement logic: a buffer whose
physical placement matters comes from the C ABI (gort_lut_alloc -> api.Engine.lut_alloc, window = this rank's slab)
and is handed to the collectives as a zero-copy tensor view (api.LutBuffer.tensor()).
"""
import torch
import torch.distributed as dist


def slab_rows(world, rows):
    """Rows per rank of the ceil partition (the last ranks may hold fewer, down to none)."""
    if world < 1:
        raise ValueError("world %d" % world)
    return -(-rows // world)


def row_slab(rank, world, rows):
    """Contiguous rows [begin, end) of rank `rank`: begin = rank * ceil(rows / world)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    per = slab_rows(world, rows)
    begin = min(rank * per, rows)
    return begin, min(begin + per, rows)


def gatherable_rows(world, rows_total):
    """Rows of the gatherable buffer: the whole LUT plus the ceil partition's tail padding."""
    return world * slab_rows(world, rows_total)


def empty_gatherable(rows_total, row_elems, world, dtype=torch.float64, device="cpu"):
    """The whole LUT plus the partition's tail padding: [world * slab_rows, row_elems]."""
    return torch.empty((gatherable_rows(world, rows_total), row_elems), dtype=dtype, device=device)


def my_window(full_padded, rank, world, rows_total):
    """This rank's slab as a view of the gatherable buffer: compute straight into it."""
    r0, r1 = row_slab(rank, world, rows_total)
    return full_padded[r0:r1]


def all_gather_in_place(full_padded, rows_total, group=None):
    """Every rank has filled its own window of `full_padded` (empty_gatherable); returns the full
    [rows_total, row_elems] LUT as a view of the same memory, identical on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = slab_rows(world, rows_total)
    if full_padded.shape[0] != world * per or not full_padded.is_contiguous():
        raise ValueError("buffer of %d rows, expected a contiguous %d x %d" % (full_padded.shape[0], world, per))
    mine = full_padded[rank * per:(rank + 1) * per]
    if dist.get_backend(group) == "nccl":
        # in place: the send buffer is this rank's window of the receive buffer (NCCL/RCCL in-place all-gather)
        dist.all_gather_into_tensor(full_padded.view(-1), mine.reshape(-1), group=group)
    else:
        # gloo (CPU tests, one-GPU rehearsals): the same exchange through views of the same buffer; gloo does not
        # promise in-place semantics, so its input is a copy of the window (the padded window only: 1/world)
        dist.all_gather([full_padded[r * per:(r + 1) * per] for r in range(world)], mine.clone(), group=group)
    return full_padded[:rows_total]


def all_gather_lut(slab, rows_total, group=None):
    """Convenience for callers that hold their slab in a buffer of its own: one allocation of the gatherable
    buffer, one copy of the slab into its window, then the in-place gather."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    r0, r1 = row_slab(rank, world, rows_total)
    if slab.shape[0] != r1 - r0:
        raise ValueError("rank %d holds %d rows, expected %d" % (rank, slab.shape[0], r1 - r0))
    full = empty_gatherable(rows_total, slab.shape[1], world, slab.dtype, slab.device)
    full[r0:r1] = slab
    return all_gather_in_place(full, rows_total, group)


def rccl_comm_for_group(group=None):
    """An RCCL communicator of the C ABI (gort_rccl_comm_init_rank) for the ranks of a torch.distributed group: rank 0
    draws the unique id, the group - any backend - carries its 128 bytes to the others.  The calling rank's GPU must be
    the current device.  What a C host does with MPI_Bcast."""
    from . import api
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [api.rccl_unique_id() if rank == 0 else None]
    # `src` is a GLOBAL rank: the group's first member, which need not be global rank 0
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast_object_list(box, src=src, group=group)
    return api.RcclComm(world, box[0], rank)


def all_gather_in_place_c_abi(engine, buf, rows_total, row_elems, comm):
    """The same exchange as all_gather_in_place, through the C ABI (gort_lut_allgather -> ncclAllGather of librccl on the
    engine's stream): buf = the api.LutBuffer of gatherable_rows(world, rows_total) x row_elems doubles every rank
    allocated with gort_lut_alloc and filled its window of."""
    per = slab_rows(comm.world, rows_total)
    if buf.nbytes != comm.world * per * row_elems * 8:
        raise ValueError("buffer of %d bytes, expected %d x %d x %d doubles" % (buf.nbytes, comm.world, per, row_elems))
    engine.lut_allgather(buf, per, row_elems, comm)
    engine.synchronize()
