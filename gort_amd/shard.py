"""Row sharding of the LUT (or of an ensemble's members) across ranks and its reassembly in place.

The BRDF grid shards with no exchange step: rows (sun zenith x view zenith) are independent, and so are the
members of an ensemble.  `row_slab` is the partition bench.py and every multi-GPU caller use: a CEIL partition,
every rank but the last holds exactly `slab_rows()` rows.  That makes the full LUT gatherable IN PLACE: a rank
allocates the whole (padded) LUT once, computes its slab straight into its own window of it, and ONE
all_gather_into_tensor (RCCL over xGMI when the process group is `nccl`) fills the other windows - no receive
buffer, no second copy, no compaction pass; the padding (< world rows) sits behind the last row.
Memory per GPU: world * slab_rows rows <= rows + world - 1, i.e. 1.0009 x the LUT for the metric grid at 8 ranks
(8288 rows for 8281).  torch.distributed is plumbing here; no numerics.
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


def empty_gatherable(rows_total, row_elems, world, dtype=torch.float64, device="cpu"):
    """The whole LUT plus the partition's tail padding: [world * slab_rows, row_elems]."""
    return torch.empty((world * slab_rows(world, rows_total), row_elems), dtype=dtype, device=device)


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


def pick_fastest_slab(eng, grid, r0, r1, nw, candidates=3, steps=5, attempts=3, good_gbs=7100.0):
    """Which physical pages a LUT slab lands on decides up to 12 % of the write rate of the expansion kernel (slabs of a
    few GB are bimodal: 7.3 or 6.5 TB/s), and a multi-GPU step ends with its slowest rank.
    So, as untimed setup: allocate `candidates` slabs (alive at once: distinct placements), step each in two
    interleaved rounds, keep the fastest, free the others and VERIFY the kept one with a longer run (a probe does
    not always predict the rate once the neighbours are gone); a slab below `good_gbs` is held as a fallback while
    the draw is repeated behind a spacer allocation, at most `attempts` times.
    Returns (slab tensor [rows*nphi, nw], log: list of dicts per attempt)."""
    n = (r1 - r0) * grid.nphi
    if n <= 0 or candidates <= 1:
        return torch.empty((max(n, 0), nw), dtype=torch.float64, device="cuda"), []
    good_ms = n * nw * 8 / (good_gbs * 1e9) * 1e3
    log, best, best_ms, spacers = [], None, float("inf"), []

    def run(lut, k):
        for _ in range(k):
            eng.rsurf_grid_dev(grid, r0, r1, lut)
        eng.synchronize()
        return eng.last_expand_ms()

    for attempt in range(attempts):
        slabs = []
        for _ in range(candidates):
            try:
                slabs.append(torch.empty((n, nw), dtype=torch.float64, device="cuda"))
            except RuntimeError:                      # out of memory: make do with what we have
                break
        if not slabs:
            break
        torch.cuda.synchronize()
        for lut in slabs:                             # first touch (and, once per size class, the XCD calibration pass)
            eng.rsurf_grid_dev(grid, r0, r1, lut)
        run(slabs[0], steps)                          # clocks up before anything is compared
        ms = [0.0] * len(slabs)
        for _round in range(2):
            for i, lut in enumerate(slabs):
                ms[i] += 0.5 * run(lut, steps)
        pick = min(range(len(slabs)), key=lambda i: ms[i])
        kept = slabs[pick]
        del slabs, lut
        torch.cuda.empty_cache()
        verified = run(kept, 4 * steps)
        log.append({"probe_ms": ms, "picked": pick, "verified_ms": verified})
        if verified < best_ms:
            best, best_ms = kept, verified
        del kept
        if best_ms <= good_ms or n * nw * 8 < (1 << 30):   # small slabs have no stable rate to aim at
            break
        try:                                              # shift the next draw: the freed blocks would come back otherwise
            spacers.append(torch.empty((int(0.37e9) * (attempt + 1),), dtype=torch.float64, device="cuda"))
        except RuntimeError:
            break
    del spacers
    torch.cuda.empty_cache()
    return best, log
