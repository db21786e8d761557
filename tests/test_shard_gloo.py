"""N>1 path on CPU: row / member sharding and the in-place LUT all-gather with the gloo backend, world sizes 2, 3
and 8 (uneven and empty slabs).  The kernels themselves need a GPU; what is covered here is everything that
makes a multi-rank run correct by construction: the partition, the windows of the gatherable buffer and the
reassembly (gort_amd/shard.py), and the member-sharded exchange of an ensemble's reduced product
(gort_amd/ensemble.py: gather_member_tables)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gort_amd.shard import all_gather_in_place, all_gather_lut, empty_gatherable, my_window, row_slab, slab_rows


def test_row_slab_is_a_ceil_partition():
    for rows in (1, 7, 10, 91, 1000, 8281):
        for world in (1, 2, 3, 4, 8):
            per = slab_rows(world, rows)
            assert per == -(-rows // world)
            slabs = [row_slab(r, world, rows) for r in range(world)]
            assert slabs[0][0] == 0 and slabs[-1][1] == rows
            for (a0, a1), (b0, b1) in zip(slabs, slabs[1:]):
                assert a1 == b0 and a0 <= a1
            # every rank's window starts at rank * per: that is what lets the all-gather land in place
            for r, (a, b) in enumerate(slabs):
                assert a == min(r * per, rows) and b - a <= per
            assert world * per - rows < world                      # padding: fewer than `world` rows
    assert [row_slab(r, 8, 8281) for r in (0, 6, 7)] == [(0, 1036), (6216, 7252), (7252, 8281)]
    assert row_slab(7, 8, 10) == (10, 10)                          # more ranks than work: empty slabs at the end
    with pytest.raises(ValueError):
        row_slab(2, 2, 10)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pattern(r0, r1, row_elems):
    # every element encodes its global (row, column): the gathered LUT must be the identity pattern
    return (torch.arange(r0, r1, dtype=torch.float64)[:, None] * 1000.0
            + torch.arange(row_elems, dtype=torch.float64)[None, :])


def _worker(rank, world, port, rows, row_elems, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r0, r1 = row_slab(rank, world, rows)
        want = _pattern(0, rows, row_elems)
        # (1) the in-place form bench.py --gather uses: compute into the own window of ONE buffer, gather there
        full = empty_gatherable(rows, row_elems, world)
        full.fill_(-1.0)
        my_window(full, rank, world, rows).copy_(_pattern(r0, r1, row_elems))
        got = all_gather_in_place(full, rows)
        ok1 = bool(torch.equal(got, want)) and got.data_ptr() == full.data_ptr()
        ok1 = ok1 and full.shape[0] == world * slab_rows(world, rows)
        # (2) the convenience form for a slab held elsewhere
        ok2 = bool(torch.equal(all_gather_lut(_pattern(r0, r1, row_elems), rows), want))
        # (3) an ensemble's reduced product: members sharded, tables [members_local][bands][3]
        from gort_amd.ensemble import gather_member_tables
        tab = _pattern(r0, r1, 2 * 3).view(r1 - r0, 2, 3)
        ok3 = bool(torch.equal(gather_member_tables(tab, rows), _pattern(0, rows, 6).view(rows, 2, 3)))
        # (4) bench.py's bookkeeping around it: buffer row == global row (so rows of ANOTHER rank's window can be read
        # by their global index after the gather), every rank's record on every rank, MAX over ranks of a timing
        from gort_amd.shard import gatherable_rows
        ok4 = full.shape[0] == gatherable_rows(world, rows)
        for other in sorted({min(1, world - 1), world - 1}):
            o0, o1 = row_slab(other, world, rows)
            ok4 = ok4 and bool(torch.equal(full[o0:o1], want[o0:o1]))
        recs = [None] * world
        dist.all_gather_object(recs, {"rank": rank, "rows": [r0, r1]})
        ok4 = ok4 and [r["rank"] for r in recs] == list(range(world))
        ok4 = ok4 and recs[0]["rows"][0] == 0 and recs[-1]["rows"][1] == rows
        ok4 = ok4 and all(recs[i]["rows"][1] == recs[i + 1]["rows"][0] for i in range(world - 1))
        t = torch.tensor([1.0 + rank, 5.0 - rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok4 = ok4 and t.tolist() == [float(world), 5.0]
        q.put((rank, ok1, ok2, ok3 and ok4))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,rows", [(2, 7), (3, 10), (8, 10)])
def test_all_gather_in_place_gloo(world, rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, rows, 13, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == [(r, True, True, True) for r in range(world)]
