"""N>1 path on CPU: row sharding + LUT all-gather with the gloo backend, world_size 2 and 3
(uneven slabs).  The kernels themselves need a GPU; what is covered here is everything that
makes a multi-rank run correct by construction: the partition and the reassembly."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gort_amd.shard import all_gather_lut, row_slab


def test_row_slab_partitions_exactly():
    for rows in (1, 7, 91, 8281):
        for world in (1, 2, 3, 4, 8):
            slabs = [row_slab(r, world, rows) for r in range(world)]
            assert slabs[0][0] == 0 and slabs[-1][1] == rows
            for (a0, a1), (b0, b1) in zip(slabs, slabs[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in slabs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        row_slab(2, 2, 10)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, rows, row_elems, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r0, r1 = row_slab(rank, world, rows)
        # every element encodes its global (row, column): the gathered LUT must be the identity pattern
        slab = (torch.arange(r0, r1, dtype=torch.float64)[:, None] * 1000.0
                + torch.arange(row_elems, dtype=torch.float64)[None, :])
        full = all_gather_lut(slab, rows)
        want = (torch.arange(rows, dtype=torch.float64)[:, None] * 1000.0
                + torch.arange(row_elems, dtype=torch.float64)[None, :])
        q.put((rank, bool(torch.equal(full, want))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,rows", [(2, 7), (2, 8), (3, 10)])
def test_all_gather_lut_gloo(world, rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, rows, 13, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == [(r, True) for r in range(world)]
