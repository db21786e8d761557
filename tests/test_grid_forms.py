"""The two partitions of geometry_grid_kernel (gort_amd/csrc/gort_geometry.hip) must write the SAME BITS:

* single-member launches are partitioned by NODES into the machine's workgroup slots (round 4), a workgroup evaluating the
  row terms of whatever rows its span touches;
* everything else - and everything under GORT_GRID_BY_ROWS=1 - by ROWS (4 / 6 / 8 per workgroup).

Grids of odd shapes, slabs that begin inside a grid (a rank's window), mirrored and unmirrored azimuths, one band (the
fused form, BASELINE config 3), a few bands, and enough bands for the record + expansion path.
Reference: the triple loop a user would script around gortt's stdin (gortt.c:232-329); SURVEY.md 8(d) C3."""
import os

import numpy as np
import pytest

from gort_amd import api

pytestmark = pytest.mark.gpu


def _grid(sza, vza, phi):
    g = api.Grid()
    g.sza0, g.dsza, g.nsza = sza
    g.vza0, g.dvza, g.nvza = vza
    g.phi0, g.dphi, g.nphi = phi
    return g


GRIDS = {
    "hemisphere_13x17": ((0.0, 7.5, 13), (0.0, 5.0, 17), (0.0, 1.0, 361)),           # mirrored, incl. both horizons (90, 85 + 5)
    "one_row": ((30.0, 1.0, 1), (40.0, 1.0, 1), (0.0, 1.0, 361)),
    "three_nodes": ((10.0, 20.0, 4), (0.0, 30.0, 3), (0.0, 180.0, 3)),               # rows of two evaluated nodes
    "unmirrored": ((5.0, 11.0, 7), (2.0, 9.0, 9), (3.0, 7.0, 50)),                   # phi0 != 0: every node evaluated
    "long_rows": ((20.0, 30.0, 2), (10.0, 35.0, 3), (0.0, 0.25, 1441)),              # a span shorter than a row
}


@pytest.mark.parametrize("name", sorted(GRIDS))
@pytest.mark.parametrize("nw", [1, 3, 40])
def test_node_partition_equals_row_partition(name, nw):
    import torch
    g = _grid(*GRIDS[name])
    rows = g.nsza * g.nvza
    wl = np.linspace(450.0, 2400.0, nw)
    e = api.Engine()
    e.set_canopy(api.gap_probabilities(api.make_canopy(lai=3.1)))
    e.set_spectra(*api.spectra(wl))
    windows = [(0, rows)] + ([(rows // 3, rows - 1)] if rows > 3 else [])
    for r0, r1 in windows:
        out = {}
        for by_rows in ("0", "1"):
            os.environ["GORT_GRID_BY_ROWS"] = by_rows
            try:
                buf = torch.full(((r1 - r0) * g.nphi * nw + 16,), -7.0, dtype=torch.float64, device="cuda")
                lut = buf[8:8 + (r1 - r0) * g.nphi * nw]
                e.rsurf_grid_dev(g, r0, r1, lut)
                e.synchronize()
            finally:
                os.environ.pop("GORT_GRID_BY_ROWS")
            assert float(buf[:8].min()) == -7.0 and float(buf[-8:].max()) == -7.0
            out[by_rows] = lut.cpu().numpy()
        assert not (out["0"] == -7.0).any()
        assert np.array_equal(out["0"].view(np.int64), out["1"].view(np.int64)), (name, nw, r0, r1)
    e.close()


def test_one_member_of_an_ensemble_takes_the_node_partition_with_its_own_canopy():
    """A launch that covers exactly member 1 of three is a single-member launch (partitioned by nodes, its canopy and band
    table read once): the same bits as that member's window of the launch over all three (partitioned by rows)."""
    import torch
    g = _grid((0.0, 15.0, 5), (0.0, 10.0, 8), (0.0, 2.0, 181))
    rows = g.nsza * g.nvza
    wl = np.linspace(400.0, 2500.0, 130)                      # member grids need >= 128 bands (records + LUT kernel)
    members = [api.gap_probabilities(api.make_canopy(lai=x)) for x in (1.5, 3.0, 4.5)]
    spectra = np.stack([np.stack(api.spectra(wl)) * f for f in (1.0, 0.9, 0.8)])
    e = api.Engine()
    e.set_members(members, spectra)
    per = rows * g.nphi * wl.size
    every = torch.empty((3 * per,), dtype=torch.float64, device="cuda")
    e.rsurf_members_grid_dev(g, 0, 3, every)
    one = torch.full((per + 8,), -7.0, dtype=torch.float64, device="cuda")
    e.rsurf_members_grid_dev(g, 1, 2, one[:per])
    e.synchronize()
    assert float(one[per:].max()) == -7.0
    a, b = one[:per].cpu().numpy(), every[per:2 * per].cpu().numpy()
    assert np.array_equal(a.view(np.int64), b.view(np.int64))
    assert not np.array_equal(a, every[:per].cpu().numpy())
    e.close()
