"""The two partitions of geometry_grid_kernel (gort_amd/csrc/gort_geometry.hip) must write the SAME BITS:

* single-member launches are partitioned by NODES into the machine's workgroup slots (round 4), a workgroup evaluating the
  row terms of whatever rows its span touches;
* everything else - and everything under GORT_GRID_BY_ROWS=1 - by ROWS (4 / 6 / 8 per workgroup);
* with the azimuth table (round 5: raa, sin, cos per azimuth node in LDS, filled by the waves that wait for the row terms)
  and with every node forming its own (GORT_GRID_AZ_TABLE=0).

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
    "long_rows": ((20.0, 30.0, 2), (10.0, 35.0, 3), (0.0, 0.25, 1441)),              # a span shorter than a row; too long for the azimuth table
    "negative_zeniths": ((-20.0, 20.0, 3), (-30.0, 30.0, 3), (0.0, 45.0, 9)),        # azimuths turned by pi row by row: no azimuth table
}


@pytest.mark.ab
@pytest.mark.parametrize("name", sorted(GRIDS))
@pytest.mark.parametrize("nw", [1, 3, 40, 100])
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
        for by_rows, az in (("0", "1"), ("1", "1"), ("0", "0"), ("1", "0")):
            os.environ["GORT_GRID_BY_ROWS"] = by_rows
            os.environ["GORT_GRID_AZ_TABLE"] = az
            try:
                buf = torch.full(((r1 - r0) * g.nphi * nw + 16,), -7.0, dtype=torch.float64, device="cuda")
                lut = buf[8:8 + (r1 - r0) * g.nphi * nw]
                torch.cuda.synchronize()        # torch fills on ITS stream; the engine works on its own (non-blocking) one
                e.rsurf_grid_dev(g, r0, r1, lut)
                e.synchronize()
            finally:
                os.environ.pop("GORT_GRID_BY_ROWS")
                os.environ.pop("GORT_GRID_AZ_TABLE")
            assert float(buf[:8].min()) == -7.0 and float(buf[-8:].max()) == -7.0
            out[by_rows + az] = lut.cpu().numpy()
        assert not (out["01"] == -7.0).any()
        for other in ("11", "00", "10"):
            a, b = out["01"], out[other]
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a.view(np.int64)[~np.isnan(a)], b.view(np.int64)[~np.isnan(b)]), (name, nw, r0, r1, other)
    e.close()


def test_one_member_of_an_ensemble_takes_the_node_partition_with_its_own_canopy():
    """A launch that covers exactly member 1 of three is a single-member launch (partitioned by nodes, its canopy and band
    table read once): the same bits as that member's window of the launch over all three (partitioned by rows)."""
    import torch
    g = _grid((0.0, 15.0, 5), (0.0, 10.0, 8), (0.0, 2.0, 181))
    rows = g.nsza * g.nvza
    wl = np.linspace(400.0, 2500.0, 130)                      # records + LUT kernel
    members = [api.gap_probabilities(api.make_canopy(lai=x)) for x in (1.5, 3.0, 4.5)]
    spectra = np.stack([np.stack(api.spectra(wl)) * f for f in (1.0, 0.9, 0.8)])
    e = api.Engine()
    e.set_members(members, spectra)
    per = rows * g.nphi * wl.size
    every = torch.empty((3 * per,), dtype=torch.float64, device="cuda")
    e.rsurf_members_grid_dev(g, 0, 3, every)
    one = torch.full((per + 8,), -7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    e.rsurf_members_grid_dev(g, 1, 2, one[:per])
    e.synchronize()
    assert float(one[per:].max()) == -7.0
    a, b = one[:per].cpu().numpy(), every[per:2 * per].cpu().numpy()
    assert np.array_equal(a.view(np.int64), b.view(np.int64))
    assert not np.array_equal(a, every[:per].cpu().numpy())
    e.close()


def test_q08_canopy_on_the_horizon_matches_the_reference_restatement():
    """The closed-form gap probabilities of -q08_pn_kopen leave the reference FINITE reflectances at a zenith of 90 degrees
    (the tabulated ones make it NaN there): lines, LUT nodes and albedo lines of such a canopy take the reference's route on
    the horizon whatever is asked for (gort_geometry.h, takes_reference_route).  Grid (node partition, 1 band fused and 130
    bands through records + the LUT kernel), stream with and without the viewed proportions, and the albedo quadrature with
    the sun on the horizon, against the oracle on the same gap tables."""
    import torch
    from conftest import relerr
    from oracle import oracle as O
    c = api.gap_probabilities(api.make_canopy(lai=2.0, q08=True))
    o = O.make_canopy(favd=c.favd, r=c.r, b=c.b, h1=c.h1, h2=c.h2, lam=c.lambda_, gaps=False)
    O.set_gap_tables(o, np.array(c.p_n0), np.array(c.epgap), c.k_open, c.k_openep)     # the same gap tables: isolates the BRDF kernels
    g = _grid((0.0, 45.0, 3), (0.0, 30.0, 4), (0.0, 90.0, 5))             # sza 0 45 90; vza 0 30 60 90; full circle in 90-degree steps
    rows = g.nsza * g.nvza
    ang = np.array([[g.vza0 + (r % g.nvza) * g.dvza, g.phi0 + l * g.dphi, g.sza0 + (r // g.nvza) * g.dsza, 0.0]
                    for r in range(rows) for l in range(g.nphi)])
    at90 = (ang[:, 0] == 90.0) | (ang[:, 2] == 90.0)
    for nw in (1, 130):
        wl = np.linspace(500.0, 2300.0, nw)
        rs, rl, tl = api.spectra(wl)
        want = O.rsurf_stream(o, ang, rs, rl, tl, want_K=False)[0]
        assert np.isfinite(want[at90]).any()                             # the point of the test: numbers on the horizon
        e = api.Engine()
        e.set_canopy(c)
        e.set_spectra(rs, rl, tl)
        lut = torch.empty((rows * g.nphi, nw), dtype=torch.float64, device="cuda")
        e.rsurf_grid_dev(g, 0, rows, lut)
        a = torch.as_tensor(ang, device="cuda")
        plain = torch.empty((ang.shape[0], nw), dtype=torch.float64, device="cuda")
        with_k = torch.empty_like(plain)
        K = torch.empty((ang.shape[0], 4), dtype=torch.float64, device="cuda")
        e.rsurf_stream_dev(a, plain)
        e.rsurf_stream_dev(a, with_k, None, K)
        e.synchronize()
        for name, got in (("grid", lut.cpu().numpy()), ("stream", plain.cpu().numpy()), ("stream with K", with_k.cpu().numpy())):
            assert np.array_equal(np.isnan(got), np.isnan(want)), (name, nw)
            assert relerr(got, want, floor=1e-12) <= 1e-9, (name, nw, relerr(got, want, floor=1e-12))
        e.close()
    # the albedo quadrature with the sun on the horizon (and beside it)
    wl = np.array([450.0, 800.0, 1650.0])
    rs, rl, tl = api.spectra(wl)
    sun = np.array([[0.0, 0.0, s, 0.0] for s in (90.0, 89.0, 30.0, -90.0)])
    want = O.energy_stream(o, sun, rs, rl, tl)
    e = api.Engine()
    e.set_canopy(c)
    e.set_spectra(rs, rl, tl)
    got = e.energy_stream(sun)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert relerr(got, want, floor=1e-12) <= 1e-9, relerr(got, want, floor=1e-12)
    e.close()


@pytest.mark.parametrize("nw", [1, 7, 9, 16, 64, 100, 127, 128, 300])
def test_a_lut_equals_the_stream_of_its_nodes_to_rounding(nw):
    """include/gort_amd.h: a LUT of any band count carries the five-term form of the sample, the stream entry points the form
    grouped around two other terms - the LUT's nodes streamed as "vza phi sza 0" (gortt.c:232-329) agree to 1e-12, NaN for NaN
    (the hemisphere's last view zenith is the horizon), mirrored images included.  Below 128 bands the geometry kernel writes the
    LUT itself (up to 8 bands a lane its node's samples, from 9 lanes as bands), from 128 the LUT kernel expands records."""
    import torch
    from conftest import relerr
    g = _grid((0.0, 22.5, 4), (0.0, 15.0, 7), (0.0, 2.0, 181))           # a full circle: mirrored
    rows = g.nsza * g.nvza
    e = api.Engine()
    e.set_canopy(api.gap_probabilities(api.make_canopy(lai=2.7)))
    e.set_spectra(*api.spectra(np.linspace(420.0, 2450.0, nw)))
    lut = torch.empty((rows * g.nphi, nw), dtype=torch.float64, device="cuda")
    e.rsurf_grid_dev(g, 0, rows, lut)
    i, j, l = np.meshgrid(np.arange(g.nsza), np.arange(g.nvza), np.arange(g.nphi), indexing="ij")
    lines = np.stack([g.vza0 + j * g.dvza, g.phi0 + l * g.dphi, g.sza0 + i * g.dsza, np.zeros(i.shape)], -1).reshape(-1, 4)
    a = torch.as_tensor(lines, device="cuda")
    out = torch.empty_like(lut)
    e.rsurf_stream_dev(a, out)
    e.synchronize()
    assert relerr(lut.cpu().numpy(), out.cpu().numpy(), floor=1e-6) <= 1e-12
    e.close()
