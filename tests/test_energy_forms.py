"""The forms of the `-energy` path (gort_amd/csrc/gort_energy.hip) must write the SAME BITS:

* the row terms of the 16 zenith nodes evaluated once per line into LDS (round 4) against every quadrature node
  evaluating its whole geometry (round 3's kernel, GORT_ENERGY_SHARE_ROWS=0);
* rows shared between lines of equal sun direction and broadcast by energy_broadcast_kernel - both of its forms, the
  two-rows-per-chunk form every stream of >= 43 bands takes included, member-batched launches included - against the
  evaluation of every line (GORT_ENERGY_DEDUP=0).

Reference: gortt_energy / gortt_albedo, gortt_albedo.c:7-138 (512 gortt_rsurf calls per line and band)."""
import os

import numpy as np
import pytest

from gort_amd import api

pytestmark = pytest.mark.gpu


def _engine(wl, dedup="1", lai=4.0):
    os.environ["GORT_ENERGY_DEDUP"] = dedup               # read when the engine is created
    try:
        e = api.Engine()
    finally:
        os.environ.pop("GORT_ENERGY_DEDUP")
    e.set_canopy(api.gap_probabilities(api.make_canopy(lai=lai)))
    e.set_spectra(*api.spectra(wl))
    return e


def _energy(e, ang, nw, torch, offset=0):
    """energy[nA][nw][3] into a buffer that starts `offset` doubles behind a 1-KiB boundary, sentinels around it."""
    n = ang.shape[0]
    buf = torch.full((n * nw * 3 + 256,), -7.0, dtype=torch.float64, device="cuda")
    base = (-(buf.data_ptr() // 8)) % 128 + offset          # doubles to the next 1-KiB boundary, then `offset` more
    out = buf[base:base + n * nw * 3].view(n, nw, 3)
    a = torch.as_tensor(np.ascontiguousarray(ang), device="cuda")
    torch.cuda.synchronize()
    e.energy_stream_dev(a, out)
    e.synchronize()
    assert float(buf[:base].min() if base else -7.0) == -7.0 and float(buf[base + n * nw * 3:].max()) == -7.0
    return out.cpu().numpy()


def _bits(x):
    return np.ascontiguousarray(x).view(np.int64)


def _same_bits_nan_for_nan(a, b):
    """Equal bit for bit where a number stands, NaN where NaN stands (sign and payload of a NaN follow the operand order of
    the instruction that made it, which differs between the forms)."""
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(_bits(a)[~na], _bits(b)[~nb])


@pytest.mark.ab
def test_row_terms_once_per_zenith_node_same_bits():
    """BASELINE config 4's 91 sun zeniths (the per-line kernel) and a 3000-line stream in which every line has its own sun
    direction (the list kernel behind the sun-direction table), a sun on the horizon and a NaN line among them."""
    import torch
    wl = np.linspace(400.0, 2500.0, 61)
    rng = np.random.default_rng(16)
    c4 = np.array([[0.0, 0.0, float(s), 0.0] for s in range(91)])
    stream = np.stack([rng.uniform(-89, 89, 3000), rng.uniform(-400, 400, 3000), rng.uniform(0, 89.9, 3000), rng.uniform(-400, 400, 3000)], 1)
    stream[5, 2] = 90.0
    stream[9, 2] = np.nan
    stream[11, 2] = -33.25
    res = {}
    for share in ("1", "0"):
        os.environ["GORT_ENERGY_SHARE_ROWS"] = share
        try:
            e = _engine(wl)
            res[share] = (_energy(e, c4, wl.size, torch), _energy(e, stream, wl.size, torch, offset=3))
            e.close()
        finally:
            os.environ.pop("GORT_ENERGY_SHARE_ROWS")
    for k in (0, 1):
        a, b = res["1"][k], res["0"][k]
        ang = c4 if k == 0 else stream
        # a sun exactly on the horizon: albedo and favegt are NaN by either form, fasoil (sun terms only) is finite, and the
        # round-3 form takes the reference's own route there (gort_device.h: near_horizon) where the shared-row form, which
        # hands out reflectances only, does not: equal to rounding, not in bits
        horizon = np.abs(ang[:, 2]) == 90.0
        assert horizon.sum() == 1
        assert np.isnan(a[horizon, :, :2]).all() and np.isnan(b[horizon, :, :2]).all()
        np.testing.assert_allclose(a[horizon, :, 2], b[horizon, :, 2], rtol=1e-13, atol=0)
        differ = (_bits(a) != _bits(b)).any(axis=(1, 2)) & ~(np.isnan(a).all(axis=(1, 2)) & np.isnan(b).all(axis=(1, 2))) & ~horizon
        bad = np.flatnonzero(differ)
        assert bad.size == 0, (k, bad[:10], ang[bad[:3]], a[bad[:1], :2], b[bad[:1], :2])
    assert np.isnan(res["1"][1][9]).all() and np.isfinite(res["1"][1][:5]).all()
    assert np.abs(res["1"][0][:90].sum(axis=2) - 1.0).max() < 1e-12        # albedo + favegt + fasoil = 1


@pytest.mark.ab
@pytest.mark.parametrize("nw,n", [(61, 3001), (2101, 702), (5, 129), (513, 1503)])
def test_batched_list_kernel_same_bits(nw, n):
    """energy_list_batched_kernel (four lines per workgroup pass: their row terms side by side on four waves, a band's
    constants loaded once for the four) against the line-after-line loop (GORT_ENERGY_BATCH=0): every line its own sun, a
    sun on the horizon, a NaN line, list lengths that take four, two and one line per pass and leave ragged last batches, band
    counts around one pass of 512."""
    import torch
    wl = np.linspace(400.0, 2500.0, nw)
    rng = np.random.default_rng(nw + n)
    stream = np.stack([rng.uniform(-89, 89, n), rng.uniform(-400, 400, n), rng.uniform(0, 89.9, n), rng.uniform(-400, 400, n)], 1)
    stream[3, 2] = 90.0
    stream[7, 2] = np.nan
    stream[n - 1, 2] = 0.0
    res = {}
    for pipe in ("1", "0"):
        os.environ["GORT_ENERGY_BATCH"] = pipe
        try:
            e = _engine(wl)
            res[pipe] = _energy(e, stream, nw, torch, offset=1)
            e.close()
        finally:
            os.environ.pop("GORT_ENERGY_BATCH")
    assert not (res["1"] == -7.0).any()
    assert np.isnan(res["1"][7]).all() and np.isfinite(res["1"][:3]).all()
    assert _same_bits_nan_for_nan(res["1"], res["0"])


@pytest.mark.ab
@pytest.mark.parametrize("nw,n", [(43, 4001), (128, 1501), (2101, 333), (7, 5001), (128, 100003), (47, 300007)])
def test_broadcast_of_shared_rows_same_bits(nw, n):
    """Rows of 3 nw doubles copied from the line that owns their sun direction, by either broadcast (chunk by chunk of the
    output, row by row of the lines; which one works is chosen on the device by the share of owner lines): 43 bands are the
    first to take the chunk form with two rows per 1-KiB chunk (the two long streams: enough chunks for its straight-line
    path of sixteen steps per wave, which short outputs never reach) (incremental row / offset carry, chunks that straddle rows, front and back edges), 128
    and 2101 bands its long rows, 7 bands the per-element form.  Odd nA x row, an output that starts off the chunk grid,
    sentinels on both sides."""
    import torch
    wl = np.linspace(400.0, 2500.0, nw)
    rng = np.random.default_rng(nw)
    sza = rng.integers(0, 12, n).astype(float) * 7.0
    saa = rng.choice(np.array([0.0, 77.5, 180.0]), n)
    flip = rng.random(n) < 0.2                                              # -sza, saa + 180: the same sun after normalisation
    ang = np.stack([rng.uniform(-89, 89, n), rng.uniform(-400, 400, n), np.where(flip, -sza, sza), np.where(flip, saa + 180.0, saa)], 1)
    ang[-1, 2] = 41.0                                                       # the last line owns its direction: nothing copied into the back edge
    assert nw % 2 == 0 or (n * nw * 3) % 2 == 1
    res = {}
    for form in ("every line", "chunks", "rows", "chosen on the device"):
        if form in ("chunks", "rows"):
            os.environ["GORT_ENERGY_BROADCAST"] = form       # read per call (gort_energy.hip, launch_energy)
        try:
            e = _engine(wl, "0" if form == "every line" else "1")
            res[form] = [_energy(e, ang, nw, torch, offset=off) for off in (0, 5)]
            e.close()
        finally:
            os.environ.pop("GORT_ENERGY_BROADCAST", None)
    for form in ("chunks", "rows", "chosen on the device"):
        for k in (0, 1):
            assert not (res[form][k] == -7.0).any(), (form, k)
            assert np.array_equal(_bits(res[form][k]), _bits(res["every line"][k])), (nw, form, k)


@pytest.mark.ab
def test_member_batched_broadcast_same_bits():
    """gort_energy_members_dev: the lines' rows are shared per member (blockIdx.y); with an odd nA x row every second
    member's slab starts on an 8-byte boundary of its own."""
    import torch
    from gort_amd.ensemble import DEFAULT, Ensemble
    rng = np.random.default_rng(5)
    wl = np.linspace(400.0, 2500.0, 45)
    n = 257
    ang = np.stack([rng.uniform(-80, 80, n), rng.uniform(0, 360, n), rng.integers(0, 9, n).astype(float) * 10.0, rng.choice([0.0, 120.0], n)], 1)
    states = [dict(DEFAULT, LAI=float(x), Cab=float(y)) for x, y in zip(rng.uniform(0.5, 6, 3), rng.uniform(10, 60, 3))]
    res = {}
    for dedup in ("1", "0"):
        os.environ["GORT_ENERGY_DEDUP"] = dedup
        try:
            ens = Ensemble(wl).set_states(states)
        finally:
            os.environ.pop("GORT_ENERGY_DEDUP")
        buf = torch.full((3 * n * wl.size * 3 + 64,), -7.0, dtype=torch.float64, device="cuda")
        out = buf[1:1 + 3 * n * wl.size * 3].view(3, n, wl.size, 3)
        a = torch.as_tensor(ang, device="cuda")
        torch.cuda.synchronize()
        ens.eng.energy_members_dev(a, 0, 3, out)
        ens.eng.synchronize()
        assert float(buf[0]) == -7.0 and float(buf[1 + 3 * n * wl.size * 3:].max()) == -7.0
        res[dedup] = out.cpu().numpy()
        ens.close()
    assert (n * wl.size * 3) % 2 == 1
    assert not (res["1"] == -7.0).any()
    assert np.array_equal(_bits(res["1"]), _bits(res["0"]))


@pytest.mark.parametrize("nw,n", [(61, 3001), (2101, 702), (5, 129), (43, 4001), (128, 100003), (7, 1), (513, 90)])
def test_indexed_rows_equal_the_dense_output_bitwise(nw, n):
    """gort_energy_stream_indexed_dev: rows[index[a]] is the row gort_energy_stream_dev writes for line a, bit for bit; the
    rows are numbered in the order in which their normalised sun directions first appear (numpy's own first-occurrence
    order of the same keys); a rows_cap below the count leaves the rows beyond it untouched and still reports the count.
    The shapes of the tests above: every line its own sun (3001, 702; one of them NaN), few suns with -sza / saa + 180
    twins, streams shorter than the dense path's table threshold (1 and 90 lines)."""
    import torch
    wl = np.linspace(400.0, 2500.0, nw)
    rng = np.random.default_rng(1000 + nw + n)
    if n in (3001, 702, 1):
        ang = np.stack([rng.uniform(-89, 89, n), rng.uniform(-400, 400, n), rng.uniform(0, 89.9, n), rng.uniform(-400, 400, n)], 1)
        if n > 10:
            ang[3, 2] = 90.0
            ang[7, 2] = np.nan
    else:
        sza = (1.0 + rng.integers(0, 12, n)) * 7.0
        saa = rng.choice(np.array([0.0, 77.5, 180.0]), n)
        flip = rng.random(n) < 0.2
        ang = np.stack([rng.uniform(-89, 89, n), rng.uniform(-400, 400, n), np.where(flip, -sza, sza), np.where(flip, saa + 180.0, saa)], 1)
    e = _engine(wl)
    dense = _energy(e, ang, nw, torch, offset=1)
    a = torch.as_tensor(np.ascontiguousarray(ang), device="cuda")
    cap = n + 2
    rows = torch.full((cap, nw, 3), -7.0, dtype=torch.float64, device="cuda")
    index = torch.full((n + 1,), 0x7fffffff, dtype=torch.int32, device="cuda")
    count = torch.full((2,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    e.energy_stream_indexed_dev(a, rows, index, count)
    e.synchronize()
    n_rows = int(count[0])
    idx = index[:n].cpu().numpy()
    assert int(index[n]) == 0x7fffffff and int(count[1]) == -1
    r = rows.cpu().numpy()
    assert (r[n_rows:] == -7.0).all() and not (r[:n_rows] == -7.0).any()
    assert idx.min() == 0 and idx.max() == n_rows - 1
    assert np.array_equal(_bits(r[idx]), _bits(dense))
    # first-appearance order: the index of a line that opens a new row is one more than everything in front of it
    first = np.full(n_rows, -1)
    for line in range(n - 1, -1, -1):
        first[idx[line]] = line
    assert (np.diff(first) > 0).all() and np.array_equal(idx[first], np.arange(n_rows))
    # no more rows than the input has distinct (sun zenith, sun azimuth) pairs as typed; every line its own sun: a row per line
    typed = len({(x, y) for x, y in zip(ang[:, 2].tolist(), ang[:, 3].tolist())})
    assert n_rows <= typed and (n_rows == n if n in (3001, 702, 1) else n_rows <= 72)
    # too little room: the count is still the full one, the rows beyond the room are not touched
    if n_rows > 2:
        rows2 = torch.full((n_rows - 1, nw, 3), -7.0, dtype=torch.float64, device="cuda")
        guard = torch.full((4 * nw,), -7.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()                 # torch fills on its stream, the engine writes on its own
        e.energy_stream_indexed_dev(a, rows2[:n_rows - 2], index, count)
        e.synchronize()
        assert int(count[0]) == n_rows and float(rows2[n_rows - 2].max()) == -7.0 and float(guard.max()) == -7.0
        assert np.array_equal(_bits(rows2[:n_rows - 2].cpu().numpy()), _bits(r[:n_rows - 2]))
    # the host form: the same rows and index; too little room is GORT_ERANGE with the count
    hrows, hidx = e.energy_stream_indexed(ang)
    assert hrows.shape[0] == n_rows and np.array_equal(hidx, idx) and np.array_equal(_bits(hrows), _bits(r[:n_rows]))
    if n_rows > 1:
        with pytest.raises(api.GortError, match="%d distinct sun directions" % n_rows):
            e.energy_stream_indexed(ang, rows_cap=n_rows - 1)
    e.close()
