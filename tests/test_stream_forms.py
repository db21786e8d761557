"""The kernels of the stream expansion (gort_amd/csrc/gort_stream_expand.hip) - the aligned flat-panel kernel of wide
streams, the band-major and per-sample kernels of narrow ones, the launch fused with the geometry for a few bands -
must write the SAME BITS for the same lines, agree with the LUT path on grid angles to rounding and with the oracle to
1e-9.  A wide stream cut into pieces of less than 4M samples goes through the narrow kernels: that is the comparison.

Reference interface: the per-line loop of main(), gortt.c:232-329 (+ gortt_rsurf, gortt.c:385-578)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import relerr
from gort_amd import api
from oracle import oracle as O

pytestmark = pytest.mark.gpu

REGRESSION = 1e-9


def _oracle_like(c):
    o = O.make_canopy(favd=c.favd, r=c.r, b=c.b, h1=c.h1, h2=c.h2, lam=c.lambda_, gaps=False)
    O.set_gap_tables(o, np.array(c.p_n0), np.array(c.epgap), c.k_open, c.k_openep)
    return o


@pytest.fixture(scope="module")
def setup():
    import torch
    c = api.gap_probabilities(api.make_canopy(lai=4.0))
    eng = api.Engine()
    eng.set_canopy(c)
    yield eng, c, torch
    eng.close()


def _run(eng, torch, ang, nw, pieces=False, out=None):
    """The stream in ONE call (wide kernel where it applies), or cut into pieces below the wide kernels' thresholds (4M
    samples for the flat-panel kernel, 256K samples for the line kernel's band counts)."""
    a = torch.as_tensor(np.ascontiguousarray(ang), device="cuda")
    n = ang.shape[0]
    if out is None:
        out = torch.full((n, nw), -7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()            # torch fills on ITS stream; the engine works on its own (non-blocking) one
    forms = set()
    lines_kernel = nw <= 255 or (nw <= 600 and nw % 128 != 0)          # gort_stream_lines.hip: stream_takes_lines_kernel
    step = n if not pieces else max(1, ((1 << (18 if lines_kernel else 22)) - 1) // nw)
    for i in range(0, n, step):
        eng.rsurf_stream_dev(a[i:i + step], out[i:i + step])
        forms.add(eng.stream_form())
    eng.synchronize()
    assert len(forms) == 1, forms
    return out, forms.pop()


def _bits_equal(x, y):
    return bool((x.view(dtype=__import__("torch").int64) == y.view(dtype=__import__("torch").int64)).all())


def _lines(rng, n, sza_pool):
    sza = rng.choice(sza_pool, n)
    return np.stack([rng.uniform(-89, 89, n), rng.uniform(-400, 400, n), sza, rng.uniform(-400, 400, n)], 1)


def test_flat_kernel_equals_narrow_kernels_bitwise_and_oracle(setup):
    """70 001 lines (ragged last panel) x 2101 bands, 91 integer sun zeniths in random order, some of them negative
    (zenith -> |zenith|, azimuth + 180)."""
    eng, c, torch = setup
    rng = np.random.default_rng(91)
    wl = np.arange(400.0, 2501.0)
    rs, rl, tl = api.spectra(wl)
    eng.set_spectra(rs, rl, tl)
    pool = np.concatenate([np.arange(0.0, 90.0), -np.arange(1.0, 45.0)])
    ang = _lines(rng, 70001, pool)
    g, form_g = _run(eng, torch, ang, wl.size)
    assert form_g == "flat"
    p, form_p = _run(eng, torch, ang, wl.size, pieces=True)
    assert form_p == "narrow"
    assert _bits_equal(g, p)
    idx = np.sort(rng.choice(ang.shape[0], 40, replace=False))
    idx[0], idx[-1] = 0, ang.shape[0] - 1
    ref, _, _ = O.rsurf_stream(_oracle_like(c), ang[idx], rs, rl, tl, want_K=False)
    got = g[torch.as_tensor(idx, device="cuda")].cpu().numpy()
    assert relerr(got, ref, floor=1e-12) <= REGRESSION


@pytest.mark.parametrize("nw", [17, 18, 31, 32, 33, 47, 48, 64, 65, 100, 127, 128, 129, 143, 144, 190, 255, 256, 257, 300, 384, 512, 513, 600, 601, 1000, 1999, 2048, 3000])
def test_flat_kernel_band_counts_and_output_alignments(setup, nw):
    """Band counts with every gcd(nw, 128) (wave strides of 1..128 chunk columns), the last panel ragged, the output
    itself starting off a 1-KiB chunk boundary (front and back edge handling).  Up to 255 bands: the fused line kernel
    (lines in lanes, rows through LDS rings indexed by absolute position, whole 128-B lines whatever the band count; the
    output offsets 1 and 5 leave partial lines at both ends of every wave's span), ragged last wave; and on to 600 bands off
    the 128-band grid, where it beats records + flat panels."""
    eng, c, torch = setup
    rng = np.random.default_rng(nw)
    wl = np.linspace(400.0, 2500.0, nw)
    eng.set_spectra(*api.spectra(wl))
    n = (1 << 22) // nw + 4097
    ang = _lines(rng, n, np.array([0.0, 12.5, 30.0, 47.25, 60.0, 75.0, 88.0]))
    p, form = _run(eng, torch, ang, nw, pieces=True)
    assert form == "narrow"
    for offset in (0, 1, 5, 16, 127):                    # doubles in front of the output
        buf = torch.full((n * nw + 160,), -7.0, dtype=torch.float64, device="cuda")
        out = buf[offset:offset + n * nw].view(n, nw)
        g, form = _run(eng, torch, ang, nw, out=out)
        # the line kernel up to 255 bands, and up to 600 where the band count is not a multiple of 128 (gort_stream_lines.hip)
        assert form == ("lines" if nw <= 255 or (nw <= 600 and nw % 128) else "flat")
        assert _bits_equal(g, p), (nw, offset)
        assert float(buf[:offset].min() if offset else -7.0) == -7.0 and float(buf[offset + n * nw:].max()) == -7.0


def test_one_sun_zenith_and_horizon_and_nan_lines(setup):
    """A principal-plane style stream (ONE sun zenith, 300 000 lines), with lines beyond the horizon and NaN zeniths,
    which give NaN rows whatever the kernel."""
    eng, c, torch = setup
    rng = np.random.default_rng(7)
    wl = np.linspace(400.0, 2500.0, 640)                     # the flat-panel kernel (300 bands would take the line kernel)
    rs, rl, tl = api.spectra(wl)
    eng.set_spectra(rs, rl, tl)
    n = 300000
    ang = _lines(rng, n, np.array([30.0]))
    ang[rng.choice(n, 50, replace=False), 2] = 95.0
    ang[rng.choice(n, 50, replace=False), 2] = np.nan
    ang[rng.choice(n, 50, replace=False), 0] = 90.0
    g, form = _run(eng, torch, ang, wl.size)
    assert form == "flat"
    p, _ = _run(eng, torch, ang, wl.size, pieces=True)
    assert _bits_equal(g, p)
    bad = np.isnan(ang[:, 2]) | (np.abs(ang[:, 2]) > 90) | (np.abs(ang[:, 0]) >= 90)
    nan_rows = torch.isnan(g).all(dim=1).cpu().numpy()
    assert np.array_equal(nan_rows, bad)
    idx = np.flatnonzero(~bad)[:25]
    ref, _, _ = O.rsurf_stream(_oracle_like(c), ang[idx], rs, rl, tl, want_K=False)
    assert relerr(g[torch.as_tensor(idx, device="cuda")].cpu().numpy(), ref, floor=1e-12) <= REGRESSION


def test_grid_lines_through_the_stream_equal_the_lut(setup):
    """The metric grid's angles (a slab of it) streamed as lines `vza phi sza 0`: the stream family (regrouped sample:
    one shared reciprocal x bilinear numerator + linear part) against the LUT family (five terms, dot5) - the same numbers up to rounding of
    two different associations: 1e-13 relative."""
    eng, c, torch = setup
    wl = np.arange(400.0, 2501.0)
    eng.set_spectra(*api.spectra(wl))
    g = api.hemisphere_grid()
    r0, r1 = 3 * 91 + 17, 3 * 91 + 17 + 12                   # 12 rows x 361 azimuths = 4332 lines, one sun zenith ... two
    lut = torch.empty(((r1 - r0) * g.nphi, wl.size), dtype=torch.float64, device="cuda")
    eng.rsurf_grid_dev(g, r0, r1, lut)
    eng.synchronize()
    rows = np.arange(r0, r1)
    ang = np.array([[float(r % 91), float(l), float(r // 91), 0.0] for r in rows for l in range(361)])
    s, form = _run(eng, torch, ang, wl.size)
    assert form == "flat"
    assert relerr(s.cpu().numpy(), lut.cpu().numpy(), floor=1e-12) <= 1e-13


@pytest.mark.ab
@pytest.mark.parametrize("nw", [1, 4, 16, 17])
def test_few_band_streams_fused_launch_equals_two_kernels(setup, nw):
    """Streams of up to 16 bands (C1, C2, an ensemble filter's observation operator) take ONE launch: geometry and samples
    fused, no records.  Same bits as the two-kernel path (GORT_STREAM_FUSE=0), K included, NaN lines included; 17 bands
    take the two-kernel path either way.  The member-batched entry point fuses the same way."""
    eng, c, torch = setup
    rng = np.random.default_rng(100 + nw)
    wl = np.sort(rng.uniform(400.0, 2500.0, nw))
    eng.set_spectra(*api.spectra(wl))
    n = 5000
    ang = _lines(rng, n, np.linspace(0.0, 89.0, 90))
    ang[7, 2] = 95.0                                             # sun below the horizon: NaN row
    ang[11, 0] = float("nan")
    a = torch.as_tensor(np.ascontiguousarray(ang), device="cuda")
    res = {}
    for fuse in ("1", "0"):
        os.environ["GORT_STREAM_FUSE"] = fuse
        out = torch.full((n, nw), -7.0, dtype=torch.float64, device="cuda")
        K = torch.full((n, 4), -7.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        eng.rsurf_stream_dev(a, out, K_t=K)
        eng.synchronize()
        res[fuse] = (out.cpu().numpy(), K.cpu().numpy())
    os.environ.pop("GORT_STREAM_FUSE", None)
    assert np.array_equal(res["1"][0].view(np.int64), res["0"][0].view(np.int64))
    assert np.array_equal(res["1"][1].view(np.int64), res["0"][1].view(np.int64))
    assert np.isnan(res["1"][0][7]).all() and np.isnan(res["1"][0][11]).all() and not np.isnan(res["1"][0][:7]).any()
    ref, _, _ = O.rsurf_stream(O.make_canopy(lai=4.0), ang[:64], *O.spectra(wl), want_K=False)
    assert relerr(res["1"][0][:64], ref, floor=1e-12) <= 1e-9


@pytest.mark.ab
def test_member_batched_few_band_stream_fuses_the_same_way():
    from gort_amd.ensemble import DEFAULT, Ensemble
    rng = np.random.default_rng(77)
    wl = np.array([450.0, 555.0, 645.0, 858.5])
    angles = np.stack([rng.uniform(-70, 70, 40), rng.uniform(0, 360, 40), rng.uniform(0, 75, 40), rng.uniform(0, 360, 40)], 1)
    states = [dict(DEFAULT, LAI=float(x), Cab=float(y)) for x, y in zip(rng.uniform(0.5, 6, 9), rng.uniform(10, 60, 9))]
    ens = Ensemble(wl).set_states(states)
    os.environ["GORT_STREAM_FUSE"] = "1"
    fused = ens.observe(angles)
    os.environ["GORT_STREAM_FUSE"] = "0"
    two = ens.observe(angles)
    os.environ.pop("GORT_STREAM_FUSE", None)
    ens.close()
    assert fused.shape == (9, 40, 4) and np.array_equal(fused.view(np.int64), two.view(np.int64))


@pytest.mark.ab
def test_flat_kernel_panel_shapes_write_the_same_bits():
    """The panel shape of the flat stream kernel (waves x steps: chosen from the stream's size, or GORT_STREAM_WAVES /
    GORT_STREAM_STEPS) changes the ORDER in which chunks are written, never a bit: one sha256 over 20 outputs (five band
    counts x four output alignments, NaN lines included; tools/probes/stream_digest.py) under the automatic shape,
    round 2's 64 steps x ~16 808 waves, very short and ragged panels, the three XCD mappings, plain stores.  (The tuning is read once per process.)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for shape in ({}, {"GORT_STREAM_WAVES": "16808", "GORT_STREAM_STEPS": "64"}, {"GORT_STREAM_STEPS": "3", "GORT_EXPAND_XCD": "2"},
                  {"GORT_STREAM_WAVES": "5000", "GORT_STREAM_STEPS": "7", "GORT_EXPAND_XCD": "0", "GORT_EXPAND_NT": "0"}):
        env = dict(os.environ)
        env.update(shape)
        run = subprocess.run([sys.executable, os.path.join(root, "tools", "probes", "stream_digest.py")], capture_output=True,
                             timeout=600, env=env, cwd=root)
        assert run.returncode == 0, run.stderr.decode()[-2000:]
        line = [x for x in run.stdout.decode().splitlines() if x.startswith("digest ")]
        assert line, run.stdout.decode()[-500:]
        digests[str(shape)] = line[-1]
    assert len(set(digests.values())) == 1, digests


CANOPIES_AT_THE_HORIZON = {
    "lai4": dict(lai=4.0),
    "lai0": dict(lai=0.0),                                            # favd = 0: the reference divides by it (inf / NaN of its own)
    "thin": dict(lai=0.05),
    "newstyle": dict(newstyle=(2.0, 2.5, 0.6), lai=3.3),
    "q08": dict(lai=2.0, q08=True),
    "beta_diffuse": dict(lai=1.0, beta=0.3, diffuse=0.4),
}


@pytest.mark.parametrize("canopy", sorted(CANOPIES_AT_THE_HORIZON))
@pytest.mark.parametrize("nw", [1, 40, 300])
def test_lines_typed_at_90_degrees_skip_the_reference_route_only_for_reflectances(nw, canopy):
    """A line whose zenith was typed as exactly +-90 degrees has the same reflectances by either arithmetic - NaN, or for a
    canopy without leaves whatever both make of it - and its wave does not walk the reference's route when nothing but
    reflectances is asked for (gort_geometry.h, stream_line_takes_reference_route): the principal plane of BASELINE config
    2 with and without the viewed proportions - the same bits wherever a number stands, NaN where NaN stands, and the
    proportions of the 90-degree lines as the reference's route makes them (gortt.c:424-449; the `horizon` CLI golden holds
    their digits).  Fused (1 band), line kernel (40) and records + flat kernel (300); six kinds of canopy."""
    import torch
    wl = np.linspace(450.0, 2400.0, nw)
    e = api.Engine()
    e.set_canopy(api.gap_probabilities(api.make_canopy(**CANOPIES_AT_THE_HORIZON[canopy])))
    e.set_spectra(*api.spectra(wl))
    n = 8192 if nw == 40 else (20000 if nw == 300 else 181)          # enough samples for the form the band count names
    vza = np.resize(np.arange(-90.0, 91.0), n)
    ang = np.stack([vza, np.zeros(n), np.full(n, 30.0), np.zeros(n)], 1)
    ang[5, 2] = 90.0                                                  # a sun on the horizon too
    ang[7, 2] = -90.0
    a = torch.as_tensor(ang, device="cuda")
    plain = torch.empty((n, nw), dtype=torch.float64, device="cuda")
    with_k = torch.empty((n, nw), dtype=torch.float64, device="cuda")
    K = torch.empty((n, 4), dtype=torch.float64, device="cuda")
    e.rsurf_stream_dev(a, plain)
    e.rsurf_stream_dev(a, with_k, None, K)
    e.synchronize()
    p, q, k = plain.cpu().numpy(), with_k.cpu().numpy(), K.cpu().numpy()
    at90 = (np.abs(ang[:, 0]) == 90.0) | (np.abs(ang[:, 2]) == 90.0)
    assert at90.sum() >= 4
    assert np.array_equal(np.isnan(p), np.isnan(q)), (canopy, nw, ang[np.flatnonzero((np.isnan(p) != np.isnan(q)).any(axis=1))[:4]])
    both = ~np.isnan(p)
    assert np.array_equal(p[both].view(np.int64), q[both].view(np.int64)), (canopy, nw)
    if canopy == "lai4":
        assert np.isnan(p[at90]).all() and np.isfinite(p[~at90]).all()
        assert np.isfinite(k[np.abs(ang[:, 0]) == 90.0]).all()
    e.close()
