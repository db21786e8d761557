"""BASELINE.json config 5 at its stated size under pytest: the 1000-member EnKF ensemble (seed 12345,
one sun zenith, 91 x 361 view directions, 2101 bands = 6.9e10 samples, 552 GB) in chunks of members on ONE GPU,
and the member-sharded exchange of its reduced product on two ranks.

There is no reference code for an ensemble (README.md:8-9 names the use only): every member is one forward run of
`gortt` with that member's flags, so the checks are (a) the members the reference itself was run on (gap tables of
members 0-39: tests/golden/fuzz_canopies.*, BRDF of members 0-7: c5_members.npz), and (b) size-independent
properties of the whole 552 GB: NaN exactly at view zenith 90 deg (gortt.c:890-897 reads past the table there),
energy closure, chunked == unchunked bit for bit."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, relerr
from gort_amd import api
from gort_amd.ensemble import c5_grid, draw_c5_members

pytestmark = pytest.mark.gpu
REGRESSION = 1e-9
N_MEMBERS, CHUNK = 1000, 100


def test_c5_thousand_members_in_chunks(golden):
    import torch
    canopies, leaf = draw_c5_members(N_MEMBERS)
    wl = np.arange(400.0, 2501.0)
    g = c5_grid()
    nang = g.nvza * g.nphi
    eng = api.Engine()
    eng.set_members_leaf(canopies, leaf, wl, compute_gaps=True)          # 1000 gap-probability evaluations on the device
    eng.synchronize()

    # (a1) members 0..39: gap tables against the reference (`gortt -W` of exactly these flags)
    specs = json.load(open(os.path.join(GOLDEN, "fuzz_canopies.json")))
    lut_tab = golden("fuzz_canopies.npz")["lut"]
    c5_rows = [i for i, sp in enumerate(specs) if sp["tag"] == "c5"]
    assert len(c5_rows) == 40
    for m, row in enumerate(c5_rows):
        cm, _, _, _ = eng.get_member(m)
        hb, br, pcc = specs[row]["kw"]["newstyle"]                          # the fixture stores the flags of its draw
        ref_c = api.make_canopy(newstyle=(hb, br, pcc), lai=specs[row]["kw"]["lai"])
        assert (cm.r, cm.b, cm.h1, cm.h2, cm.lambda_, cm.favd) == (ref_c.r, ref_c.b, ref_c.h1, ref_c.h2, ref_c.lambda_, ref_c.favd)
        tab = lut_tab[row]
        assert relerr(np.array(cm.p_n0)[:90], tab[:90, 0], floor=1e-12) <= REGRESSION
        assert relerr(np.array(cm.epgap)[:90], tab[:90, 1], floor=1e-12) <= REGRESSION
        assert relerr(np.array([cm.k_open, cm.k_openep]), tab[90], floor=1e-12) <= REGRESSION

    # the whole ensemble, chunk by chunk
    lut = torch.empty((CHUNK, g.nvza, g.nphi, wl.size), dtype=torch.float64, device="cuda")
    keep = {}                                                            # members kept for the checks below
    want_keep = list(range(8)) + [95, 99, 100, 104, 999]
    horizon_ok = finite_ok = True
    checksum = 0.0
    for m0 in range(0, N_MEMBERS, CHUNK):
        torch.cuda.synchronize()
        eng.rsurf_members_grid_dev(g, m0, m0 + CHUNK, lut)
        eng.synchronize()
        # (b1) NaN exactly at view zenith 90 deg, finite everywhere below: over all 6.9e9 samples of the chunk
        horizon_ok &= bool(torch.isnan(lut[:, 90]).all())
        finite_ok &= bool(torch.isfinite(lut[:, :90]).all())
        checksum += float(lut[:, :90].sum())
        for m in want_keep:
            if m0 <= m < m0 + CHUNK:
                keep[m] = lut[m - m0, ::15, ::40].cpu().numpy().copy()   # view zenith 0,15,..,90 x azimuth 0,40,..,360
    assert horizon_ok and finite_ok and np.isfinite(checksum) and checksum > 0

    # (a2) members 0..7 against the reference's BRDF goldens (sza 30; vza 0, 45, 90; phi 0, 90, 180, 300 are in the fixture;
    #      the kept sub-grid holds vza 0, 45, 90 at rows 0, 3, 6 and phi 0, 40, ..: phi = 0 is common to both)
    gm = golden("c5_members.npz")
    for i in range(8):
        for a, (vz, ph) in enumerate(gm["angles"][:, :2]):
            if ph == 0.0 and vz in (0.0, 45.0, 90.0):
                assert relerr(keep[i][int(vz // 15), 0], gm["m%d/rsurf" % i][a], floor=1e-12) <= REGRESSION, (i, vz)

    # (b2) chunked == unchunked bit for bit: members 95..104 straddle a chunk boundary
    small = torch.empty((10, g.nvza, g.nphi, wl.size), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    eng.rsurf_members_grid_dev(g, 95, 105, small)
    eng.synchronize()
    for m in (95, 99, 100, 104):
        assert np.array_equal(small[m - 95, ::15, ::40].cpu().numpy().view(np.int64), keep[m].view(np.int64)), m
    # ... and a member equals the single-canopy engine run with its inputs
    single = api.Engine()
    cm, rs, rl, tl = eng.get_member(999)
    single.set_canopy(cm); single.set_spectra(rs, rl, tl)
    one = torch.empty((g.nvza, g.nphi, wl.size), dtype=torch.float64, device="cuda")
    g1 = c5_grid()
    torch.cuda.synchronize()
    single.rsurf_grid_dev(g1, 0, g1.nvza, one)
    single.synchronize()
    assert np.array_equal(one[::15, ::40].cpu().numpy().view(np.int64), keep[999].view(np.int64))
    single.close()

    # (b3) the reduced product: albedo + vegetation absorption + soil absorption = 1 for every member and band
    sun = torch.tensor([[0.0, 0.0, 30.0, 0.0]], dtype=torch.float64, device="cuda")
    energy = torch.empty((N_MEMBERS, 1, wl.size, 3), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    eng.energy_members_dev(sun, 0, N_MEMBERS, energy)
    eng.synchronize()
    e = energy.cpu().numpy()
    assert np.isfinite(e).all() and np.abs(e.sum(axis=3) - 1.0).max() <= 1e-12
    assert (e[..., 0] > 0).all() and (e[..., 0] < 1).all()
    eng.close()


def test_c5_member_sharded_albedo_table_two_ranks(tmp_path):
    """The one exchange step of config 5 on two ranks (both on this GPU, gloo: a 1-GPU box cannot run RCCL between
    two ranks): members sharded by row_slab, each rank computes its members' albedo tables on the device, one
    all-gather; the gathered 60 x 2101 x 3 table equals the one-rank table bit for bit."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GORT_OUT=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    tool = os.path.join(root, "tools", "ensemble_multi.py")
    one = subprocess.run(["python3", tool, "--members", "60", "--no-lut"], capture_output=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    two = subprocess.run(["python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29617", tool, "--members", "60", "--no-lut", "--rehearse"],
                         capture_output=True, timeout=600, env=env)
    assert two.returncode == 0, two.stderr.decode()[-2000:]
    a = np.load(os.path.join(str(tmp_path), "ensemble_energy_w1.npy"))
    b = np.load(os.path.join(str(tmp_path), "ensemble_energy_w2.npy"))
    assert a.shape == (60, 2101, 3) and np.array_equal(a.view(np.int64), b.view(np.int64))
    assert np.abs(a.sum(axis=2) - 1.0).max() <= 1e-12
