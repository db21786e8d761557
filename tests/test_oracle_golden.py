"""Pin the CPU oracle (oracle/gort_oracle.c) to the REAL reference.

tests/golden/* were produced by tools/make_golden.py from /root/reference compiled in
place (oracle/Makefile `ref`): the reference ships no tests or expected outputs of its
own, so these dumps are the golden vectors.  The oracle keeps
the reference's evaluation order and is expected to agree BIT-EXACTLY (same libm);
the asserted bound is 1e-13 relative to stay robust to a different glibc.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, relerr
from oracle import oracle as O

TOL = 1e-13


def canopy_from_flags(flags):
    """Minimal flag -> oracle canopy mapping for the golden canopy sets."""
    kw, ns, i = {}, {}, 0
    q08 = False
    while i < len(flags):
        f = flags[i]
        if f == "-q08_pn_kopen":
            q08 = True; i += 1; continue
        v = float(flags[i + 1]); i += 2
        if f == "-LAI": kw["lai"] = v
        elif f in ("-HB", "-BR", "-PCC"): ns[f] = v
        elif f == "-favd": kw["favd"] = v
        elif f == "-h1": kw["h1"] = v
        elif f == "-h2": kw["h2"] = v
        elif f == "-lambda": kw["lam"] = v
        elif f == "-r": kw["r"] = v
        elif f == "-b": kw["b"] = v
        else: raise KeyError(f)
    if ns:
        kw["newstyle"] = (ns.get("-HB", 2.0), ns.get("-BR", 1.0), ns.get("-PCC", 0.5))
    return O.make_canopy(q08=q08, **kw)


FLAGS = json.load(open(os.path.join(GOLDEN, "canopies_flags.json")))


@pytest.mark.parametrize("tag", sorted(FLAGS))
def test_gap_tables(tag, golden):
    g = golden("canopies.npz")
    c = canopy_from_flags(FLAGS[tag])
    sc = g[tag + "/scalars"]
    names = ["r", "b", "h1", "h2", "lambda_", "favd", "ell", "h", "elai", "tau", "z1", "z2", "lv", "favd_p",
             "tau_p", "lv_p", "z1_p", "z2_p", "h1_p", "h2_p", "dz", "ds", "dz_p", "dth"]
    for i, n in enumerate(names):
        assert getattr(c, n) == sc[i], (n, getattr(c, n), sc[i])
    assert c.nth == int(sc[24]) and c.nlayers == int(sc[25])
    assert relerr(np.array(c.theta), g[tag + "/theta"]) == 0
    assert relerr(np.array(c.theta_p), g[tag + "/theta_p"]) == 0
    assert relerr(np.array(c.height_p), g[tag + "/height_p"]) == 0
    if "q08" not in tag:
        assert relerr(np.array(c.v_g).reshape(15, 91), g[tag + "/v_g"]) <= TOL
        assert relerr(np.array(c.p_n0).reshape(15, 91), g[tag + "/p_n0"]) <= TOL
        assert relerr(np.array(c.p_s0).reshape(15, 91), g[tag + "/p_s0"], floor=1e-30) <= 1e-9
    else:
        assert relerr(np.array(c.p_n0)[:91], g[tag + "/p_n0"][0]) <= TOL
    assert relerr(np.array(c.epgap0), g[tag + "/epgap0"]) <= TOL
    assert relerr(np.array([c.k_open0, c.k_openep0]), g[tag + "/kk"]) <= TOL


def test_lut_rows_match_cli(golden):
    """-W text of the real CLI (40 decimals) parses to the oracle's tables."""
    cases = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "cli_cases.json")))}
    for name, kw in (("lut_default", dict(lai=4.0)), ("lut_q08", dict(lai=4.0, q08=True)),
                     ("lut_newstyle", dict(newstyle=(2.0, 2.0, 0.6), lai=3.3))):
        rows = [ln.split() for ln in cases[name]["stdout"].strip().split("\n")]
        assert len(rows) == 91 and rows[90][0] == "-1"
        c = O.make_canopy(**kw)
        pn0, ep, ko, kep = O.gap_tables(c)
        for j in range(90):
            assert int(rows[j][0]) == j
            # %0.40f keeps 40 decimals: exact for values >~1e-24, flushed below 1e-40
            assert float(rows[j][1]) == pytest.approx(pn0[j], rel=1e-12, abs=1e-40)
            assert float(rows[j][2]) == pytest.approx(ep[j], rel=1e-12, abs=1e-40)
        assert float(rows[90][1]) == pytest.approx(ko, rel=1e-15)
        assert float(rows[90][2]) == pytest.approx(kep, rel=1e-15)


@pytest.mark.parametrize("tag", ["default", "dry", "dense", "zero_abs", "opaque"])
def test_prospect_d_all_bands(tag, golden):
    s = golden("spectra.npz")
    RT = O.prospect_d(*s["prospect/%s/params" % tag])
    assert relerr(RT, s["prospect/%s/RT" % tag]) <= TOL


def test_spectra_interpolation(golden):
    s = golden("spectra.npz")
    rs, rl, tl = O.spectra(s["interp/default/wl"])
    assert relerr(rs, s["interp/default/rsoil"]) <= TOL
    assert relerr(rl, s["interp/default/rleaf"]) <= TOL
    assert relerr(tl, s["interp/default/tleaf"]) <= TOL
    pr = s["interp/alt/prospect"]
    rs, rl, tl = O.spectra(s["interp/alt/wl"], rsl=tuple(s["interp/alt/rsl"]),
                           prospect=dict(N=pr[0], Cab=pr[1], Car=pr[2], Anth=pr[3], Cbrown=pr[4], Cw=pr[5], Cm=pr[6]))
    assert relerr(rs, s["interp/alt/rsoil"]) <= TOL
    assert relerr(rl, s["interp/alt/rleaf"]) <= TOL
    assert relerr(tl, s["interp/alt/tleaf"]) <= TOL
    with pytest.raises(ValueError):
        O.spectra([399.0])
    with pytest.raises(ValueError):
        O.spectra([2500.5])


def test_gauleg(golden):
    s = golden("spectra.npz")
    x, w = O.gauleg(32)
    assert relerr(x, s["gauleg32/x"]) <= TOL and relerr(w, s["gauleg32/w"]) <= TOL
    # the reference's Newton loop stops at 3e-11 and evaluates the weight one step behind
    assert abs(w.sum() - 2.0) < 1e-10


def test_c2_principal_plane(golden):
    g = golden("c2_principal_plane.npz")
    c = O.make_canopy(lai=4.0)
    rs, rl, tl = O.spectra(g["wl"])
    r, sc, K = O.rsurf_stream(c, g["angles"], rs, rl, tl, want_scomp=True)
    assert relerr(r, g["rsurf"]) <= TOL
    assert relerr(K, g["K"]) <= TOL
    assert relerr(sc.reshape(g["scomp"].shape), g["scomp"]) <= TOL
    # reference returns NaN exactly at +-90 deg on the direct path
    assert np.isnan(r[[0, 180], 0]).all() and np.isfinite(r[1:180]).all()


def test_c3_subgrid(golden):
    g = golden("c3_subgrid.npz")
    c = O.make_canopy(lai=4.0)
    rs, rl, tl = O.spectra(g["wl"])
    r, _, K = O.rsurf_stream(c, g["angles"], rs, rl, tl)
    assert relerr(r, g["rsurf"]) <= TOL
    assert relerr(K, g["K"]) <= TOL


def test_random_stream_second_canopy(golden):
    g = golden("random_stream_newstyle.npz")
    c = O.make_canopy(newstyle=(2.0, 2.0, 0.6), lai=3.3)
    rs, rl, tl = O.spectra(g["wl"])
    r, sc, K = O.rsurf_stream(c, g["angles"], rs, rl, tl, want_scomp=True)
    assert relerr(r, g["rsurf"]) <= TOL
    assert relerr(K, g["K"]) <= TOL
    assert relerr(sc.reshape(g["scomp"].shape), g["scomp"]) <= TOL


def test_c4_albedo_all_sun_zeniths(golden):
    g = golden("c4_albedo.npz")
    c = O.make_canopy(lai=4.0)
    rs, rl, tl = O.spectra(g["wl_b"])
    z = np.zeros_like(g["sza_b"])
    e = O.energy_stream(c, np.stack([z, z, g["sza_b"], z], 1), rs, rl, tl)
    assert relerr(e, g["energy_b"]) <= TOL
    # sun at the horizon: albedo and favegt are NaN (kuusk's log(0)); fasoil does not depend on them
    assert np.isnan(e[90, :, :2]).all() and np.isfinite(e[90, :, 2]).all() and np.isfinite(e[:90]).all()


@pytest.mark.slow
def test_c4_albedo_full_spectrum(golden):
    g = golden("c4_albedo.npz")
    c = O.make_canopy(lai=4.0)
    rs, rl, tl = O.spectra(g["wl_a"])
    z = np.zeros_like(g["sza_a"])
    e = O.energy_stream(c, np.stack([z, z, g["sza_a"], z], 1), rs, rl, tl)
    assert relerr(e, g["energy_a"]) <= TOL


def c5_member_inputs(params):
    hb, br, pcc, lai, cab, cw, cm, N, rsl1 = params
    c = O.make_canopy(newstyle=(hb, br, pcc), lai=lai)
    pro = dict(N=N, Cab=cab, Cw=cw, Cm=cm)
    rsl = (rsl1, 0.1, 0.03726, -0.002426)
    return c, pro, rsl


@pytest.mark.parametrize("i", range(8))
def test_c5_members(i, golden):
    g = golden("c5_members.npz")
    c, pro, rsl = c5_member_inputs(g["m%d/params" % i])
    pn0, ep, ko, kep = O.gap_tables(c)
    assert relerr(pn0[:90], g["m%d/p_n0" % i]) <= TOL
    assert relerr(ep[:90], g["m%d/epgap0" % i]) <= TOL
    assert relerr(np.array([ko, kep]), g["m%d/kk" % i]) <= TOL
    rs, rl, tl = O.spectra(g["wl"], rsl=rsl, prospect=pro)
    r, _, _ = O.rsurf_stream(c, g["angles"], rs, rl, tl)
    assert relerr(r, g["m%d/rsurf" % i]) <= TOL


def _parse_fp(text, nw, prnspec, prnprop, energy):
    rows = []
    for ln in text.strip("\n").split("\n")[1:]:
        tok = ln.replace("{", " ").replace("}", " ").replace("[", " ").replace("]", " ").split()
        rows.append([float(t) for t in tok])
    return rows


def test_cli_full_precision_rows(golden):
    """Rows printed by the reference main() (full precision) == oracle stream drivers,
    including the angle normalisation quirks (negative zeniths, azimuth wraps, (int) truncation)."""
    cases = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "cli_cases.json")))}

    def check(name, c, spec_kw, prnspec=False, prnprop=False, energy=False):
        case = cases[name]
        lines = case["stdin"].strip("\n").split("\n")
        wl = [float(t) for t in lines[0].split()[2:]]
        ang = np.array([[float(t) for t in ln.split()[:4]] for ln in lines[1:]])
        rs, rl, tl = O.spectra(wl, **spec_kw)
        r, sc, K = O.rsurf_stream(c, ang, rs, rl, tl, want_scomp=True)
        e = O.energy_stream(c, ang, rs, rl, tl) if energy else None
        rows = _parse_fp(case["stdout_fp"], len(wl), prnspec, prnprop, energy)
        assert len(rows) == len(ang)
        for a, v in enumerate(rows):
            exp = list(ang[a])
            for k in range(len(wl)):
                exp.append(r[a, k])
                if prnspec: exp.extend(sc[a, 4 * k:4 * k + 4])
            if prnprop: exp.extend(K[a])
            if energy: exp.extend(e[a].reshape(-1))
            assert relerr(np.array(v), np.array(exp)) <= TOL, (name, a)

    lai4 = O.make_canopy(lai=4.0)
    check("readme", lai4, {})
    check("readme_all", lai4, {}, prnspec=True, prnprop=True, energy=True)
    check("readme_q08", O.make_canopy(lai=4.0, q08=True), {})
    check("newstyle", O.make_canopy(newstyle=(2.0, 2.0, 0.6), lai=3.3), {}, prnprop=True)
    check("overrides", O.make_canopy(lai=2.0, beta=0.5, diffuse=0.3), dict(alb_leaf=0.9, alb_soil=0.2))
    check("oldstyle", O.make_canopy(favd=0.6, h1=2.5, h2=9, lam=0.3, r=1.1, b=2.0), {}, prnprop=True)
    check("prospect_flags", O.make_canopy(lai=3.0),
          dict(rsl=(0.3, 0.05, 0.01, 0.001), prospect=dict(N=1.8, Cab=45, Car=8, Anth=2, Cbrown=0.1, Cw=0.01, Cm=0.005)),
          prnspec=True)
    check("horizon", lai4, {}, prnprop=True)
    check("azimuth_wrap", lai4, {}, prnprop=True)


# ------------------------------------------------------------------ fuzz canopies (230, from the real reference)
FUZZ = json.load(open(os.path.join(GOLDEN, "fuzz_canopies.json")))


def fuzz_kw(spec):
    kw = dict(spec["kw"])
    if "newstyle" in kw:
        kw["newstyle"] = tuple(kw["newstyle"])
    return kw


def test_fuzz_canopies_gap_tables(golden):
    """230 canopies dumped from the REAL reference (`gortt -W` at %.17g): exact-tie geometries (-BR 1/2/3), oblate
    crowns, LAI 0.1 and 9, 150 draws of the GPU fuzz test, the first 40 C5 members, old-style flags.  Two of
    them (HB = BR = 1) give NaN in the reference; relerr() demands the same NaN pattern."""
    lut = golden("fuzz_canopies.npz")["lut"]
    assert lut.shape == (len(FUZZ), 91, 2) and len(FUZZ) >= 200
    worst = 0.0
    for spec, tab in zip(FUZZ, lut):
        c = O.make_canopy(**fuzz_kw(spec))
        pn0, ep, ko, kep = O.gap_tables(c)
        e = max(relerr(pn0[:90], tab[:90, 0]), relerr(ep[:90], tab[:90, 1]), relerr(np.array([ko, kep]), tab[90]))
        assert e <= TOL, (spec, e)
        worst = max(worst, e)
    print("oracle vs reference, %d fuzz canopies: worst %.1e" % (len(FUZZ), worst))


def test_fuzz_canopies_brdf_rows(golden):
    """BRDF rows (rsurf, C/G/T/Z, K) of 24 of those canopies from the reference: exact hot-spot lines, table nodes,
    view and sun near the horizon, the hot spot AT 89 deg."""
    g = golden("fuzz_canopies.npz")
    wl, lines = g["brdf_wl"], g["brdf_lines"]
    rs, rl, tl = O.spectra(wl)
    for k, i in enumerate(g["brdf_pick"]):
        c = O.make_canopy(**fuzz_kw(FUZZ[int(i)]))
        r, sc, K = O.rsurf_stream(c, lines, rs, rl, tl, want_scomp=True)
        assert relerr(r, g["brdf_rsurf"][k]) <= TOL, FUZZ[int(i)]
        assert relerr(sc.reshape(g["brdf_scomp"][k].shape), g["brdf_scomp"][k]) <= TOL
        assert relerr(K, g["brdf_K"][k], floor=1.0) <= TOL


def test_reference_branch_coverage_table_is_complete():
    """tests/golden/ref_branch_coverage.md (tools/ref_coverage.py, gcov build of the reference): every branch outcome
    of gortt_get_s's thresholds and of gortt_vol and its helpers is taken by at least one reference-generated canopy,
    so the oracle is pinned on every piece of the piecewise geometry it restates."""
    import re
    text = open(os.path.join(GOLDEN, "ref_branch_coverage.md")).read()
    m = re.search(r"\*\*(\d+) of (\d+) branch outcomes", text)
    assert m and m.group(1) == m.group(2) and int(m.group(2)) >= 30
    assert "never reached: none" in text
    for line in (582, 589, 679, 683, 690, 697, 725, 736, 752):
        row = re.search(r"^\| %d \| (\d+) \| (\d+) / (\d+) \|" % line, text, re.M)
        assert row and int(row.group(2)) > 0 and int(row.group(3)) > 0, line


def test_reference_build_manifest_matches_what_is_on_disk():
    """oracle/_ref.MANIFEST (tracked) names the reference build the goldens, the parity_reference check and the CPU baseline
    of bench.py come from; where the build itself is present (build container, gpurun snapshot) its hashes must be the
    manifest's - a stale manifest or a foreign build would make BENCH's `reference_build` meaningless."""
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    man = json.load(open(os.path.join(root, "oracle", "_ref.MANIFEST")))
    assert set(man["artefacts_sha256"]) == {"gortt", "gortt_fp", "libgortt_ref.so"}
    assert man["reference_build"] == man["artefacts_sha256"]["libgortt_ref.so"][:16]
    ref = os.path.join(root, "oracle", "_ref")
    if not os.path.exists(os.path.join(ref, "libgortt_ref.so")):
        pytest.skip("oracle/_ref is not built here (clean clone without /root/reference)")
    for name, want in man["artefacts_sha256"].items():
        got = hashlib.sha256(open(os.path.join(ref, name), "rb").read()).hexdigest()
        assert got == want, "oracle/_ref/%s is not the build of oracle/_ref.MANIFEST: run `make -C oracle ref`" % name
