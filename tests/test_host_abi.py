"""CPU-side checks of the product library: it loads, exports every symbol that
include/gort_amd.h declares, and its HOST precompute (canopy derivation, Price soil,
PROSPECT-D, Gauss-Legendre nodes, LUT text) agrees with the golden vectors dumped from
the real reference.  No device entry point is called here.
"""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, relerr
from gort_amd import api

TOL = 1e-13


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "gort_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gort_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(api.DECLARED_SYMBOLS), declared ^ set(api.DECLARED_SYMBOLS)
    # measurement and tuning hooks live in a header of their own, outside the drop-in boundary
    tun = open(os.path.join(ROOT, "include", "gort_amd_tuning.h")).read()
    tun = re.sub(r"/\*.*?\*/", "", tun, flags=re.S)
    tuning = set(re.findall(r"\b(gort_[a-z0-9_]+)\s*\(", tun))
    assert tuning == set(api.TUNING_SYMBOLS), tuning ^ set(api.TUNING_SYMBOLS)
    assert not (tuning & declared)
    L = api.lib()
    for name in sorted(declared | tuning):
        assert hasattr(L, name), "libgort_amd.so does not export %s" % name
    assert b"gfx950" in L.gort_version()


def test_struct_layout_matches_header():
    # sizes implied by include/gort_amd.h (all members 8-byte aligned, int32 pairs packed)
    assert C.sizeof(api.Canopy) == 8 * (6 + 1 + 1 + 1 + 1 + 21 + 15 + 91 * 4 + 2)
    assert C.sizeof(api.LeafSoil) == 8 * (7 + 4 + 1 + 2)
    assert C.sizeof(api.Grid) == 8 * 9


FLAGS = json.load(open(os.path.join(GOLDEN, "canopies_flags.json")))


def canopy_from_flags(flags):
    kw, ns, i, q08 = {}, {}, 0, False
    while i < len(flags):
        f = flags[i]
        if f == "-q08_pn_kopen":
            q08 = True; i += 1; continue
        v = float(flags[i + 1]); i += 2
        key = {"-LAI": "lai", "-favd": "favd", "-h1": "h1", "-h2": "h2", "-lambda": "lam", "-r": "r", "-b": "b"}.get(f)
        if key: kw[key] = v
        else: ns[f] = v
    if ns:
        kw["newstyle"] = (ns.get("-HB", 2.0), ns.get("-BR", 1.0), ns.get("-PCC", 0.5))
    return api.make_canopy(q08=q08, **kw)


@pytest.mark.parametrize("tag", sorted(FLAGS))
def test_canopy_init_matches_reference(tag, golden):
    g = golden("canopies.npz")
    c = canopy_from_flags(FLAGS[tag])
    sc = g[tag + "/scalars"]
    names = ["r", "b", "h1", "h2", "lambda_", "favd", "ell", "h", "elai", "tau", "z1", "z2", "lv", "favd_p",
             "tau_p", "lv_p", "z1_p", "z2_p", "h1_p", "h2_p", "dz", "ds", "dz_p", "dth"]
    for i, n in enumerate(names):
        assert getattr(c, n) == sc[i], (n, getattr(c, n), sc[i])
    assert c.k == sc[26] and c.rr == sc[27] and c.rrr == sc[28]
    assert relerr(np.array(c.theta), g[tag + "/theta"]) == 0
    assert relerr(np.array(c.theta_p), g[tag + "/theta_p"]) == 0
    assert relerr(np.array(c.height_p), g[tag + "/height_p"]) == 0


@pytest.mark.parametrize("tag", ["default", "dry", "dense", "zero_abs", "opaque"])
def test_prospect_d(tag, golden):
    s = golden("spectra.npz")
    RT = api.prospect_d(*s["prospect/%s/params" % tag])
    assert relerr(RT, s["prospect/%s/RT" % tag]) <= TOL


def test_spectra(golden):
    s = golden("spectra.npz")
    rs, rl, tl = api.spectra(s["interp/default/wl"])
    assert relerr(rs, s["interp/default/rsoil"]) <= TOL
    assert relerr(rl, s["interp/default/rleaf"]) <= TOL
    assert relerr(tl, s["interp/default/tleaf"]) <= TOL
    pr = s["interp/alt/prospect"]
    ls = api.leaf_soil(prospect=dict(N=pr[0], Cab=pr[1], Car=pr[2], Anth=pr[3], Cbrown=pr[4], Cw=pr[5], Cm=pr[6]),
                       rsl=s["interp/alt/rsl"])
    rs, rl, tl = api.spectra(s["interp/alt/wl"], ls)
    assert relerr(rs, s["interp/alt/rsoil"]) <= TOL
    assert relerr(rl, s["interp/alt/rleaf"]) <= TOL
    assert relerr(tl, s["interp/alt/tleaf"]) <= TOL
    rs, rl, tl = api.spectra([500.0, 900.0], api.leaf_soil(alb_leaf=0.9, alb_soil=0.2))
    assert (rs == 0.2).all() and (rl == 0.45).all() and (tl == 0.45).all()


@pytest.mark.parametrize("bad", [[399.0], [2500.5], [500.0, 2600.0], [500.0, float("nan")]])
def test_wavelength_range_error(bad):
    with pytest.raises(api.GortError) as e:
        api.spectra(bad)
    assert e.value.code == api.ERANGE
    # message text of the reference (gortt.c:1299-1302), printed verbatim by the CLI
    assert "wavlength out of range (400-2500)" in str(e.value)


def test_gauleg(golden):
    s = golden("spectra.npz")
    x, w = api.gauleg(32)
    assert relerr(x, s["gauleg32/x"]) <= TOL and relerr(w, s["gauleg32/w"]) <= TOL


def test_lut_text_roundtrip(golden, tmp_path):
    """gort_lut_format writes the byte format of `gortt -W`; gort_lut_read parses a file the
    REAL reference wrote."""
    cases = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "cli_cases.json")))}
    ref_text = cases["lut_default"]["stdout"]
    c = api.make_canopy(lai=4.0)
    p = tmp_path / "lut.dat"
    p.write_text(ref_text)
    api.lut_read(str(p), c)
    g = golden("canopies.npz")
    # "%0.40f" keeps 40 decimals: an ABSOLUTE resolution of 1e-40 (values below it read back as 0)
    assert np.abs(np.array(c.p_n0)[:90] - g["default_lai4/p_n0"][0][:90]).max() <= 1e-40
    assert np.abs(np.array(c.epgap)[:90] - g["default_lai4/epgap0"][:90]).max() <= 1e-40
    assert c.p_n0[90] == 0.0 and c.epgap[90] == 0.0          # index 90 is never in the file
    assert relerr(np.array([c.k_open, c.k_openep]), g["default_lai4/kk"]) <= 1e-15
    # re-emitting the parsed tables reproduces the reference's bytes
    assert api.lut_text(c) == ref_text
    with pytest.raises(api.GortError) as e:
        api.lut_read("/nonexistent/lut.dat", c)
    assert e.value.code == api.EIO


def test_device_entry_points_fail_loudly_without_gpu():
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(api.GortError) as e:
        api.Engine()
    assert e.value.code == api.ENODEVICE
    with pytest.raises(api.GortError):
        api.gap_probabilities(api.make_canopy(lai=4.0))
    with pytest.raises(api.GortError):
        api.set_device(0)                                     # bench.py pins every rank to its GPU through this


def test_format_f6_equals_printf():
    """gort_format_f6 writes the bytes of printf("%f") (Python's '%f' is the same correctly rounded,
    ties-to-even conversion as glibc's) for every magnitude the rows contain, plus the edge cases."""
    import ctypes as C
    L = api.lib()
    L.gort_format_f6.argtypes = [C.c_double, C.c_char_p]
    buf = C.create_string_buffer(400)

    def f6(x):
        n = L.gort_format_f6(x, buf)
        return buf.raw[:n].decode()

    rng = np.random.default_rng(0)
    vals = np.concatenate([
        rng.uniform(0, 1, 200000), rng.uniform(-90, 360, 50000), 10.0 ** rng.uniform(-12, 9.5, 100000),
        -(10.0 ** rng.uniform(-12, 9.5, 20000)), rng.integers(0, 2 ** 31, 20000) / 128.0,     # exact ties .5e-6
        (rng.integers(0, 10 ** 6, 50000) + 0.5) * 1e-6,                                      # decimal near-ties
        np.array([0.0, -0.0, 1e-7, 5e-7, 4.999999999999999e-7, 5.000000000000001e-7, -1e-9, 0.0078125, 0.0234375,
                  1.0, 0.9999995, 0.99999949999999, 123456.7890125, 3.9999999e9, 4.0e9, 1e10, 1e300, -1e300,
                  2.5e-6, 3.5e-6, 1.5e-6, 0.5e-6, np.inf, -np.inf]),
    ])
    for x in vals:
        want = "%f" % x
        assert f6(float(x)) == want, (repr(float(x)), f6(float(x)), want)
    assert f6(float("nan")) == "-nan" and f6(-float("nan")) == "-nan"


def test_ensemble_state_vector_to_gortt_inputs():
    """gort_amd.ensemble.member_inputs: a state vector becomes what `gortt -HB .. -BR .. -PCC .. -LAI ..` with
    PROSPECT-D / Price parameters would build (float32 flag parse included); host-only, no GPU."""
    from gort_amd import api
    from gort_amd.ensemble import DEFAULT, STATE, f32, member_inputs
    assert STATE[:4] == ("HB", "BR", "PCC", "LAI") and set(STATE) == set(DEFAULT)
    state = dict(DEFAULT, HB=1.7, BR=2.4, PCC=0.55, LAI=3.1, Cab=42.0, rsl1=0.31)
    c, leaf = member_inputs(state)
    ref = api.make_canopy(newstyle=(f32(1.7), f32(2.4), f32(0.55)), lai=f32(3.1))
    for name in ("favd", "r", "b", "h1", "h2", "lambda_"):
        assert getattr(c, name) == getattr(ref, name), name
    assert leaf.Cab == 42.0 and leaf.N == DEFAULT["N"] and leaf.rsl[0] == 0.31 and leaf.rsl[1] == 0.1
    c2, _ = member_inputs([state[k] for k in STATE])            # sequence form, STATE order
    assert c2.favd == c.favd and c2.h2 == c.h2
    # gort_canopy_newstyle takes the flags as float, like the reference's parser: rounding them beforehand is idempotent
    c3 = api.make_canopy(newstyle=(1.7, 2.4, 0.55), lai=3.1)
    assert c3.favd == c.favd and c3.r == c.r


def test_index_math_selftest():
    """Host-side self-test of the LUT kernel's index arithmetic: the multiply-shift divisions equal '/', and the
    XCD duty mapping sends the workgroups of a launch onto every logical block exactly once (300 weightings)."""
    assert api.lib().gort_selftest_index_math() == 0


def test_error_codes_are_negative_and_null_handles_fail_cleanly():
    """gort_engine_xcd_mapping returned +3/+5 for device errors (it negated an already negative code).
    Every entry point that reports through its return value gives a NEGATIVE GORT_E* code for a null handle,
    without touching a device; the new pipe / pinned-memory surface included."""
    import ctypes as C
    L = api.lib()
    assert L.gort_engine_xcd_mapping(None) == api.EINVAL
    assert L.gort_engine_stream_form(None) == api.EINVAL
    assert L.gort_engine_last_stream_ms(None) < 0
    assert L.gort_engine_synchronize(None) == api.EINVAL
    w = (C.c_int * 8)()
    assert L.gort_engine_xcd_weights(None, w) == api.EINVAL
    assert L.gort_pipe_submit(None, 1) == api.EINVAL and L.gort_pipe_release(None) == api.EINVAL
    chunk = api.PipeChunk()
    assert L.gort_pipe_wait(None, C.byref(chunk)) == api.EINVAL
    h = C.c_void_p()
    assert L.gort_pipe_create(None, 10, 2, 0, C.byref(h)) == api.EINVAL and not h.value
    L.gort_pipe_destroy(None)
    L.gort_host_free(None)
    if api.device_count() == 0:
        assert L.gort_set_device(0) == api.ENODEVICE
        assert L.gort_get_device() < 0


def test_format_f6_row_equals_single_values():
    """gort_format_f6_row: n values, each followed by a space, the bytes of printf("%f ") per value - the fast path
    (hardware FMA + ROUNDSD where the CPU has them) and the one-by-one path for rows with huge values."""
    import ctypes as C
    L = api.lib()
    L.gort_format_f6_row.restype = C.c_long
    L.gort_format_f6_row.argtypes = [C.c_void_p, C.c_long, C.c_char_p, C.c_size_t]
    rng = np.random.default_rng(3)
    rows = [rng.uniform(0, 1, 2101), rng.uniform(-400, 400, 50), np.array([0.0, -0.0, 0.5e-6, 1.5e-6, 2.5e-6, 0.9999995, 3.9999e9]),
            np.array([np.nan, 1e300, -5e9, 4.0e9, 1.0, np.inf]), np.zeros(0)]
    for v in rows:
        v = np.ascontiguousarray(v, dtype=np.float64)
        buf = C.create_string_buffer(max(24 * v.size + 8, 4096))
        n = L.gort_format_f6_row(v.ctypes.data_as(C.c_void_p), v.size, buf, len(buf))
        want = "".join(("-nan" if np.isnan(x) else "%f" % x) + " " for x in v)
        assert n == len(want) and buf.raw[:n].decode() == want
    tiny = C.create_string_buffer(16)
    big = np.array([1e300, 1.0])
    assert L.gort_format_f6_row(big.ctypes.data_as(C.c_void_p), 2, tiny, len(tiny)) == api.EINVAL


def test_gap_table_file_cache_keyed_on_crown_geometry(golden, tmp_path):
    """`-W`/`-P` generalised to a cache keyed on crown geometry.  No GPU: the tables stored are
    the REFERENCE's own (golden canopies.npz, %.17g), the round trip through the hex-float file keeps every bit, the
    file is a valid `-P` file, and another geometry / a damaged file is a miss, not a wrong hit."""
    g = golden("canopies.npz")
    c = api.make_canopy(lai=4.0)
    for t in range(91):
        c.p_n0[t] = float(g["default_lai4/p_n0"][0][t])
        c.epgap[t] = float(g["default_lai4/epgap0"][t]) if t < 90 else 0.0
    c.k_open, c.k_openep = (float(v) for v in g["default_lai4/kk"])
    key = api.canopy_key(c)
    assert key == api.canopy_key(api.make_canopy(lai=4.0)) and key != 0
    d = str(tmp_path)
    fresh = api.make_canopy(lai=4.0)
    assert api.lut_cache_load(d, fresh) is False                      # empty directory
    api.lut_cache_store(d, c)
    path = os.path.join(d, "gap-%016x.lut" % key)
    assert os.path.exists(path) and [f for f in os.listdir(d)] == [os.path.basename(path)]   # no temporary left behind
    assert api.lut_cache_load(d, fresh) is True
    assert list(fresh.p_n0) == list(c.p_n0) and list(fresh.epgap) == list(c.epgap)           # bit for bit
    assert (fresh.k_open, fresh.k_openep) == (c.k_open, c.k_openep)
    assert min(v for v in c.p_n0 if v > 0) < 1e-40                    # values "%0.40f" would have flushed to zero survive
    # the same file through the -P reader: rows 0..89 and the KOpen line; the key line stops fscanf as EOF would
    via_p = api.make_canopy(lai=4.0)
    api.lut_read(path, via_p)
    assert list(via_p.p_n0)[:90] == list(c.p_n0)[:90] and (via_p.k_open, via_p.k_openep) == (c.k_open, c.k_openep)
    # the key covers exactly what the tables depend on
    other = api.make_canopy(lai=3.0)
    assert api.canopy_key(other) != key and api.lut_cache_load(d, other) is False
    q08 = api.make_canopy(lai=4.0)
    q08.use_q08 = 1
    assert api.canopy_key(q08) != key
    beta = api.make_canopy(lai=4.0)
    beta.beta, beta.use_user_beta = 0.3, 1                            # -beta does not enter the gap tables
    assert api.canopy_key(beta) == key
    # a file under the right name whose key line belongs to another geometry (hash collision / renamed file): miss
    text = open(path).read()
    os.rename(path, os.path.join(d, "gap-%016x.lut" % api.canopy_key(other)))
    assert api.lut_cache_load(d, other) is False
    # truncated file: miss
    open(path, "w").write(text[:len(text) // 2])
    assert api.lut_cache_load(d, api.make_canopy(lai=4.0)) is False
    with pytest.raises(api.GortError):
        api.lut_cache_store(os.path.join(d, "no", "such", "dir"), c)


def test_cli_usage_and_help_need_no_gpu():
    """`gortt -u` prints the reference's usage text byte for byte (golden case from the real reference) and `--help`
    the same plus the list of this implementation's extensions; both exit 0 before any device call."""
    import subprocess
    cases = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "cli_cases.json")))}
    u = subprocess.run([api.GORTT_BIN, "-u"], capture_output=True, timeout=60)
    assert u.returncode == 0 and u.stdout == b""
    assert u.stderr.decode("latin-1").replace(api.GORTT_BIN, "gortt") == cases["usage"]["stderr"]
    h = subprocess.run([api.GORTT_BIN, "--help"], capture_output=True, timeout=60)
    assert h.returncode == 0 and h.stderr.startswith(u.stderr)
    for flag in (b"--binary-in", b"--binary-out", b"--lut-hex", b"--lut-cache DIR", b"--gpus N"):
        assert flag in h.stderr[len(u.stderr):]


def test_host_entry_points_under_sanitizers(tmp_path):
    """tools/probes/host_fuzz.cpp: gort_host.cpp (the part of the library without device code) compiled with
    AddressSanitizer + UBSan and driven with hostile inputs - the exact "%f" formatter over random bit patterns against
    snprintf, row buffers at their capacity limits, the `-P` reader and the gap-table cache on truncated / corrupted /
    padded files (a damaged file must be a miss, never a wrong hit), wavelengths at and beyond the range incl. NaN."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = ROOT
    exe = str(tmp_path / "host_fuzz")
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-fno-omit-frame-pointer", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "gort_amd", "csrc"),
                            '-DGORT_DATA_DIR="%s"' % os.path.join(root, "gort_amd", "data"),
                            os.path.join(root, "gort_amd", "csrc", "gort_host.cpp"), os.path.join(root, "tools", "probes", "host_fuzz.cpp"),
                            "-o", exe], capture_output=True, timeout=600)
    assert build.returncode == 0, build.stderr.decode()[-3000:]
    run = subprocess.run([exe, "0.1"], capture_output=True, timeout=600, cwd=str(tmp_path))
    assert run.returncode == 0 and b"host_fuzz: ok" in run.stdout, (run.stdout[-2000:], run.stderr[-3000:])


def test_degenerate_crown_geometry_is_refused_where_the_gap_probabilities_would_be_computed():
    """The reference allocates a negative size, loops for ever or reports negative volumes for such crowns; the library
    returns EINVAL from gort_canopy_check_geometry (called by every entry point that computes gap probabilities, before
    anything reaches the device) - gort_canopy_init itself stays as permissive as gortt_init_params, because with
    `-P file` the reference never looks at the crown that closely."""
    L = api.lib()
    L.gort_canopy_check_geometry.argtypes = [C.POINTER(api.Canopy)]
    assert L.gort_canopy_check_geometry(C.byref(api.make_canopy(lai=4.0))) == 0
    assert L.gort_canopy_check_geometry(C.byref(api.make_canopy(newstyle=(2.0, 2.0, 0.6), lai=3.3))) == 0
    for kw in (dict(r=0.0), dict(r=-1.0), dict(b=0.0), dict(b=-2.0), dict(h1=5.0, h2=5.0), dict(h1=9.0, h2=5.0),
               dict(r=float("nan")), dict(r=float("inf")), dict(h2=float("inf")), dict(newstyle=(0.0, 2.0, 0.6)),
               dict(newstyle=(2.0, 0.0, 0.6)), dict(newstyle=(2.0, -1.0, 0.6))):
        c = api.make_canopy(lai=4.0, **kw)                                # init accepts it ...
        assert L.gort_canopy_check_geometry(C.byref(c)) == api.EINVAL, kw  # ... the check does not
        assert b"invalid crown geometry" in L.gort_last_error()
        if api.device_count() == 0:
            with pytest.raises(api.GortError) as e:                      # refused before the missing device is noticed
                api.gap_probabilities(c)
            assert e.value.code == api.EINVAL
    # lambda and favd may be anything: the reference carries NaN through
    assert L.gort_canopy_check_geometry(C.byref(api.make_canopy(lai=float("inf")))) == 0
    assert L.gort_canopy_check_geometry(C.byref(api.make_canopy(lai=4.0, lam=-1.0))) == 0


def test_prospect_d_random_parameter_vectors(golden):
    """96 random PROSPECT-D parameter vectors through the reference's own Fortran (tests/golden/prospect_fuzz.npz: the
    usual ranges, N = 1, no absorbers, absorbers strong enough for every branch of the exponential-integral fit, negative
    contents): the host implementation reproduces R and T on every 7th band, NaN pattern included."""
    g = golden("prospect_fuzz.npz")
    worst, nan_rows = 0.0, 0
    for params, want in zip(g["params"], g["RT"]):
        RT = api.prospect_d(*params)
        got = np.stack([RT[:2101][g["bands"]], RT[2101:][g["bands"]]])
        assert np.array_equal(np.isnan(got), np.isnan(want)), params
        nan_rows += int(np.isnan(want).any())
        m = np.isfinite(want)
        if m.any():
            worst = max(worst, float(np.max(np.abs(got[m] - want[m]) / np.maximum(np.abs(want[m]), 1e-12))))
    assert worst <= 1e-12, worst
    assert nan_rows >= 5                                     # the fixture does reach the NaN-producing corners
