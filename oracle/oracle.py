"""ctypes binding of oracle/libgort_oracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product (gort_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libgort_oracle.so")
D = C.c_double
NL, NTH, NB = 15, 91, 2101


class Canopy(C.Structure):
    """Mirror of gort_o_canopy (oracle/gort_oracle.h)."""
    _fields_ = [
        ("r", D), ("b", D), ("h1", D), ("h2", D), ("lambda_", D), ("favd", D),
        ("use_user_beta", C.c_int), ("beta", D), ("use_user_fd", C.c_int), ("fd_user", D),
        ("ell", D), ("rr", D), ("rrr", D), ("h", D), ("k", D), ("elai", D), ("tau", D),
        ("z1", D), ("z2", D), ("lv", D),
        ("favd_p", D), ("tau_p", D), ("lv_p", D), ("z1_p", D), ("z2_p", D), ("h1_p", D), ("h2_p", D),
        ("dz", D), ("ds", D), ("dz_p", D), ("dth", D),
        ("nlayers", C.c_int), ("nth", C.c_int), ("maxcrowns", C.c_int), ("nh_es", C.c_int),
        ("height", D * NL), ("height_p", D * NL), ("theta", D * NTH), ("theta_p", D * NTH),
        ("factorial", D * 31),
        ("v_g", D * (NL * NTH)), ("p_n0", D * (NL * NTH)), ("p_s0", D * (NL * NTH)),
        ("es0", D * NTH), ("epgap0", D * NTH), ("k_open0", D), ("k_openep0", D),
    ]


class Geom(C.Structure):
    _fields_ = [(n, D) for n in (
        "vza", "vaa", "sza", "saa", "raa", "vza_p", "sza_p", "fd",
        "pn0_s", "epgap_s", "pn0_v", "epgap_v", "Kc", "Kg", "Kt", "Kz")]


def build(force=False):
    if force or not os.path.exists(LIB) or \
            os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "gort_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "libgort_oracle.so"])


_lib = None
#: bench.py sets this to False: a benchmark process (GPU initialised, possibly under rocprofv3) must never start a
#: compiler; it loads the prebuilt library or fails.  Tests keep the convenience of an automatic build.
AUTO_BUILD = True


def lib():
    global _lib
    if _lib is None:
        if AUTO_BUILD:
            build()
        elif not os.path.exists(LIB):
            raise ImportError("%s missing: build it with `make -C oracle` (or __graft_entry__.build())" % LIB)
        _lib = C.CDLL(LIB)
        _lib.gort_o_crown_proj_volume.restype = D
        _lib.gort_o_get_es.restype = D
        _lib.gort_o_vol.restype = D
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(D))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def make_canopy(lai=None, newstyle=None, favd=None, r=None, b=None, h1=None, h2=None,
                lam=None, beta=None, diffuse=None, q08=False, gaps=True):
    """Canopy as the gortt flags would build it (order of application as gortt.c:1117-1131)."""
    L = lib()
    c = Canopy()
    L.gort_o_canopy_defaults(C.byref(c))
    if favd is not None: c.favd = favd
    if r is not None: c.r = r
    if b is not None: c.b = b
    if h1 is not None: c.h1 = h1
    if h2 is not None: c.h2 = h2
    if lam is not None: c.lambda_ = lam
    if newstyle is not None:
        hb, br, pcc = newstyle
        L.gort_o_canopy_newstyle(C.byref(c), C.c_float(hb), C.c_float(br), C.c_float(pcc))
    if lai is not None:
        L.gort_o_canopy_set_lai(C.byref(c), C.c_float(lai))
    if beta is not None:
        c.use_user_beta = 1
        c.beta = beta
    if diffuse is not None:
        c.use_user_fd = 1
        c.fd_user = 1.0 - diffuse
    L.gort_o_canopy_init(C.byref(c))
    if gaps:
        if q08:
            L.gort_o_gap_probabilities_q08(C.byref(c))
        else:
            rc = L.gort_o_gap_probabilities(C.byref(c))
            if rc != 0:
                raise RuntimeError("oracle: path-length index outside histogram")
    return c


def gap_tables(c):
    """(p_n0[91], epgap0[91], k_open0, k_openep0) -- the four live products."""
    return (np.array(c.p_n0[:NTH]), np.array(c.epgap0), c.k_open0, c.k_openep0)


def set_gap_tables(c, p_n0, epgap0, k_open0, k_openep0):
    for t in range(NTH):
        c.p_n0[t] = p_n0[t]
        c.epgap0[t] = epgap0[t]
    c.k_open0 = k_open0
    c.k_openep0 = k_openep0


def prospect_d(N=1.2, Cab=30., Car=10., Anth=1.0, Cbrown=0.0, Cw=0.015, Cm=0.009):
    RT = np.zeros(2 * NB)
    lib().gort_o_prospect_d(D(N), D(Cab), D(Car), D(Anth), D(Cbrown), D(Cw), D(Cm), _p(RT))
    return RT


def spectra(wl, rsl=(0.2, 0.1, 0.03726, -0.002426), prospect=None, alb_leaf=None, alb_soil=None):
    """rsoil, rleaf, tleaf at wavelengths wl (nm), as gortt.c:224-227 would fill them."""
    wl = f64(wl)
    nw = wl.size
    rsoil = np.zeros(nw); rleaf = np.zeros(nw); tleaf = np.zeros(nw)
    if alb_soil is not None:
        rsoil[:] = alb_soil
    else:
        rs = (D * 4)(*rsl)
        if lib().gort_o_price_soil(_p(wl), nw, rs, _p(rsoil)) != 0:
            raise ValueError("wavelength out of range (400-2500)")
    if alb_leaf is not None:
        rleaf[:] = alb_leaf / 2.0
        tleaf[:] = alb_leaf / 2.0
    else:
        RT = prospect_d(**(prospect or {}))
        if lib().gort_o_leaf_interp(_p(wl), nw, _p(RT), _p(rleaf), _p(tleaf)) != 0:
            raise ValueError("wavelength out of range (400-2500)")
    return rsoil, rleaf, tleaf


def rsurf_stream(c, angles_deg, rsoil, rleaf, tleaf, want_scomp=False, want_K=True):
    ang = f64(angles_deg).reshape(-1, 4)
    nA, nw = ang.shape[0], len(rsoil)
    rs, rl, tl = f64(rsoil), f64(rleaf), f64(tleaf)
    out = np.zeros((nA, nw))
    sc = np.zeros((nA, 4 * nw)) if want_scomp else None
    K = np.zeros((nA, 4)) if want_K else None
    lib().gort_o_rsurf_stream(C.byref(c), _p(ang), C.c_long(nA), nw, _p(rs), _p(rl), _p(tl),
                              _p(out), _p(sc) if want_scomp else None, _p(K) if want_K else None)
    return out, sc, K


def energy_stream(c, angles_deg, rsoil, rleaf, tleaf):
    ang = f64(angles_deg).reshape(-1, 4)
    nA, nw = ang.shape[0], len(rsoil)
    rs, rl, tl = f64(rsoil), f64(rleaf), f64(tleaf)
    out = np.zeros((nA, nw, 3))
    lib().gort_o_energy_stream(C.byref(c), _p(ang), C.c_long(nA), nw, _p(rs), _p(rl), _p(tl), _p(out))
    return out


def gauleg(n=32):
    x = np.zeros(n); w = np.zeros(n)
    lib().gort_o_gauleg(D(-1.0), D(1.0), _p(x), _p(w), n)
    return x, w


# ---- the REAL reference at function level, where its build travelled (oracle/_ref/libgortt_ref.so, oracle/Makefile) ----
REF_SO = os.path.join(HERE, "_ref", "libgortt_ref.so")


def _reference_rows_worker(args):
    flags, angles_deg, wl = args
    L = C.CDLL(REF_SO)
    argv_list = [b"gortt"] + [f.encode() for f in flags]
    argv = (C.c_char_p * len(argv_list))(*argv_list)
    L.refshim_canopy(len(argv_list), argv)
    nw = len(wl)
    w = np.ascontiguousarray(wl, dtype=np.float64)
    rs, rl, tl = np.zeros(nw), np.zeros(nw), np.zeros(nw)
    L.refshim_spectra(_p(w), nw, _p(rs), _p(rl), _p(tl))
    L.refshim_rsurf.argtypes = [D] * 5 + [C.POINTER(D)] * 4
    out, sc, K, pr = np.zeros(nw), np.zeros(4 * nw), np.zeros(4), np.zeros(4)
    rows = np.empty((len(angles_deg), nw))
    rad = np.pi / 180.0
    for i, (vza, vaa, sza, saa) in enumerate(angles_deg):
        # main()'s normalisation (gortt.c:240-279) for non-negative zeniths and azimuths within [0, 360]
        assert vza >= 0 and sza >= 0 and 0 <= vaa <= 360 and 0 <= saa <= 360
        vza, vaa, sza, saa = vza * rad, vaa * rad, sza * rad, saa * rad
        raa = saa - vaa
        raa = abs(raa - 2 * np.pi * int(0.5 + raa / (2 * np.pi)))
        L.refshim_rsurf(vza, vaa, sza, saa, raa, _p(out), _p(sc), _p(K), _p(pr))
        rows[i] = out
    return rows


def reference_rows(flags, angles_deg, wl, timeout=600):
    """rsurf[len(angles)][len(wl)] from the reference's own gortt_rsurf (any number of bands - the CLI's 999-character
    header is not in the way here), in a process of its own (the reference keeps global state and brings a Fortran
    runtime); None where the reference build is absent."""
    if not os.path.exists(REF_SO):
        return None
    import json
    import sys
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        job, res = os.path.join(d, "job.json"), os.path.join(d, "rows.npy")
        json.dump({"flags": list(flags), "angles": [[float(x) for x in a] for a in angles_deg], "wl": [float(w) for w in wl]},
                  open(job, "w"))
        code = ("import json, sys, numpy as np; sys.path.insert(0, %r); from oracle import oracle as O; j = json.load(open(%r)); "
                "np.save(%r, O._reference_rows_worker((j['flags'], j['angles'], j['wl'])))" % (os.path.dirname(HERE), job, res))
        run = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=timeout)
        if run.returncode != 0:
            raise RuntimeError("reference worker failed: " + run.stderr.decode()[-2000:])
        return np.load(res)
