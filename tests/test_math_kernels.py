"""The bounded-range fp64 kernels of the geometry stage (gort_amd/csrc/gort_math.h: exp, log, atan, acos, sincos, sqrt,
reciprocal and quotient by Newton steps) against mpmath at 160 bits, on the CPU: the header compiles for the host with
the four hardware primitives spelled in C - the reciprocal and reciprocal-root estimates cut to 24 bits, as coarse as the
hardware's, so that the refinement steps are what is tested.  Each function within one ulp (1.5 for the by-product
1/sqrt) over the range the geometry uses it on, IEEE special values where the device library has them.

Reference call sites: gortt_brdf.c:23-100, 118-238, 638-702; gortt.c:581-588."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

mp = pytest.importorskip("mpmath")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gm(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("gm") / "libgm.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-shared", "-fPIC",
                           "-I" + os.path.join(ROOT, "gort_amd", "csrc"), os.path.join(ROOT, "tests", "support", "math_kernels_host.cpp"),
                           "-o", so])
    return C.CDLL(so)


def _call(lib, name, *arrays):
    outs = [np.empty_like(arrays[0]) for _ in range(2 if name in ("gm_sincos", "gm_root_and_inverse") else 1)]
    args = [a.ctypes.data_as(C.c_void_p) for a in list(arrays) + outs] + [C.c_long(arrays[0].size)]
    getattr(lib, name)(*args)
    return outs if len(outs) > 1 else outs[0]


def _ulps(y, exact):
    """|y - exact| in units of the last place of `exact` (subnormal spacing below 2^-1022)."""
    worst = 0.0
    for yi, t in zip(y, exact):
        if t == 0:
            assert yi == 0
            continue
        e = max(int(mp.floor(mp.log(abs(t), 2))), -1022)
        worst = max(worst, float(abs(mp.mpf(float(yi)) - t) / mp.mpf(2) ** (e - 52)))
    return worst


def test_kernels_within_one_ulp(gm):
    mp.mp.prec = 160
    rng = np.random.default_rng(20261004)
    n = 3000
    f = lambda x: mp.mpf(float(x))
    # exp: optical depths (negative, to underflow) and Kuusk's positive exponent
    x = np.concatenate([rng.uniform(-745, 709, n), rng.uniform(-40, 3, n), [0.0, -0.0, 1e-300, -1e-10, -745.1, 709.7]])
    assert _ulps(_call(gm, "gm_exp", x), [mp.exp(f(v)) for v in x]) <= 1.0
    # log: gap probabilities in (0, 1], down to subnormals
    x = np.concatenate([rng.uniform(0, 1, n), 10 ** rng.uniform(-320, 300, n), [1.0, 0.5, 2.0, 5e-324]])
    assert _ulps(_call(gm, "gm_log", x), [mp.log(f(v)) for v in x]) <= 1.0
    # atan: (b/r) tan(zenith), any magnitude; the four break points
    x = np.concatenate([rng.uniform(-3, 3, n), 10 ** rng.uniform(-10, 18, n), [0.4375, 0.6875, 1.1875, 2.4375]])
    assert _ulps(_call(gm, "gm_atan", x), [mp.atan(f(v)) for v in x]) <= 1.0
    # acos on [-1, 1], dense at both ends (the overlap function's argument is clamped there)
    x = np.concatenate([rng.uniform(-1, 1, n), 1 - 10 ** rng.uniform(-16, 0, n), -1 + 10 ** rng.uniform(-16, 0, n), [0.5, -0.5, 0.0, 1.0, -1.0]])
    assert _ulps(_call(gm, "gm_acos", x), [mp.acos(f(v)) for v in x]) <= 1.0
    x = np.concatenate([rng.uniform(0, 1, n), 1 - 10 ** rng.uniform(-16, 0, n), [0.5, 0.0, 1.0, 0.49999999999999994]])
    assert _ulps(_call(gm, "gm_acos_unit", x), [mp.acos(f(v)) for v in x]) <= 1.0
    # sine and cosine up to the bound of the one-step reduction
    x = np.concatenate([rng.uniform(-7, 7, n), rng.uniform(-262144, 262144, n)])
    s, c = _call(gm, "gm_sincos", x)
    assert _ulps(s, [mp.sin(f(v)) for v in x]) <= 1.5 and _ulps(c, [mp.cos(f(v)) for v in x]) <= 1.5
    assert np.array_equal(_call(gm, "gm_cos", x), c)
    # at multiples of pi/2 the tiny one of the two is absolutely, not relatively, accurate
    x = np.arange(0, 64) * (np.pi / 2)
    s, c = _call(gm, "gm_sincos", x)
    assert max(abs(float(mp.mpf(float(a)) - mp.sin(f(v)))) for a, v in zip(s, x)) < 2e-16
    # square root (correctly rounded in all these cases), reciprocal root, reciprocal, quotient
    x = np.concatenate([rng.uniform(0, 4, n), 10 ** rng.uniform(-200, 200, n)])
    r = _call(gm, "gm_sqrt", x)
    assert np.array_equal(r, np.sqrt(x))
    root, inv = _call(gm, "gm_root_and_inverse", x)
    assert _ulps(root, [mp.sqrt(f(v)) for v in x]) <= 1.0 and _ulps(inv, [1 / mp.sqrt(f(v)) for v in x]) <= 1.5
    a = rng.uniform(-10, 10, n) * 10 ** rng.uniform(-100, 100, n)
    b = rng.uniform(-10, 10, n) * 10 ** rng.uniform(-100, 100, n)
    assert _ulps(_call(gm, "gm_recip", b), [1 / f(v) for v in b]) <= 1.0
    assert _ulps(_call(gm, "gm_quot", a, b), [f(u) / f(v) for u, v in zip(a, b)]) <= 1.0
    assert np.array_equal(_call(gm, "gm_quot_finite", a, b), _call(gm, "gm_quot", a, b))
    # degrees -> radians as main() writes it, x * M_PI / 180.0: the division by the constant is correctly rounded
    x = np.concatenate([rng.uniform(-400, 400, 20 * n), np.arange(-720, 721) * 0.5, np.arange(-360, 361) * 1.0]) * np.pi
    assert np.array_equal(_call(gm, "gm_div180", x), x / 180.0)


def test_special_values_as_the_library(gm):
    inf, nan = np.inf, np.nan
    e = _call(gm, "gm_exp", np.array([-inf, nan, -1000.0, 1000.0, 0.0, -1e10, -1e300]))
    assert e[0] == 0 and np.isnan(e[1]) and e[2] == 0 and e[3] == inf and e[4] == 1 and e[5] == 0 and e[6] == 0
    l = _call(gm, "gm_log", np.array([0.0, -1.0, inf, nan, 1.0]))
    assert l[0] == -inf and np.isnan(l[1]) and l[2] == inf and np.isnan(l[3]) and l[4] == 0
    a = _call(gm, "gm_acos", np.array([1.0, -1.0, nan, 1.0000000000000002]))
    assert a[0] == 0 and a[1] == np.pi and np.isnan(a[2]) and np.isnan(a[3])
    t = _call(gm, "gm_atan", np.array([inf, -inf, nan, 0.0, -0.0, 1e308]))
    assert t[0] == np.pi / 2 and t[1] == -np.pi / 2 and np.isnan(t[2]) and t[3] == 0 and np.signbit(t[4]) and t[5] == np.pi / 2
    r = _call(gm, "gm_sqrt", np.array([0.0, inf, -1.0, nan]))
    assert r[0] == 0 and r[1] == inf and np.isnan(r[2]) and np.isnan(r[3])
    q = _call(gm, "gm_quot", np.array([1.0, -1.0, 0.0, inf, 1.0, nan]), np.array([0.0, 0.0, 0.0, 2.0, inf, 1.0]))
    assert q[0] == inf and q[1] == -inf and np.isnan(q[2]) and q[3] == inf and q[4] == 0 and np.isnan(q[5])
