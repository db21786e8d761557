"""The algebra behind the stream family's sample (gort_amd/csrc/gort_device.h: line_terms(), stream_band(), stream_sample() - 13 scalars
per line, 12 per band, 22 instructions + a reciprocal) restated in numpy and held against the LUT family's form (sun_terms() + dot5(),
the reference's own grouping of gortt.c:484-557 and gortt_brdf.c:348-365, 467-471, 552, 616-634, which the goldens pin) in extended
precision: a CPU check of the DERIVATION - the regrouping around 1 / (1 - x^2), the factor 1 + 2 mu carried by the line's weights,
n1 = c1 - c1 x - on inputs far wider than any canopy produces.  No GPU, no library: the device code is what tests/test_stream_forms.py
and tests/test_gpu_parity.py hold to the oracle."""
import numpy as np

LD = np.longdouble


def band_terms(rs, rl, tl, k, elai, k_open, k_openep, T=np.float64):
    """lambda_table_kernel (gort_tables.hip): the band-only two-stream closed forms (gortt_brdf.c:348-634 hoisted)."""
    rs, rl, tl, k, elai, k_open, k_openep = (np.asarray(v, T) for v in (rs, rl, tl, k, elai, k_open, k_openep))
    omega = rl + tl
    gam = np.sqrt(1 - omega)
    Rff = (1 - gam) / (1 + gam)
    Tff = np.exp(-(2 * gam * k * elai))
    RT = Rff * Tff
    tff = Tff * (1 - Rff * Rff) / (1 - RT * RT)
    pff = Rff * (1 - Tff * Tff) / (1 - RT * RT)
    kopen = k_open + k_openep
    tpff = tff * (1 - kopen) + kopen
    gfun = -(T(4) / T(9)) * (rl - tl) / omega
    mgk = (rs / (1 - rs * pff)) * (tpff - k_open)
    return dict(gam=gam, omega=omega, Rff=Rff, Tff=Tff, tff=tff, pff=pff, rs=rs, mgk=mgk, Zf=(tpff - k_openep) * rs, Tf=tpff * mgk,
                B=(1 - omega) * omega * (1 - gfun))


def lut_family(t, aC, aB, aZ, aG, aT, fd, mu, t0, tp0, eps, ko, kep):
    """sun_terms() + dot5() (gort_device.h): the five (sun zenith, band) numbers, then the five-term sum."""
    g2 = 2 * t["gam"] * mu
    inv = 1 / ((1 + g2) * (1 - g2))
    Rdf = (1 - t["gam"]) * ((1 - g2) * inv)
    Tdf = (t["omega"] / 2) * ((1 + 2 * mu) * inv) * (t["Tff"] - t0)
    X = t0 * Rdf + Tdf * t["Rff"]
    tdf = Tdf - t["pff"] * X
    pdf = Rdf - t["tff"] * X
    tpdf = tdf * (1 - tp0)
    G = fd * t["rs"] + (1 - fd) * t["rs"]
    Z = fd * ((tpdf + eps) * t["rs"]) + (1 - fd) * t["Zf"]
    Td = (tpdf + tp0) * t["mgk"]
    Tt = fd * Td + (1 - fd) * t["Tf"]
    kk = kep + ko
    CfG = (kk * G + (1 - kk) * Z) * kep
    C0 = fd * (pdf + Td) + (1 - fd) * (t["pff"] + CfG + t["Tf"])
    return aC * C0 + aB * t["B"] + aZ * Z + aG * G + aT * Tt


def stream_family(t, aC, aB, aZ, aG, aT, fd, mu, t0, tp0, eps, ko, kep):
    """line_terms() + stream_band() + stream_sample() (gort_device.h), operation for operation (in float64)."""
    kk = kep + ko
    omfd = 1.0 - fd
    cfk = (aC * omfd) * kep
    Zc = cfk * (1.0 - kk) + aZ
    sCT = aC + aT
    p1, p2 = fd * sCT, Zc * fd
    omtp0 = 1.0 - tp0
    m = 1.0 + 2.0 * mu
    alpha = aC * fd
    am, P1m, P2m = alpha * m, (p1 * omtp0) * m, (p2 * omtp0) * m
    Q1, Q2, Q3, Q4, Q5, Q6 = tp0 * p1, p2 * eps + (cfk * kk + aG), Zc * omfd, omfd * sCT, aC * omfd, aB
    t0m = t0 / m
    # stream_band()
    g2, c1, c2 = 2.0 * t["gam"], 1.0 - t["gam"], t["omega"] / 2.0
    cT = c2 * t["Tff"]
    # stream_sample()
    x = g2 * mu
    inv = 1.0 / (1.0 - x * x)
    n1 = c1 - c1 * x
    n2 = cT - c2 * t0
    W = P2m * t["rs"] + P1m * t["mgk"]
    S = W * t["pff"] + am * t["tff"]
    A = alpha - t0m * S
    Bc = W - t["Rff"] * S
    num = n1 * A + n2 * Bc
    lin = Q6 * t["B"] + (Q5 * t["pff"] + (Q4 * t["Tf"] + (Q3 * t["Zf"] + (Q2 * t["rs"] + Q1 * t["mgk"]))))
    return inv * num + lin


def draw(n, rng):
    rl, tl = rng.uniform(0.01, 0.55, n), rng.uniform(0.005, 0.44, n)           # omega < 1
    band = dict(rs=rng.uniform(0.01, 0.6, n), rl=rl, tl=tl, k=rng.uniform(0.3, 0.8, n), elai=rng.uniform(0.1, 8.0, n),
                k_open=rng.uniform(0.0, 0.6, n), k_openep=rng.uniform(0.0, 0.3, n))
    line = dict(aC=rng.uniform(0, 1, n), aB=rng.uniform(0, 2, n), aZ=rng.uniform(0, 1, n), aG=rng.uniform(0, 1, n), aT=rng.uniform(0, 1, n),
                fd=rng.uniform(0, 1, n), mu=rng.uniform(1e-6, 1, n), t0=rng.uniform(0, 1, n), tp0=rng.uniform(0, 1, n), eps=rng.uniform(0, 0.5, n))
    return band, line


def test_the_stream_familys_sample_is_the_lut_familys_regrouped():
    rng = np.random.default_rng(2101)
    band, line = draw(400000, rng)
    ko, kep = band["k_open"], band["k_openep"]
    t64 = band_terms(**band)
    exact = lut_family(band_terms(T=LD, **band), *(LD(line[k]) for k in ("aC", "aB", "aZ", "aG", "aT", "fd", "mu", "t0", "tp0", "eps")), LD(ko), LD(kep))
    a = stream_family(t64, *(line[k] for k in ("aC", "aB", "aZ", "aG", "aT", "fd", "mu", "t0", "tp0", "eps")), ko, kep)
    b = lut_family(t64, *(line[k] for k in ("aC", "aB", "aZ", "aG", "aT", "fd", "mu", "t0", "tp0", "eps")), ko, kep)
    x = 2 * t64["gam"] * line["mu"]
    away = np.abs(1 - x * x) > 1e-3                                             # 2 gamma mu = 1 is singular in either form (and in the reference)
    scale = np.maximum(np.abs(np.asarray(exact, np.float64)), 1e-3)
    err_stream = np.abs(a - np.asarray(exact, np.float64))[away] / scale[away]
    err_lut = np.abs(b - np.asarray(exact, np.float64))[away] / scale[away]
    assert away.mean() > 0.99
    # both groupings are the same function: 1e-16 typically, the rounding of x amplified by 1 / |1 - x^2| <= 1e3 at worst
    assert err_stream.max() < 5e-11 and np.median(err_stream) < 1e-15, (err_stream.max(), np.median(err_stream))
    assert err_lut.max() < 5e-11 and np.median(err_lut) < 1e-15, (err_lut.max(), np.median(err_lut))
    # and the regrouped form is no worse conditioned than the reference's own
    assert err_stream.max() < 8 * max(err_lut.max(), 1e-14), (err_stream.max(), err_lut.max())
    far = np.abs(1 - x * x)[away] > 0.1                                         # away from the pole what is left is cancellation between the terms
    assert err_stream[far].max() < 5e-12 and err_stream[far].max() < 8 * max(err_lut[far].max(), 1e-14), (err_stream[far].max(), err_lut[far].max())


def test_near_the_singular_line_the_two_forms_blow_up_together():
    """2 gamma mu -> 1: R_df and T_df share the pole; the forms agree relative to their own (huge) value."""
    rng = np.random.default_rng(7)
    band, line = draw(20000, rng)
    t64 = band_terms(**band)
    line["mu"] = np.minimum((1 + rng.uniform(-1e-6, 1e-6, 20000)) / (2 * t64["gam"]), 1.0)
    ok = 2 * t64["gam"] * line["mu"] != 1.0
    args = [line[k] for k in ("aC", "aB", "aZ", "aG", "aT", "fd", "mu", "t0", "tp0", "eps")]
    a = stream_family(t64, *args, band["k_open"], band["k_openep"])[ok]
    b = lut_family(t64, *args, band["k_open"], band["k_openep"])[ok]
    near = np.abs(1 - (2 * t64["gam"] * line["mu"])[ok] ** 2) < 1e-4
    assert near.sum() > 1000
    assert np.all(np.abs(a - b)[near] <= 1e-6 * np.maximum(np.abs(a), np.abs(b))[near])
