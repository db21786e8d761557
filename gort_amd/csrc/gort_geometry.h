// gort_geometry.h -- the per-angle geometry of the GORT BRDF as device functions: mutual-shadowing overlap, the
// azimuth-independent terms of a (view zenith, sun zenith) pair, the areal proportions Kc/Kg/Kt/Kz with Kuusk's
// hot spot, and the angle record the expansion kernels read.  Shared by the geometry kernels (gort_geometry.hip)
// and the albedo quadrature (gort_energy.hip), so that both evaluate exactly the same functions.
// Reference: gortt_brdf.c:7-238, 638-702; gortt.c:424-449, 872-915.
#ifndef GORT_GEOMETRY_H
#define GORT_GEOMETRY_H

#include "gort_device.h"

namespace gort {
namespace {

// mutual-shadowing overlap O(theta_s', theta_v', phi)  (gortt_brdf.c:23-100)
// Unfused on purpose: for equal primed zeniths and phi = 0 the reference gets d = t^2 + t^2 - 2 t t = 0
// EXACTLY; an FMA leaves 1e-16 t^2 of product rounding in d, sqrt turns it into 1e-8 t, and Kc at 89/89 deg
// moves by 3e-8.  All geometry below keeps plain IEEE multiply/add for the same reason.
// M: the arithmetic (gort_device.h: FastMath for all lines but those at the horizon, LibMath there).
template <class M, bool PRINCIPAL>
__device__ inline double overlap(double hb, const Primed &s, const Primed &v, double cphi, double sphi)
{
#pragma clang fp contract(off)
    const double d = s.t * s.t + v.t * v.t - 2.0 * s.t * v.t * cphi;
    const double D = M::sqrt(ref_max(0.0, d));
    const double x = s.t * v.t * sphi;
    const double t2 = PRINCIPAL ? M::hypot_principal(D, x) : M::sqrt(D * D + x * x);
    const double t1 = s.sec + v.sec;
    double cos_t = M::div(hb * t2, t1);
    cos_t = ref_max(-1.0, cos_t);
    cos_t = ref_min(1.0, cos_t);
    const double t = M::acos_unit(cos_t);                 // hb, t2, t1 >= 0
    return ref_max(0.0, M::over_pi((t - M::sin_of_acos(t, cos_t) * cos_t) * t1));
}

struct GeomOut {
    double Kc, Kg, Kt, Kz, Kpg, Kpz, A;    // A = kuusk / (2 cos(sza') cos(vza'))
    SunScalars sun;
};

// Everything of one (view zenith, sun zenith) pair that does not depend on the relative azimuth:
// about 25 of the ~35 fp64 transcendentals of a tuple.  The LUT path evaluates it once per row
// (361 azimuths) into LDS; the stream path once per line.
struct RowTerms {
    Primed v, s;
    double sin_vz, cos_vz, sin_sz, cos_sz;
    double cov, hb, t1, es, ev, Gv;
    double fF0, fFpi, beta;
    double eps_s, eps_v, ls, lv, h1, kf;
    SunScalars sun;
    int horizon;                           // the line takes the reference's route (LibMath)
};

// Restates the azimuth-independent parts of gortt_kg/gortt_kc/gortt_kc_fFbeta (gortt_brdf.c:7-238),
// gortt_set_zenith_dependant_probabilities (gortt.c:872-915) and gortt_kuusk (gortt_brdf.c:638-702).
// r.sin_vz ... r.cos_sz are set by the caller.
template <class M>
__device__ void row_terms_with(const gort_canopy &c, double vza, double sza, RowTerms &r)
{
#pragma clang fp contract(off)
    const double ell = c.ell;                             // b / r, formed by gort_canopy_init as the reference does (gortt.c:641)
    r.v = M::prime(ell, M::div(r.sin_vz, r.cos_vz));
    r.s = M::prime(ell, M::div(r.sin_sz, r.cos_sz));
    const Primed &v = r.v, &s = r.s;
    r.cov = c.lambda * PI * c.rr;                         // lambda pi r^2
    r.hb = M::div(c.h, c.b);
    r.t1 = s.sec + v.sec;
    const double cov = r.cov, t1 = r.t1;

    // principal-plane overlaps (Kc is interpolated between phi = 0 and pi, gortt_brdf.c:143-159)
    const double O_0 = overlap<M, true>(r.hb, s, v, 1.0, 0.0);
    const double O_pi = overlap<M, true>(r.hb, s, v, -1.0, 1.2246467991473532e-16);   // sin(M_PI) in double
    const double Kg0 = M::exp(-(cov * (t1 - O_0)));
    const double Kgpi = M::exp(-(cov * (t1 - O_pi)));

    const double xs = cov * s.sec, xv = cov * v.sec;
    r.es = M::exp(-xs);
    r.ev = M::exp(-xv);
    const double Mi = 1.0 - M::div(1.0 - r.es, xs);
    const double Mv = 1.0 - M::div(1.0 - r.ev, xv);
    const double theta_Mi = M::acos(1.0 - 2.0 * Mi);
    r.Gv = PI * c.rr * v.sec;
    const double Gv = r.Gv;

    // f*F on the principal plane, phi = 0 and phi = pi
    const bool view_steeper = fabs(vza) > fabs(sza);
    double fF[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const double cphi = q ? -1.0 : 1.0;
        const double Oq = q ? O_pi : O_0, Kgq = q ? Kgpi : Kg0;
        const double ph = v.c * s.c + v.s * s.s * cphi;
        const double Gam = PI * c.rr * (t1 - Oq);
        const double Gc = Gv * 0.5 * (1.0 + ph);
        const double F = M::div(Gc, Gam);
        const double Mq = 1.0 - M::div(1.0 - Kgq, c.lambda * Gam);
        const double PiMi = (1 - M::cos(theta_Mi * (1 - M::over_pi(s.ang - v.ang * cphi)))) / 2.0;
        const double PvMv = Mv - (1.0 - M::cos_of_difference(v, s, cphi, ph)) / 2.0;
        // phi = pi lies in (90,270) deg -> Po = PvMv; phi = 0 -> by steepness (gortt_brdf.c:219-221)
        const double Po = (q == 1) ? PvMv : (view_steeper ? PiMi : PvMv);
        const double f = M::div(F * (1.0 - M::div(Gv * (PvMv + PiMi - Po), Gc)), 1.0 - Mq);
        fF[q] = f * F;
    }
    r.fF0 = fF[0];
    r.fFpi = fF[1];

    if (c.use_user_beta) {
        r.beta = c.beta;
    } else if (s.ang < 0.000000001) {
        r.beta = 0.0;
    } else {
        const double Dd = c.r * M::cot_half(s);             // D = r / tan(sza'/2), gortt_brdf.c:196
        const double dh = M::div(c.h2 - c.h1, Dd);
        const double lg = c.lambda * Gv;
        r.beta = M::div(lg, lg + dh) * M::div(1.0 - M::exp(-lg - dh), 1.0 - M::exp(-lg));
    }

    // zenith-dependent gap probabilities; path lengths of Kuusk's hot spot
    double pn0_v;
    r.sun = sun_scalars<M>(c, sza, r.cos_sz, s);
    r.eps_s = r.sun.eps;
    gap_lookup(c, vza, pn0_v, r.eps_v);
    r.kf = c.k * c.favd;
    r.ls = M::div_ieee(-M::log(r.eps_s), r.kf);                // favd = 0 (-LAI 0): inf or NaN as the reference's
    r.lv = M::div_ieee(-M::log(r.eps_v), 0.5 * c.favd);
    r.h1 = (r.ls * r.lv) > 0.0 ? M::sqrt(r.ls * r.lv) : 0.0;
}

// The sine and cosine of the two ZENITHS are the device library's in both arithmetics.  In the exact hot-spot
// direction (vza = sza, raa = 0) Kuusk's cos xi is cos^2 + sin^2 of one angle, and whether that sum rounds to 1 or
// to 1 - 2^-53 decides between q2 = 0 and q2 = 2e-16 ls^2, i.e. 1e-8 of the reflectance at any zenith
// (gortt_brdf.c:650-666): the reference's value follows the last bit of glibc's sin and cos, which the library's
// reproduce (both are correctly rounded almost always) and a 1-ulp kernel does not - 24 reference hot-spot rows of
// tests/golden/fuzz_canopies.npz moved by up to 5e-8 with it.  Two calls per line, ~80 instructions more than the kernels.
// reflectances_only = true: nothing but the reflectance leaves the caller (LUT kernels, the albedo quadrature, a stream
// without -prnprop / -prnspec).  Then a line with a zenith of EXACTLY 90 degrees - the cosine of the double nearest pi/2
// is 6e-17; "exactly" is |cos| < 1e-15 - need not walk the reference's route: its reflectance is NaN by either arithmetic,
// because EPgap is 0 there and Kuusk's term 0 x inf (gortt_brdf.c:638-702; asserted for every node of the hemisphere grid
// in tests/test_gpu_parity.py and for six kinds of canopy in tests/test_stream_forms.py).  Not so for a Q08 canopy, whose
// closed-form gap probabilities leave the reference finite numbers on the horizon (-q08_pn_kopen: those take the route
// whatever is asked for), nor for lines merely NEAR the horizon.  What it buys: the route's library calls cost the whole
// wave of such a lane 10 us - hemisphere grids hold both horizons in every launch and waited for the few workgroups that
// walked it (C3 69 -> 55 us), the 90-degree sun of BASELINE config 4 took twice as long as the other 90 lines, and config 2
// is the principal plane from -90 to 90 degrees: two of its three waves held one such lane each (15 us against 5).
__device__ __forceinline__ bool takes_reference_route(const gort_canopy &c, double cos_vz, double cos_sz, bool reflectances_only)
{
    if (!near_horizon(cos_vz, cos_sz)) return false;
    const bool exactly = fabs(cos_vz) < 1e-15 || fabs(cos_sz) < 1e-15;
    return !(reflectances_only && exactly && !c.use_q08);
}

__device__ void row_terms(const gort_canopy &c, double vza, double sza, RowTerms &r, bool reflectances_only = false)
{
    sincos(vza, &r.sin_vz, &r.cos_vz);
    sincos(sza, &r.sin_sz, &r.cos_sz);
    r.horizon = takes_reference_route(c, r.cos_vz, r.cos_sz, reflectances_only) ? 1 : 0;
    if (__builtin_expect(r.horizon, 0)) row_terms_with<LibMath>(c, vza, sza, r);
    else row_terms_with<FastMath>(c, vza, sza, r);
}

// The azimuth-dependent rest: overlap and Kg at the actual azimuth, the interpolated Kc, the other
// proportions (gortt.c:424-449) and the hot spot.
template <class M>
__device__ void finish_angle_with(const gort_canopy &c, const RowTerms &r, double raa, GeomOut &o)
{
#pragma clang fp contract(off)
    const Primed &v = r.v, &s = r.s;
    double sin_r, cos_r;
    M::sincos(raa, sin_r, cos_r);
    const double O_r = overlap<M, false>(r.hb, s, v, cos_r, sin_r);
    const double Kg = M::exp(-(r.cov * (r.t1 - O_r)));
    const double ph_r = v.c * s.c + v.s * s.s * cos_r;
    const double F_r = M::div(r.Gv * 0.5 * (1.0 + ph_r), PI * c.rr * (r.t1 - O_r));

    double frac = M::over_pi(raa);
    if (frac > 1.0) frac = 2.0 - frac;
    double f = (1. - frac) * r.fF0 + frac * r.fFpi;
    f = r.beta * f + (1.0 - r.beta) * F_r;
    const double Kc = f * (1.0 - Kg);

    const double Kz = r.ev - Kg;                             // gortt.c:439
    const double Kt = ref_max(0.0, 1.0 - Kc - Kz - Kg);      // gortt.c:443-444
    const double Kpg = r.es - Kg;                            // gortt.c:448
    const double Kpz = 1.0 - r.ev - Kpg;                     // gortt.c:449

    // Kuusk's hot spot (unprimed angles in cos xi).  In the exact hot-spot direction (vza = sza, raa = 0)
    // q2 is pure rounding noise of cos_xi around 1, and exp(kf*h1*h2) amplifies it (up to ~1e-4 relative at
    // 89 deg): the reference's value there is decided by the last bit of its own libm.  The operations below
    // are kept unfused and in the reference's order (gortt_brdf.c:650-666) so that the same noise comes out
    // whenever the device sin/cos agree with glibc's.
    double h2 = 1.0;
    {
        const double cos_xi = r.cos_sz * r.cos_vz + r.sin_sz * r.sin_vz * cos_r;
        const double q2 = r.ls * r.ls + r.lv * r.lv - 2. * r.ls * r.lv * cos_xi;
        if (q2 > 0.0) {
            const double x = M::div(M::sqrt(q2), c.r);
            h2 = M::div(1.0 - M::exp(-x), x);
        }
    }
    const double kuusk = r.eps_s * r.eps_v * M::exp(r.kf * r.h1 * h2);

    o.Kc = Kc;  o.Kg = Kg;  o.Kt = Kt;  o.Kz = Kz;  o.Kpg = Kpg;  o.Kpz = Kpz;
    o.A = M::div(kuusk, 2.0 * s.c * v.c);
    o.sun = r.sun;
}

__device__ void finish_angle(const gort_canopy &c, const RowTerms &r, double raa, GeomOut &o)
{
    if (__builtin_expect(r.horizon, 0)) finish_angle_with<LibMath>(c, r, raa, o);
    else finish_angle_with<FastMath>(c, r, raa, o);
}

// areal proportions + hot spot for one normalised geometry
__device__ void geometry_core(const gort_canopy &c, double vza, double sza, double raa, GeomOut &o,
                              bool reflectances_only = false)
{
    RowTerms r;
    row_terms(c, vza, sza, r, reflectances_only);
    finish_angle(c, r, raa, o);
}

__device__ inline void store_coef(double *rec, const gort_canopy &c, const GeomOut &g)
{
#pragma clang fp contract(off)
    const double fd = g.sun.fd, kep = c.k_openep;
    rec[C_FDA] = fd * g.A;
    rec[C_KPZ] = fd * kep * g.Kpz;
    rec[C_KPG] = fd * kep * g.Kpg;
    rec[A_C] = g.Kc;
    rec[A_B] = g.Kc * rec[C_FDA];
    rec[A_Z] = g.Kc * rec[C_KPZ] + g.Kz;
    rec[A_G] = g.Kc * rec[C_KPG] + g.Kg;
    rec[A_T] = g.Kt;
    rec[S_FD] = fd;  rec[S_MU] = g.sun.mu;  rec[S_T0] = g.sun.t0;  rec[S_TP0] = g.sun.tp0;
    rec[S_EPS] = g.sun.eps;  rec[S_PN0] = g.sun.pn0;
    rec[C_PAD0] = 0.0;  rec[C_PAD1] = 0.0;
}

}  // namespace
}  // namespace gort
#endif
