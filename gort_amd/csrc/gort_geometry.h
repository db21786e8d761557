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
// What every overlap of one (view zenith, sun zenith) pair shares - the three of the row's terms and those of all its azimuth
// nodes: the products of the primed tangents as overlap() had them inline, t1 = s.sec + v.sec and M::recip(t1)
struct OverlapPair { double tt_sum, tt_2, tt_prod, t1, rcp_t1; };
template <class M>
__device__ __forceinline__ OverlapPair overlap_pair(const Primed &s, const Primed &v)
{
#pragma clang fp contract(off)
    OverlapPair p;
    p.tt_sum = s.t * s.t + v.t * v.t;
    p.tt_2 = 2.0 * s.t * v.t;
    p.tt_prod = s.t * v.t;
    p.t1 = s.sec + v.sec;
    p.rcp_t1 = M::recip(p.t1);
    return p;
}
template <class M, bool PRINCIPAL>
__device__ inline double overlap(double hb, const OverlapPair &p, double cphi, double sphi)
{
#pragma clang fp contract(off)
    const double d = p.tt_sum - p.tt_2 * cphi;            // s.t s.t + v.t v.t - 2.0 s.t v.t cphi
    const double D = M::sqrt(ref_max(0.0, d));
    const double x = p.tt_prod * sphi;
    const double t2 = PRINCIPAL ? M::hypot_principal(D, x) : M::sqrt(D * D + x * x);
    const double t1 = p.t1;
    double cos_t = M::div_with(hb * t2, t1, p.rcp_t1);
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
    // what a row's azimuth nodes would each form again (round 5): reciprocals of the denominators they share - t1 (overlap),
    // the crown radius (hot spot), 2 cos(sza') cos(vza') (A) - and the row-only factors of Kuusk's term.  Formed by the
    // expressions finish_angle_with() had inline, used through M::div_with(): the same bits.
    OverlapPair op;
    double rcp_r, den_A, rcp_den_A, q2_a, q2_b, kf_h1, eps_sv, half_Gv, cc, ss, czz, szz, one_m_beta, one_m_ev;
    SunScalars sun;
    int horizon;                           // the line takes the reference's route (LibMath)
};

// the row-only operands of finish_angle_with() (RowTerms: "what a row's azimuth nodes would each form again"); needs ls, lv, kf,
// h1, eps_s, eps_v, Gv and the primed angles
template <class M>
__device__ __forceinline__ void row_node_invariants(const gort_canopy &c, RowTerms &r)
{
#pragma clang fp contract(off)
    r.rcp_r = M::recip(c.r);
    r.den_A = 2.0 * r.s.c * r.v.c;
    r.rcp_den_A = M::recip(r.den_A);
    r.q2_a = r.ls * r.ls + r.lv * r.lv;
    r.q2_b = 2. * r.ls * r.lv;
    r.kf_h1 = r.kf * r.h1;
    r.eps_sv = r.eps_s * r.eps_v;
    r.half_Gv = r.Gv * 0.5;
    r.cc = r.v.c * r.s.c;
    r.ss = r.v.s * r.s.s;
    r.czz = r.cos_sz * r.cos_vz;
    r.szz = r.sin_sz * r.sin_vz;
    r.one_m_beta = 1.0 - r.beta;
    r.one_m_ev = 1.0 - r.ev;
}

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
    r.op = overlap_pair<M>(s, v);
    r.t1 = r.op.t1;
    const double cov = r.cov, t1 = r.t1;

    // principal-plane overlaps (Kc is interpolated between phi = 0 and pi, gortt_brdf.c:143-159)
    const double O_0 = overlap<M, true>(r.hb, r.op, 1.0, 0.0);
    const double O_pi = overlap<M, true>(r.hb, r.op, -1.0, 1.2246467991473532e-16);   // sin(M_PI) in double
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
    row_node_invariants<M>(c, r);
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

// ---- the same row terms, SPLIT over lanes (round 5) -------------------------------------------------------------------
// row_terms() is one chain of ~1100 dependent instructions per row, and a kernel that evaluates a handful of rows in front
// of its nodes (the grid kernels, the albedo quadrature) spends it on a dozen lanes of ONE wave while every other wave of
// the workgroup waits at the barrier: 2.5 us (hemisphere grid) to 4.5 us (albedo line) at the front of every workgroup with
// the SIMDs a quarter busy.  But the chain is three chains that do not need each other ("Mi, theta_Mi on theta_s;
// Mv, theta_Mv on theta_v"; gortt_brdf.c:118-238):
//   stage 1, a lane per (row, zenith): the zenith's sine and cosine, its gap-table lookup and its primed angle - the view
//            zenith's on one lane, the sun zenith's on its neighbour;
//   stage 2, the same lanes: exp(-cov sec), M, the path length of Kuusk's hot spot, the sun's theta_Mi - while the lanes of
//            ANOTHER wave (other issue slots), a lane per (row, principal-plane azimuth 0 | pi), form the overlap and Kg there,
//            which need the primed angles only;
//   stage 3, those lanes: f F on the principal plane - while a lane per row on the first wave forms the sun's fd and t0, beta,
//            and the hot spot's sqrt(ls lv).
// The longest path is ~450 instructions (~650 with two stages, round 5's first form).  Every number is formed by the expression row_terms_with() forms it by, from the
// same operands (contraction off in both): the same bits (tests/test_grid_forms.py, tests/test_energy_forms.py).
struct RowScratch { double Mv, Mi, theta_Mi, O[2], Kg[2]; };

// stage 1, thread `unit` of the workgroup's first 2 n_rows threads = (row unit >> 1, which = unit & 1: 0 view, 1 sun): the
// zenith's primed angle (and what needs nothing else of it).  (pn0, eps) = gap_lookup(c, za): handed in - the caller asks for
// them in front of the zenith's sine and cosine, so that the table's four loads are on their way while the library's sincos runs
template <class M>
__device__ void row_prime_with(const gort_canopy &c, bool is_sun, double sn, double cs, double pn0, double eps, RowTerms &r)
{
#pragma clang fp contract(off)
    const Primed p = M::prime(c.ell, M::div(sn, cs));
    if (is_sun) {
        r.s = p;  r.sin_sz = sn;  r.cos_sz = cs;
        r.sun.pn0 = pn0;  r.sun.eps = eps;  r.eps_s = eps;
    } else {
        r.v = p;  r.sin_vz = sn;  r.cos_vz = cs;
        r.Gv = PI * c.rr * p.sec;
        r.eps_v = eps;
    }
}

// stage 2 on the same threads: what else belongs to one zenith - exp(-cov sec), M, the hot spot's path length, the sun's theta_Mi
template <class M>
__device__ void row_unit_with(const gort_canopy &c, bool is_sun, RowTerms &r, RowScratch &x)
{
#pragma clang fp contract(off)
    const Primed p = is_sun ? r.s : r.v;
    const double eps = is_sun ? r.eps_s : r.eps_v;
    const double cov = c.lambda * PI * c.rr;
    const double xx = cov * p.sec;
    const double e = M::exp(-xx);
    const double Mm = 1.0 - M::div(1.0 - e, xx);
    const double l = M::div_ieee(-M::log(eps), is_sun ? c.k * c.favd : 0.5 * c.favd);
    if (is_sun) {
        r.es = e;
        r.kf = c.k * c.favd;
        r.ls = l;
        x.Mi = Mm;
        x.theta_Mi = M::acos(1.0 - 2.0 * Mm);
    } else {
        r.ev = e;
        r.lv = l;
        x.Mv = Mm;
    }
}

// stage 2 on the threads of another wave, lane (row, q): the overlap on the principal plane at phi = 0 (q = 0) or pi (q = 1)
// and Kg there - they need the two primed angles and nothing else
template <class M>
__device__ void row_overlap_with(const gort_canopy &c, int q, RowTerms &r, RowScratch &x)
{
#pragma clang fp contract(off)
    const Primed v = r.v, s = r.s;
    const double cov = c.lambda * PI * c.rr;
    const double hb = M::div(c.h, c.b);
    const OverlapPair op = overlap_pair<M>(s, v);
    const double t1 = op.t1;
    const double Oq = q ? overlap<M, true>(hb, op, -1.0, 1.2246467991473532e-16) : overlap<M, true>(hb, op, 1.0, 0.0);
    x.O[q] = Oq;
    x.Kg[q] = M::exp(-(cov * (t1 - Oq)));
    if (q == 0) { r.cov = cov;  r.hb = hb;  r.t1 = t1;  r.op = op; }
}

// stage 3 on those threads: f F on the principal plane from everything above
template <class M>
__device__ void row_plane_with(const gort_canopy &c, int q, double vza, double sza, RowTerms &r, const RowScratch &x)
{
#pragma clang fp contract(off)
    const Primed v = r.v, s = r.s;
    const double t1 = s.sec + v.sec;
    const double cphi = q ? -1.0 : 1.0;
    const double Oq = x.O[q], Kgq = x.Kg[q];
    const double Gv = r.Gv;
    const bool view_steeper = fabs(vza) > fabs(sza);
    const double ph = v.c * s.c + v.s * s.s * cphi;
    const double Gam = PI * c.rr * (t1 - Oq);
    const double Gc = Gv * 0.5 * (1.0 + ph);
    const double F = M::div(Gc, Gam);
    const double Mq = 1.0 - M::div(1.0 - Kgq, c.lambda * Gam);
    const double PiMi = (1 - M::cos(x.theta_Mi * (1 - M::over_pi(s.ang - v.ang * cphi)))) / 2.0;
    const double PvMv = x.Mv - (1.0 - M::cos_of_difference(v, s, cphi, ph)) / 2.0;
    const double Po = (q == 1) ? PvMv : (view_steeper ? PiMi : PvMv);
    const double f = M::div(F * (1.0 - M::div(Gv * (PvMv + PiMi - Po), Gc)), 1.0 - Mq);
    if (q) r.fFpi = f * F;
    else r.fF0 = f * F;
}

// stage 3, a lane per row on the first wave again: what is left of the sun scalars, beta, the hot spot's sqrt(ls lv)
template <class M>
__device__ void row_rest_with(const gort_canopy &c, RowTerms &r)
{
#pragma clang fp contract(off)
    const Primed s = r.s;
    r.sun.fd = c.use_user_fd ? c.fd_user : M::div(r.cos_sz, r.cos_sz + 0.09);
    r.sun.mu = s.c;
    r.sun.t0 = M::exp(-(c.k * c.elai * s.sec));
    r.sun.tp0 = r.sun.pn0 + r.sun.eps;
    if (c.use_user_beta) {
        r.beta = c.beta;
    } else if (s.ang < 0.000000001) {
        r.beta = 0.0;
    } else {
        const double Dd = c.r * M::cot_half(s);
        const double dh = M::div(c.h2 - c.h1, Dd);
        const double lg = c.lambda * r.Gv;
        r.beta = M::div(lg, lg + dh) * M::div(1.0 - M::exp(-lg - dh), 1.0 - M::exp(-lg));
    }
    r.h1 = (r.ls * r.lv) > 0.0 ? M::sqrt(r.ls * r.lv) : 0.0;
    row_node_invariants<M>(c, r);
}

// All threads of the workgroup call this (three barriers inside); rows[i], scr[i] in LDS for i < n_rows; args(i, c, vza, sza)
// names row i's canopy and its two normalised zeniths.  With U = 2 n_rows and P = U rounded up to whole waves:
//   stage 1  threads [0, U): primed angles;   stage 2  [0, U): the zeniths' other terms  |  [P, P + U): overlap and Kg;
//   stage 3  [P, P + U): f F                  |  [0, n_rows): the rest.       blockDim.x >= P + U.
template <class RowArgs>
__device__ __forceinline__ void row_terms_split(int n_rows, RowTerms *rows, RowScratch *scr, bool reflectances_only, RowArgs args)
{
    const int tid = threadIdx.x;
    const int U = 2 * n_rows, P = (U + 63) & ~63;
    const gort_canopy *c = nullptr;
    double vza = 0.0, sza = 0.0;
    int i = -1;
    if (tid < U) i = tid >> 1;
    else if (tid >= P && tid < P + U) i = (tid - P) >> 1;
    if (i >= 0) args(i, c, vza, sza);
    if (tid < U) {
        const int is_sun = tid & 1;
        const double za = is_sun ? sza : vza;
        double pn0, eps;
        gap_lookup(*c, za, pn0, eps);                          // first: its loads fly while the sine and cosine are formed
        double sn, cs;
        sincos(za, &sn, &cs);                                  // the zeniths' own sine and cosine: the library's (row_terms)
        const double other = __shfl_xor(cs, 1, 64);            // the row's other zenith sits on the neighbouring lane
        const bool horizon = takes_reference_route(*c, is_sun ? other : cs, is_sun ? cs : other, reflectances_only);
        if (!is_sun) rows[i].horizon = horizon ? 1 : 0;
        if (__builtin_expect(horizon, 0)) row_prime_with<LibMath>(*c, is_sun != 0, sn, cs, pn0, eps, rows[i]);
        else row_prime_with<FastMath>(*c, is_sun != 0, sn, cs, pn0, eps, rows[i]);
    }
    __syncthreads();
    if (tid < U) {
        if (__builtin_expect(rows[i].horizon, 0)) row_unit_with<LibMath>(*c, (tid & 1) != 0, rows[i], scr[i]);
        else row_unit_with<FastMath>(*c, (tid & 1) != 0, rows[i], scr[i]);
    } else if (i >= 0) {
        if (__builtin_expect(rows[i].horizon, 0)) row_overlap_with<LibMath>(*c, (tid - P) & 1, rows[i], scr[i]);
        else row_overlap_with<FastMath>(*c, (tid - P) & 1, rows[i], scr[i]);
    }
    __syncthreads();
    if (tid >= P && i >= 0) {
        if (__builtin_expect(rows[i].horizon, 0)) row_plane_with<LibMath>(*c, (tid - P) & 1, vza, sza, rows[i], scr[i]);
        else row_plane_with<FastMath>(*c, (tid - P) & 1, vza, sza, rows[i], scr[i]);
    } else if (tid < n_rows) {
        const gort_canopy *cr;
        double vz, sz;
        args(tid, cr, vz, sz);
        if (__builtin_expect(rows[tid].horizon, 0)) row_rest_with<LibMath>(*cr, rows[tid]);
        else row_rest_with<FastMath>(*cr, rows[tid]);
    }
    __syncthreads();
}

// The azimuth-dependent rest: overlap and Kg at the actual azimuth, the interpolated Kc, the other
// proportions (gortt.c:424-449) and the hot spot.
// (sin_r, cos_r) = M::sincos(raa): handed in, because a grid kernel has them in a table per azimuth node
template <class M>
__device__ void finish_angle_with(const gort_canopy &c, const RowTerms &r, double raa, double sin_r, double cos_r, GeomOut &o)
{
#pragma clang fp contract(off)
    const double O_r = overlap<M, false>(r.hb, r.op, cos_r, sin_r);
    const double Kg = M::exp(-(r.cov * (r.t1 - O_r)));
    const double ph_r = r.cc + r.ss * cos_r;               // v.c s.c + v.s s.s cos_r
    const double F_r = M::div(r.half_Gv * (1.0 + ph_r), PI * c.rr * (r.t1 - O_r));

    double frac = M::over_pi(raa);
    if (frac > 1.0) frac = 2.0 - frac;
    double f = (1. - frac) * r.fF0 + frac * r.fFpi;
    f = r.beta * f + r.one_m_beta * F_r;
    const double Kc = f * (1.0 - Kg);

    const double Kz = r.ev - Kg;                             // gortt.c:439
    const double Kt = ref_max(0.0, 1.0 - Kc - Kz - Kg);      // gortt.c:443-444
    const double Kpg = r.es - Kg;                            // gortt.c:448
    const double Kpz = r.one_m_ev - Kpg;                     // gortt.c:449: 1.0 - ev - K'g

    // Kuusk's hot spot (unprimed angles in cos xi).  In the exact hot-spot direction (vza = sza, raa = 0)
    // q2 is pure rounding noise of cos_xi around 1, and exp(kf*h1*h2) amplifies it (up to ~1e-4 relative at
    // 89 deg): the reference's value there is decided by the last bit of its own libm.  The operations below
    // are kept unfused and in the reference's order (gortt_brdf.c:650-666) so that the same noise comes out
    // whenever the device sin/cos agree with glibc's.
    double h2 = 1.0;
    {
        const double cos_xi = r.czz + r.szz * cos_r;           // cos_sz cos_vz + sin_sz sin_vz cos_r
        const double q2 = r.q2_a - r.q2_b * cos_xi;            // ls ls + lv lv - 2 ls lv cos xi, the row's parts formed ahead
        if (q2 > 0.0) {
            const double x = M::div_with(M::sqrt(q2), c.r, r.rcp_r);
            h2 = M::div(1.0 - M::exp(-x), x);
        }
    }
    const double kuusk = r.eps_sv * M::exp(r.kf_h1 * h2);

    o.Kc = Kc;  o.Kg = Kg;  o.Kt = Kt;  o.Kz = Kz;  o.Kpg = Kpg;  o.Kpz = Kpz;
    o.A = M::div_with(kuusk, r.den_A, r.rcp_den_A);
    o.sun = r.sun;
}

template <class M>
__device__ void finish_angle_with(const gort_canopy &c, const RowTerms &r, double raa, GeomOut &o)
{
    double sin_r, cos_r;
    M::sincos(raa, sin_r, cos_r);
    finish_angle_with<M>(c, r, raa, sin_r, cos_r, o);
}

__device__ void finish_angle(const gort_canopy &c, const RowTerms &r, double raa, GeomOut &o)
{
    if (__builtin_expect(r.horizon, 0)) finish_angle_with<LibMath>(c, r, raa, o);
    else finish_angle_with<FastMath>(c, r, raa, o);
}

// the same with the azimuth's FastMath sine and cosine from the caller's table (a row on the reference's route forms its own)
__device__ void finish_angle(const gort_canopy &c, const RowTerms &r, double raa, double sin_fast, double cos_fast, GeomOut &o)
{
    if (__builtin_expect(r.horizon, 0)) finish_angle_with<LibMath>(c, r, raa, o);
    else finish_angle_with<FastMath>(c, r, raa, sin_fast, cos_fast, o);
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
