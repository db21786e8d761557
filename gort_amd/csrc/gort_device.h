// gort_device.h -- device-side building blocks shared by the HIP translation units:
// angle conventions, gap-table lookup, the (sun zenith, band) two-stream terms, the 5-term expansion and the
// workgroup -> XCD-range mapping of the flat kernels.  Every kernel form calls THESE functions, so that all of them
// write the same bits for the same inputs (tests: stream == grid, grouped == per-line).
#ifndef GORT_DEVICE_H
#define GORT_DEVICE_H

#include <hip/hip_runtime.h>

#include "gort_internal.h"
#include "gort_math.h"

namespace gort {
namespace {

constexpr double PI = 3.14159265358979323846;
constexpr double INV_PI = 0.318309886183790671538;   // M_1_PI

// reference MAX/MIN macros (gortt.h:9-10): a NaN in the second slot survives
__device__ inline double ref_max(double x, double y) { return x > y ? x : y; }
__device__ inline double ref_min(double x, double y) { return x < y ? x : y; }

struct SunScalars { double fd, mu, t0, tp0, eps, pn0; };

// rsurf = aC*C0 + aB*B + aZ*Z + aG*G + aT*T with ONE fixed association (an explicit FMA chain), so that
// every kernel form and every template instantiation writes the same bits for the same inputs.
__device__ __forceinline__ double dot5(double aC, double aB, double aZ, double aG, double aT,
                                       double C0, double B, double Z, double G, double T)
{
    return __builtin_fma(aT, T, __builtin_fma(aG, G, __builtin_fma(aZ, Z, __builtin_fma(aB, B, aC * C0))));
}

// ------------------------------------------------------------------ geometry

// linear interpolation in the 1-degree gap tables (gortt.c:872-915).  The reference
// indexes past the table for zenith > 90 deg; defined here as NaN.
__device__ inline void gap_lookup(const gort_canopy &c, double za, double &pn0, double &epg)
{
#pragma clang fp contract(off)
    const double pos = fabs(za) / c.dth;
    const double cf = ceil(pos), ff = floor(pos);
    if (!(cf <= (double)(GORT_NTH - 1))) { pn0 = epg = __builtin_nan(""); return; }
    const int ci = (int)cf, fi = (int)ff;
    const double d = pos - ff;
    pn0 = d * c.p_n0[ci] + (1.0 - d) * c.p_n0[fi];
    epg = d * c.epgap[ci] + (1.0 - d) * c.epgap[fi];
}

// sign / azimuth conventions of main() (gortt.c:240-279); degrees in, radians out
__device__ inline void normalise_angles(double vza_deg, double vaa_deg, double sza_deg, double saa_deg,
                                        double &vza, double &sza, double &saa, double &raa)
{
#pragma clang fp contract(off)
    // x * M_PI / 180.0 (gortt.c:240-243): the product, then the correctly rounded quotient by the constant
    vza = gm::div_by_constant(vza_deg * PI, 180.0, 1.0 / 180.0);
    double vaa = gm::div_by_constant(vaa_deg * PI, 180.0, 1.0 / 180.0);
    sza = gm::div_by_constant(sza_deg * PI, 180.0, 1.0 / 180.0);
    saa = gm::div_by_constant(saa_deg * PI, 180.0, 1.0 / 180.0);
    if (sza < 0.0) { saa += PI; sza *= -1.0; }      // normalised_sza() below restates these two lines
    if (vza < 0.0) { vaa += PI; vza *= -1.0; }
    // the reference wraps by repeated subtraction; beyond +-64 turns fold first so that
    // absurd inputs cannot stall a wavefront (documented deviation)
    if (fabs(saa) > 128 * PI) saa = fmod(saa, 2 * PI);
    if (fabs(vaa) > 128 * PI) vaa = fmod(vaa, 2 * PI);
    while (saa > 2 * PI) saa -= 2 * PI;
    while (vaa > 2 * PI) vaa -= 2 * PI;
    while (saa < 0) saa += 2 * PI;
    while (vaa < 0) vaa += 2 * PI;
    raa = saa - vaa;
    raa = fabs((raa - 2 * PI * (int)(0.5 + raa * INV_PI * 0.5)));    // C truncation toward zero
}

// the sun zenith in radians exactly as normalise_angles() leaves it: the key under which the stream path groups lines
__device__ inline double normalised_sza(double sza_deg)
{
#pragma clang fp contract(off)
    double sza = gm::div_by_constant(sza_deg * PI, 180.0, 1.0 / 180.0);
    if (sza < 0.0) sza *= -1.0;
    return sza;
}

struct Primed { double ang, s, c, sec, t; };     // theta' = atan((b/r) tan theta)  (gortt.c:581-588)

// sine and cosine of an azimuth difference in radians: main() folds it into [0, 2 pi] (gortt.c:266-279), so one
// reduction step and the kernels of gort_math.h do; the fold below only guards the function's contract
__device__ inline void sincos_of(double x, double &s, double &c)
{
    if (__builtin_expect(fabs(x) > gm::SINCOS_MAX, 0)) x = fmod(x, 2 * PI);      // inf -> NaN
    gm::sincos_reduced(x, s, c);
}

// ---- the two arithmetics of the geometry stage --------------------------------------------------------------------
// FastMath: the bounded-range kernels of gort_math.h (within ~1 ulp, divisions by reciprocal + Newton) and the primed
// angle's cosine, secant and tangent straight from x = (b/r) tan theta (cos = 1/sqrt(1 + x^2), tan = x) instead of
// through atan and back.  What every line takes - except the ones below.
// LibMath: the reference's own route step by step - atan, then sin / cos / 1/cos / sin/cos OF THE ROUNDED ANGLE, IEEE
// division and square root, the device library's (nearly correctly rounded) functions.  Within ~1e-6 rad of a zenith of
// 90 degrees the reference's numbers are made of exactly that rounding: tan(90 deg) = 1.6e16 and atan((b/r) 1.6e16)
// rounds to the double nearest pi/2, whose cosine is 6.1e-17 whatever b/r is; 1 - M of gortt_brdf.c:205 is then ONE ulp
// of 1, and Kc, printed by -prnprop, is a quotient of two such numbers (0.901419 at vza = 90 deg for the default canopy).
// Lines that close to the horizon evaluate the reference's route so that they print what the reference prints; they are
// rare, the branch is wave-uniform almost always, and its code is round 3's (which the goldens pinned).
struct FastMath {
    static __device__ __forceinline__ double div(double a, double b) { return gm::quot_finite(a, b); }     // b finite, not 0
    // div(a, b) with recip(b) formed ahead (a denominator shared by many quotients): the same bits as div(a, b)
    static __device__ __forceinline__ double recip(double b) { return gm::recip(b); }
    static __device__ __forceinline__ double div_with(double a, double b, double rb) { return gm::quot_finite_with(a, b, rb); }
    static __device__ __forceinline__ double div_ieee(double a, double b) { return gm::quot(a, b); }       // b = 0, inf as IEEE
    static __device__ __forceinline__ double sqrt(double x) { return gm::sqrt_(x); }
    static __device__ __forceinline__ double acos_unit(double x) { return gm::acos_unit(x); }
    // sqrt(D D + x x) on the principal plane: x = tan tan sin(phi) is 0 (phi = 0) or 1e-16 of it (phi = M_PI), x x vanishes
    // beside D D, and the correctly rounded root of a rounded square is the number itself
    static __device__ __forceinline__ double hypot_principal(double D, double) { return D; }
    static __device__ __forceinline__ double exp(double x) { return gm::exp_(x); }
    static __device__ __forceinline__ double log(double x) { return gm::log_(x); }
    static __device__ __forceinline__ double acos(double x) { return gm::acos_(x); }
    static __device__ __forceinline__ double cos(double x) { return gm::cos_reduced(x); }
    static __device__ __forceinline__ void sincos(double x, double &s, double &c) { sincos_of(x, s, c); }
    static __device__ __forceinline__ double over_pi(double x) { return x * INV_PI; }
    // sin(acos(c)) = sqrt((1 - c)(1 + c)): both factors exact or nearly so
    static __device__ __forceinline__ double sin_of_acos(double, double c) { return gm::sqrt_((1.0 - c) * (1.0 + c)); }
    // cos(vza' cphi - sza') for cphi = +-1 is cos vza' cos sza' + cphi sin vza' sin sza', which the caller has: ph
    static __device__ __forceinline__ double cos_of_difference(const Primed &, const Primed &, double, double ph) { return ph; }
    // 1 / tan(a / 2) = (1 + sec a) / tan a
    static __device__ __forceinline__ double cot_half(const Primed &p) { return gm::quot(1.0 + p.sec, p.t); }
    static __device__ __forceinline__ Primed prime(double ell, double tan_za)
    {
#pragma clang fp contract(off)
        Primed p;
        const double x = ell * tan_za;
        p.ang = gm::atan_(x);
        gm::root_and_inverse(1.0 + x * x, p.sec, p.c);
        p.s = x * p.c;
        p.t = x;           // equal zeniths give bitwise equal primed tangents either way: the exact zero of overlap()'s distance
        return p;
    }
};
// (The library's functions are CALLED from that route, not inlined: beside the kernels their code - huge-argument
// reduction and all - costs every line 40 to 70 registers of allocation, i.e. a wave or two of occupancy per SIMD.)
#ifndef GORT_LIB_CALL
#define GORT_LIB_CALL __attribute__((noinline))
#endif
namespace lib {
__device__ GORT_LIB_CALL double exp_call(double x) { return ::exp(x); }
__device__ GORT_LIB_CALL double log_call(double x) { return ::log(x); }
__device__ GORT_LIB_CALL double acos_call(double x) { return ::acos(x); }
__device__ GORT_LIB_CALL double atan_call(double x) { return ::atan(x); }
__device__ GORT_LIB_CALL double tan_call(double x) { return ::tan(x); }
__device__ GORT_LIB_CALL double sin_call(double x) { return ::sin(x); }
__device__ GORT_LIB_CALL double cos_call(double x) { return ::cos(x); }
}  // namespace lib
struct LibMath {
    static __device__ __forceinline__ double div(double a, double b) { return a / b; }
    static __device__ __forceinline__ double recip(double) { return 0.0; }                       // (IEEE division has no use for it)
    static __device__ __forceinline__ double div_with(double a, double b, double) { return a / b; }
    static __device__ __forceinline__ double div_ieee(double a, double b) { return a / b; }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double acos_unit(double x) { return lib::acos_call(x); }
    static __device__ __forceinline__ double hypot_principal(double D, double x) { return ::sqrt(D * D + x * x); }
    static __device__ __forceinline__ double exp(double x) { return lib::exp_call(x); }
    static __device__ __forceinline__ double log(double x) { return lib::log_call(x); }
    static __device__ __forceinline__ double acos(double x) { return lib::acos_call(x); }
    static __device__ __forceinline__ double cos(double x) { return lib::cos_call(x); }
    // (the library's sincos(x) returns exactly its sin(x) and cos(x): one reduction, the same two kernels)
    static __device__ __forceinline__ void sincos(double x, double &s, double &c) { s = lib::sin_call(x);  c = lib::cos_call(x); }
    static __device__ __forceinline__ double over_pi(double x) { return x / PI; }
    static __device__ __forceinline__ double sin_of_acos(double t, double) { return lib::sin_call(t); }
    static __device__ __forceinline__ double cos_of_difference(const Primed &v, const Primed &s, double cphi, double)
    {
        return lib::cos_call(v.ang * cphi - s.ang);
    }
    static __device__ __forceinline__ double cot_half(const Primed &p) { return 1.0 / lib::tan_call(p.ang / 2.0); }
    static __device__ __forceinline__ Primed prime(double ell, double tan_za)
    {
        Primed p;
        p.ang = lib::atan_call(ell * tan_za);
        sincos(p.ang, p.s, p.c);
        p.sec = 1.0 / p.c;
        p.t = p.s / p.c;
        return p;
    }
};
// does a line with these zenith cosines take the reference's route (a zenith within 1e-6 rad of 90 degrees)?
constexpr double HORIZON_COS = 1e-6;
__device__ __forceinline__ bool near_horizon(double cos_vz, double cos_sz)
{
    return fabs(cos_vz) < HORIZON_COS || fabs(cos_sz) < HORIZON_COS;
}

// the scalars of a line that depend on its sun zenith only (gortt.c:290-291, 872-915; gortt_brdf.c:447,534)
template <class M>
__device__ inline SunScalars sun_scalars(const gort_canopy &c, double sza, double cos_sz, const Primed &sp)
{
#pragma clang fp contract(off)
    SunScalars s;
    gap_lookup(c, sza, s.pn0, s.eps);
    s.fd = c.use_user_fd ? c.fd_user : M::div(cos_sz, cos_sz + 0.09);   // Ni et al. '99, gortt.c:290-291
    s.mu = sp.c;
    s.t0 = M::exp(-(c.k * c.elai * sp.sec));
    s.tp0 = s.pn0 + s.eps;
    return s;
}

// the same from the normalised zenith alone, as row_terms() derives them for a line with that sun zenith
__device__ inline SunScalars sun_from_zenith(const gort_canopy &c, double sza)
{
    double sin_sz, cos_sz;
    sincos(sza, &sin_sz, &cos_sz);                 // the zeniths' own sine and cosine: the library's (gort_geometry.h)
    if (near_horizon(1.0, cos_sz)) return sun_scalars<LibMath>(c, sza, cos_sz, LibMath::prime(c.ell, sin_sz / cos_sz));
    return sun_scalars<FastMath>(c, sza, cos_sz, FastMath::prime(c.ell, FastMath::div(sin_sz, cos_sz)));
}

// t'_ff = t_ff (1 - kopen) + kopen with kopen = k_open + k_openep (gortt_brdf.c:348-365): ONE fused operation in
// lambda_table_kernel (what the compiler made of it since round 1) and wherever else it is re-derived from t_ff
__device__ __forceinline__ double tpff_of(double tff, double kopen) { return __builtin_fma(tff, 1.0 - kopen, kopen); }

struct SunTerms { double C0, B, Z, G, T; };
struct BandTerms { double gam, omega, Rff, Tff, tff, pff, rs, mgk, Zf, Tf, B; };

__device__ inline BandTerms load_band(const double *__restrict__ L, int nw, int i)
{
    BandTerms t;
    t.gam = L[L_GAMMA * nw + i];  t.omega = L[L_OMEGA * nw + i];
    t.Rff = L[L_RFF * nw + i];    t.Tff = L[L_TFF * nw + i];
    t.tff = L[L_tFF * nw + i];    t.pff = L[L_PFF * nw + i];
    t.rs = L[L_RS * nw + i];      t.mgk = L[L_MGK * nw + i];
    t.Zf = L[L_ZF * nw + i];      t.Tf = L[L_TF * nw + i];
    t.B = L[L_B * nw + i];
    return t;
}

// the five (sun zenith, band) numbers.  The two quotients of the reference, 1/(1+2 mu gamma) and
// 1/(1-(2 gamma mu)^2), share ONE fp64 division: 1-(g2)^2 = (1+g2)(1-g2).
__device__ inline SunTerms sun_terms(const BandTerms &t, const SunScalars &s, double ko, double kep)
{
    const double mu = s.mu, fd = s.fd;
    const double g2 = 2. * t.gam * mu;
    const double inv = 1.0 / ((1.0 + g2) * (1.0 - g2));
    const double Rdf = (1.0 - t.gam) * ((1.0 - g2) * inv);                      // (1-gamma)/(1+2 mu gamma), gortt_brdf.c:552
    const double Tdf = (t.omega / 2.0) * ((1. + 2. * mu) * inv) * (t.Tff - s.t0);   // :467-471
    const double X = s.t0 * Rdf + Tdf * t.Rff;
    const double tdf = Tdf - t.pff * X;                                         // :423-424
    const double pdf = Rdf - t.tff * X;                                         // :628-630
    const double tpdf = tdf * (1 - s.tp0);                                      // :361
    SunTerms o;
    o.B = t.B;
    o.G = fd * t.rs + (1 - fd) * t.rs;                                          // gortt.c:481-484
    o.Z = fd * ((tpdf + s.eps) * t.rs) + (1 - fd) * t.Zf;                       // gortt.c:491-494
    const double Td = (tpdf + s.tp0) * t.mgk;                                   // gortt.c:541-543
    o.T = fd * Td + (1 - fd) * t.Tf;                                            // gortt.c:550
    const double kk = kep + ko;
    const double CfG = (kk * o.G + (1 - kk) * o.Z) * kep;                       // gortt.c:516-517
    o.C0 = fd * (pdf + Td) + (1 - fd) * (t.pff + CfG + t.Tf);
    return o;
}

__device__ inline SunTerms sun_terms(const double *__restrict__ L, int nw, int i, const SunScalars &s,
                                     double ko, double kep)
{
    return sun_terms(load_band(L, nw, i), s, ko, kep);
}

// ---- the STREAM family's form of a sample ---------------------------------------------------------------------
// rsurf is linear in everything but the two numbers R_df and T_df (gortt_brdf.c:552, 467-471), the only ones in which
// sun zenith and band meet non-linearly, and those two share one denominator.  Regrouping gortt.c:484-557 and
// gortt_brdf.c:348-365, 616-634 (p_df = R_df - t_ff X, t'_df = (T_df - p_ff X)(1 - t'_0), X = R_ff T_df + t_0 R_df)
// around them,
//
//   rsurf = [ n1 (alpha - t0 S) + n2 (W - R_ff S) ] / (1 - x^2) + (Q1 mgk + Q2 rs + Q3 Zf + Q4 Tf + Q5 p_ff + Q6 B)
//   x = 2 gamma mu,  n1 = (1-gamma)(1-x),  n2 = (1+2mu) (omega/2) (T_ff - t0),  W = P1 mgk + P2 rs,  S = alpha t_ff + W p_ff
//
// Round 6: the factor m = 1 + 2 mu of n2 is a number of the LINE, so it rides on the line's weights instead of being
// multiplied into every sample - W_m = m W = (m P1) mgk + (m P2) rs, S_m = m S = (m alpha) t_ff + W_m p_ff,
// n2 (W - R_ff S) = n2' (W_m - R_ff S_m) with n2' = (omega/2)(T_ff - t0), alpha - t0 S = alpha - (t0 / m) S_m - and
// n1 = c1 - c1 x is one fused operation: THIRTEEN scalars per line (alpha, m alpha, m P1, m P2, Q1 .. Q6, mu, t0, t0 / m:
// line_terms) and twelve band constants, 22 instructions + one reciprocal (26 issue slots) per sample (rounds 3-5: 24 + 1;
// sun_terms() + dot5() take ~54; the round-2 form of this regrouping, p_df and t'_df formed explicitly, took 32): the
// stream kernels are bound by fp64 issue or close to it, and an instruction less per sample is an instruction less per sample.
// Every kernel that expands a STREAM (flat panels, band-major, per sample, fused with the geometry) evaluates exactly
// these functions, written with explicit FMAs and without contraction so that all of them produce the same bits;
// the LUT family keeps the five (sun zenith, band) terms and dot5().  The two families differ by rounding only
// (a few 1e-16 relative; both are held to 1e-9 against the reference).
struct LineTerms { double alpha, am, P1m, P2m, Q1, Q2, Q3, Q4, Q5, Q6, mu, t0, t0m; };
constexpr int LINE_NTERMS = 13;                // <= GORT_COEF_STRIDE: a LineTerms record is what layout 1 of the stream records holds

__device__ inline LineTerms line_terms(double aC, double aB, double aZ, double aG, double aT, double fd, double mu,
                                       double t0, double tp0, double eps, double kep, double kk)
{
#pragma clang fp contract(off)
    LineTerms l;
    const double omfd = 1.0 - fd;
    const double cfk = (aC * omfd) * kep;                 // aC (1-fd) k_openep
    const double Zc = __builtin_fma(cfk, 1.0 - kk, aZ);   // what multiplies Z in the end
    const double sCT = aC + aT;
    const double p1 = fd * sCT, p2 = Zc * fd;             // the weights of t'_df: mgk and rs
    const double omtp0 = 1.0 - tp0;
    const double m = 1.0 + 2.0 * mu;                      // 1 ... 3
    l.alpha = aC * fd;
    l.am = l.alpha * m;
    l.P1m = (p1 * omtp0) * m;                             // (1 - t'_0) of gortt_brdf.c:361 folded into the line
    l.P2m = (p2 * omtp0) * m;
    l.Q1 = tp0 * p1;
    l.Q2 = __builtin_fma(p2, eps, __builtin_fma(cfk, kk, aG));
    l.Q3 = Zc * omfd;
    l.Q4 = omfd * sCT;
    l.Q5 = aC * omfd;
    l.Q6 = aB;
    l.mu = mu;
    l.t0 = t0;
    l.t0m = gm::quot_finite(t0, m);
    return l;
}

// 96 B, in this order in the band table of the flat stream kernel (stream_band_table(): six 16-B loads per band)
struct alignas(16) StreamBand { double g2, c1, c2, Rff, cT, tff, pff, rs, mgk, Zf, Tf, B; };
constexpr int STREAM_BAND_DOUBLES = 12;
static_assert(sizeof(StreamBand) == STREAM_BAND_DOUBLES * sizeof(double), "StreamBand is the band table's record");

__device__ inline StreamBand stream_band(const BandTerms &t)
{
#pragma clang fp contract(off)
    StreamBand b;
    b.g2 = 2.0 * t.gam;
    b.c1 = 1.0 - t.gam;
    b.c2 = t.omega / 2.0;
    b.cT = b.c2 * t.Tff;
    b.Rff = t.Rff;  b.tff = t.tff;  b.pff = t.pff;
    b.rs = t.rs;    b.mgk = t.mgk;  b.Zf = t.Zf;    b.Tf = t.Tf;   b.B = t.B;
    return b;
}

// 1 / den for the shared denominator 1 - x^2 of R_df = (1-gamma)/(1+x) and T_df (gortt_brdf.c:552, 467-471)
__device__ __forceinline__ double stream_reciprocal(double den)
{
#pragma clang fp contract(off)
#ifdef GORT_IEEE_DIV
    return 1.0 / den;
#else
    // v_rcp_f64 + ONE third-order step (e = 1 - den r;  r (1 + e + e^2): 3 FMAs, error e^3) instead of the correctly
    // rounded division sequence (~14 issue slots) or two Newton steps (4 FMAs): an ulp or two off - the same function
    // in every stream kernel, so the same bits everywhere; x = 1 (2 gamma mu = 1) is singular either way
    const double r = __builtin_amdgcn_rcp(den);
    const double e = __builtin_fma(-den, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
#endif
}

__device__ __forceinline__ double stream_sample(double alpha, double am, double P1m, double P2m, double Q1, double Q2,
                                                double Q3, double Q4, double Q5, double Q6, double mu, double t0, double t0m,
                                                const StreamBand &b)
{
#pragma clang fp contract(off)
    const double x = b.g2 * mu;
    const double inv = stream_reciprocal(__builtin_fma(-x, x, 1.0));
    const double n1 = __builtin_fma(-b.c1, x, b.c1);
    const double n2 = __builtin_fma(-b.c2, t0, b.cT);           // without its factor 1 + 2 mu: that one is in W, S
    const double W = __builtin_fma(P2m, b.rs, P1m * b.mgk);
    const double S = __builtin_fma(W, b.pff, am * b.tff);
    const double A = __builtin_fma(-t0m, S, alpha);
    const double Bc = __builtin_fma(-b.Rff, S, W);
    const double num = __builtin_fma(n1, A, n2 * Bc);
    const double lin = __builtin_fma(Q6, b.B, __builtin_fma(Q5, b.pff, __builtin_fma(Q4, b.Tf, __builtin_fma(Q3, b.Zf,
                       __builtin_fma(Q2, b.rs, Q1 * b.mgk)))));
    return __builtin_fma(inv, num, lin);
}

__device__ __forceinline__ double stream_sample(const LineTerms &l, const StreamBand &b)
{
    return stream_sample(l.alpha, l.am, l.P1m, l.P2m, l.Q1, l.Q2, l.Q3, l.Q4, l.Q5, l.Q6, l.mu, l.t0, l.t0m, b);
}

// the line terms of a classic record (narrow kernels derive them on the fly; the wide ones read them precomputed)
__device__ inline LineTerms line_terms_of_record(const double *__restrict__ rec, double kep, double ko)
{
    return line_terms(rec[A_C], rec[A_B], rec[A_Z], rec[A_G], rec[A_T], rec[S_FD], rec[S_MU], rec[S_T0], rec[S_TP0],
                      rec[S_EPS], kep, kep + ko);
}

__device__ inline SunScalars load_sun(const double *__restrict__ rec)
{
    SunScalars s;
    s.fd = rec[S_FD];  s.mu = rec[S_MU];  s.t0 = rec[S_T0];  s.tp0 = rec[S_TP0];
    s.eps = rec[S_EPS];  s.pn0 = rec[S_PN0];
    return s;
}

// ---- which logical block (= 4 consecutive waves of the flat kernels) does this workgroup work on ----
// Any bijection is correct; only speed depends on it.  Each XCD gets ONE contiguous range of logical blocks
// (eight compact write windows, one per L2, instead of one window interleaved over all eight).
//   mode 0  identity (interleaved)
//   mode 1  static: workgroups b, b+8, ... run on one XCD each (round-robin dispatch, probed per engine).
//           The XCDs do not write equally fast - on the parts measured the XCDs of one parity (the odd XCC_IDs in
//           every standalone probe; the even dispatch slots in one process) sustain ~80 % of the others, and a launch ends with its slowest XCD - so XCD x uses only w[x] of every 32 of its
//           workgroups (the others return at once) and owns a range of logical blocks in proportion
//           (calibrate_xcd_weights; tools/probes/xcd_stream_probe.hip: 7.06 -> 6.70 ms for the 50 GB slab).
//   mode 2  dynamic: read the XCD the workgroup really runs on (HW_REG_XCC_ID) and take the next free slot of
//           that XCD's range with one returning atomic; if the range is used up take one from the next XCD.
//           The ranges sum to the grid, so every workgroup finds a slot within 8 tries.  The launcher zeroes
//           the counters (one per 128-B line) on the stream before every launch.
// Returns -1 for a workgroup without work.
// mode 1 for workgroup b (host-callable so that gort_selftest_index_math can check the bijection without a GPU)
__host__ __device__ __forceinline__ long duty_logical_block(long b, const XcdDuty &duty, long useful)
{
    const unsigned sh = ((unsigned)b & 7u) * 8u;
    const unsigned w = (unsigned)(duty.w8 >> sh) & 0xffu;
    // sum of the weights of the XCDs in front (<= 7 x 32, fits the top byte of the byte-wise product)
    const unsigned long long below = duty.w8 & ((1ull << sh) - 1ull);
    const unsigned pw = (unsigned)((below * 0x0101010101010101ull) >> 56);
    const long i = b >> 3;                               // < 32 q by the size of the grid
    const long li = (i * w) >> 5;                        // evenly spread: slot i works iff floor((i+1)w/32) > floor(iw/32)
    if ((((i + 1) * w) >> 5) == li) return -1;
    // (round 3 let XCD x start x/8 of the way into its range, so that the eight write streams never sit at the same
    // offset of their ranges: no effect on the comb of profiles/r03/placement_scan_rotate.log)
    const long block = duty.q * pw + li;
    return block < useful ? block : -1;
}

__device__ __forceinline__ long xcd_logical_block(int xcd_mode, const XcdDuty &duty, long useful,
                                                  int *__restrict__ xcd_slots)
{
    const long b = blockIdx.x;
    if (xcd_mode == 1) return duty_logical_block(b, duty, useful);
    if (xcd_mode == 2) {
        const long base = useful >> 3, rem = useful & 7;     // XCD y owns [y*base + min(y,rem), +base (+1 if y < rem))
        __shared__ long s_block;
        if (threadIdx.x == 0) {
            unsigned x;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
            long L = -1;
            for (int t = 0; t < 8 && L < 0; ++t) {
                const long y = (x + t) & 7;
                const long quota = base + (y < rem ? 1 : 0);
                const long s = atomicAdd(&xcd_slots[y * XCD_SLOT_PITCH], 1);
                if (s < quota) L = y * base + (y < rem ? y : rem) + s;
            }
            s_block = L;
        }
        __syncthreads();
        return s_block;                                       // never -1 (pigeonhole), checked by the caller anyway
    }
    return b < useful ? b : -1;
}

inline int check_launch(const char *what)
{
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(GORT_ENODEVICE, "%s: %s", what, hipGetErrorString(err));
    return GORT_OK;
}

constexpr int EPL = 2;                  // elements (adjacent bands) per lane and step
constexpr int CHUNK = 64 * EPL;         // doubles per wave-step
typedef double dbl2 __attribute__((ext_vector_type(2)));

// exact n / d for n < 2^31 and a divisor fixed per launch: (n * mul) >> (31 + sh), mul and sh from the host
// (make_fast_div); five scalar instructions instead of the ~35 of a 32-bit division with a run-time divisor
__host__ __device__ __forceinline__ unsigned fast_div(unsigned n, FastDiv d)
{
    return (unsigned)(((unsigned long long)n * d.mul) >> (31u + d.sh));
}
}  // namespace
}  // namespace gort
#endif
