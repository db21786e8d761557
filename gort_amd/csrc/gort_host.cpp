// gort_host.cpp -- host-side precompute of libgort_amd: canopy derivation, Price soil
// and PROSPECT-D leaf spectra, Gauss-Legendre nodes, the -W/-P probability LUT text.
//
// north_star keeps these on the host ("PROSPECT-D leaf and Price soil spectra
// precomputed on the host"); they run once per canopy / leaf parameter set.
// Behaviour follows the reference functions cited in include/gort_amd.h; the
// arithmetic association of each formula is kept so results agree with the reference
// to rounding (this file is compiled with -ffp-contract=off).
#include "gort_amd.h"
#include "gort_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#ifndef GORT_DATA_DIR
#error "compile with -DGORT_DATA_DIR=\"<repo>/gort_amd/data\""
#endif

// Coefficient tables (data, see tools/extract_spectral_tables.py):
//   PROSPECT-D v6.0 specific absorption coefficients + refractive index, 7 x 2101 binary32
//   Price (1990) soil EOFs, 4 x 421 binary64
__asm__(".section .rodata\n"
        ".balign 16\n"
        "gort_prospect_coeffs:\n"
        ".incbin \"" GORT_DATA_DIR "/prospect_d_coeffs.f32\"\n"
        ".balign 16\n"
        "gort_price_eofs:\n"
        ".incbin \"" GORT_DATA_DIR "/price_soil_eofs.f64\"\n"
        ".previous\n");
extern "C" const float  gort_prospect_coeffs[7 * GORT_NBANDS];
extern "C" const double gort_price_eofs[4 * 421];

namespace gort {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace gort

extern "C" const char *gort_last_error(void) { return gort::g_err; }
extern "C" const char *gort_version(void) { return "gort_amd 0.1 (gfx950)"; }

// ------------------------------------------------------------------------- canopy

extern "C" void gort_canopy_defaults(gort_canopy *c)
{
    std::memset(c, 0, sizeof *c);
    c->lambda = 0.405;
    c->r = 0.76;
    c->b = 3.55263 * c->r;
    c->h1 = 3.0;
    c->h2 = 8.5;
    c->favd = 0.858;
}

extern "C" void gort_leaf_soil_defaults(gort_leaf_soil *s)
{
    std::memset(s, 0, sizeof *s);
    s->N = 1.2;  s->Cab = 30.;  s->Car = 10.;  s->Anth = 1.0;
    s->Cbrown = 0.0;  s->Cw = 0.015;  s->Cm = 0.009;
    s->rsl[0] = 0.2;  s->rsl[1] = 0.1;  s->rsl[2] = 0.03726;  s->rsl[3] = -0.002426;
}

extern "C" void gort_canopy_newstyle(gort_canopy *c, float hb, float br, float pcc)
{
    c->r = 10.;
    c->b = br * c->r;
    c->h1 = c->b * 2.;
    c->h2 = hb * c->b + c->h1;
    c->lambda = pcc / (c->r * c->r * M_PI);
}

extern "C" void gort_canopy_set_lai(gort_canopy *c, float lai)
{
    c->favd = lai * 3. / (c->lambda * c->r * c->r * M_PI * c->b * 4.0);
}

extern "C" int gort_canopy_init(gort_canopy *c)
{
    if (!c) return gort::fail(GORT_EINVAL, "gort_canopy_init: null canopy");
    const int nl = GORT_NLAYERS;
    c->dth = 1 * M_PI / 180.0;
    c->ell = c->b / c->r;
    c->rr = c->r * c->r;
    c->rrr = c->rr * c->r;
    c->h = 2.0 * c->r * c->ell + c->h2 - c->h1;
    c->k = 0.5;                       // G-function fixed at 0.5 (LAD_05)
    c->elai = c->favd * ((1.333333) * c->lambda * M_PI * c->ell * c->rrr);   // sic: 1.333333
    c->tau = c->k * c->favd;
    c->z1 = c->h1 - c->r * c->ell;
    c->z2 = c->h2 + c->r * c->ell;
    c->lv = c->lambda / (c->h2 - c->h1);
    c->favd_p = c->favd * c->ell;
    c->tau_p = c->k * c->favd_p;
    c->lv_p = c->lv * c->ell;
    c->z1_p = c->z1 / c->ell;
    c->z2_p = c->z2 / c->ell;
    c->h1_p = c->h1 / c->ell;
    c->h2_p = c->h2 / c->ell;
    c->dz = (c->z2 - c->z1) / ((double)nl - 1.0);
    c->ds = c->dz;
    c->dz_p = c->dz / c->ell;
    const int nth = (int)((90.0 * M_PI / 180.0) / c->dth + 0.5) + 1;
    if (nth != GORT_NTH) return gort::fail(GORT_EINVAL, "gort_canopy_init: nth=%d", nth);
    for (int i = nl - 1; i >= 0; --i) {
        double height = c->z2 - c->dz * (double)(nl - 1 - i);
        c->height_p[i] = height / c->ell;
    }
    for (int i = 0; i < nth; ++i) {
        double th = c->dth * (double)i;
        if (th >= M_PI / 2.0) th = M_PI / 2.0 - 1.0 * M_PI / 180.0;     // horizon node clamps to 89 deg
        double thp = std::atan(std::tan(th) * c->ell);
        if (thp >= M_PI / 2.0) thp = M_PI / 2.0 - 1.0 * M_PI / 180.0;
        c->theta[i] = th;
        c->theta_p[i] = thp;
    }
    return GORT_OK;
}

// Degenerate crowns.  The reference has no check: a zero, negative, infinite or NaN radius ends in "Memory allocation
// failed (gortt_alloc_1d)" (the size of its path-length histogram overflows), h1 = h2 or -HB 0 in an endless loop (a
// midpoint rule whose step is 0), h2 < h1 in "Significant negative volume calculated in gortt_calc_vb" - all of them
// inside gortt_gap_probabilities, i.e. not when the tables come from a -P file.  Here they are one clean error at the
// same place: before the gap probabilities of such a crown would be computed.  (lambda and favd may be anything: the
// reference carries NaN through, and so do the kernels.)
extern "C" int gort_canopy_check_geometry(const gort_canopy *c)
{
    if (!c) return gort::fail(GORT_EINVAL, "gort_canopy_check_geometry: null canopy");
    const double must_be_positive[] = {c->r, c->b, c->h2 - c->h1, c->ell, c->dz, c->dz_p, c->h2_p - c->h1_p};
    for (double v : must_be_positive)
        if (!(v > 0.0) || !std::isfinite(v))
            return gort::fail(GORT_EINVAL, "invalid crown geometry (r=%g b=%g h1=%g h2=%g): radii and the height range of the "
                              "crown centres must be positive and finite", c->r, c->b, c->h1, c->h2);
    return GORT_OK;
}

// ------------------------------------------------------------------------ spectra

extern "C" int gort_price_soil(const double *wl, int nw, const double rsl[4], double *rsoil)
{
    const double *v1 = gort_price_eofs, *v2 = v1 + 421, *v3 = v2 + 421, *v4 = v3 + 421;
    for (int i = 0; i < nw; ++i) {
        // written so that a NaN is out of range too (the reference's `wl < 400 || wl > 2500`, gortt.c:1299, lets it
        // through and indexes the soil vectors with (int)NaN)
        if (!(wl[i] >= 400 && wl[i] <= 2500))
            return gort::fail(GORT_ERANGE, "gortt_price_soil: wavlength out of range (400-2500)");
        const int upper = (int)(1. + (wl[i] - 400) / 5.0);
        const int lower = (int)((wl[i] - 400) / 5.0);
        const double fraction = (wl[i] - 400.) / 5.0 - lower;
        const double lo = rsl[0] * v1[lower] + rsl[1] * v2[lower] + rsl[2] * v3[lower] + rsl[3] * v4[lower];
        // node 421 does not exist; it is only ever weighted by fraction == 0 (2500 nm)
        const double up = upper > 420 ? 0.0
            : rsl[0] * v1[upper] + rsl[1] * v2[upper] + rsl[2] * v3[upper] + rsl[3] * v4[upper];
        rsoil[i] = lo * (1 - fraction) + up * fraction;
    }
    return GORT_OK;
}

namespace {

// Average interface transmissivity (Stern 1964 / Allen 1973) for incidence cone theta (deg).
// Single-precision pi as in the Fortran original (tav_abs.f90:30).
double tav_interface(double theta, double nr)
{
    const double pi = (double)(std::atan(1.0f) * 4.0f);
    const double rd = pi / 180.;
    const double n2 = nr * nr, np = n2 + 1., nm = n2 - 1.;
    const double a = (nr + 1) * (nr + 1.) / 2.;
    const double k = -((n2 - 1) * (n2 - 1.) / 4.);
    const double sa = std::sin(theta * rd);
    const double b2 = sa * sa - np / 2;
    const double b1 = (theta == 90.) ? 0. : std::sqrt(b2 * b2 + k);
    const double b = b1 - b2;
    const double b3 = b * b * b, a3 = a * a * a;
    const double ts = (k * k / (6 * b3) + k / b - b / 2) - (k * k / (6 * a3) + k / a - a / 2);
    const double nm2 = nm * nm;
    const double tp1 = -(2 * n2 * (b - a) / (np * np));
    const double tp2 = -(2 * n2 * np * std::log(b / a) / nm2);
    const double tp3 = n2 * (1. / b - 1. / a) / 2;
    const double tp4 = 16 * (n2 * n2) * (n2 * n2 + 1) * std::log((2 * np * b - nm2) / (2 * np * a - nm2))
                       / ((np * np * np) * nm2);
    const double tp5 = 16 * (n2 * n2 * n2) * (1. / (2 * np * b - nm2) - 1. / (2 * np * a - nm2)) / (np * np * np);
    return (ts + (tp1 + tp2 + tp3 + tp4 + tp5)) / (2 * (sa * sa));
}

// tav(90) and tav(40) depend on the refractive-index table only: evaluate once.
struct InterfaceTables {
    double t12[GORT_NBANDS], talf[GORT_NBANDS];
    InterfaceTables()
    {
        for (int i = 0; i < GORT_NBANDS; ++i) {
            const double nr = gort_prospect_coeffs[i];
            t12[i] = tav_interface(90., nr);
            talf[i] = tav_interface(40., nr);
        }
    }
};
const InterfaceTables &interface_tables()
{
    static const InterfaceTables t;
    return t;
}

}  // namespace

// table access for the device-side spectra kernel (gort_spectra.hip)
namespace gort {
const float *prospect_coeff_table() { return gort_prospect_coeffs; }
const double *price_eof_table() { return gort_price_eofs; }
void interface_transmissivity_tables(const double **t12, const double **talf)
{
    const InterfaceTables &t = interface_tables();
    *t12 = t.t12;
    *talf = t.talf;
}
}  // namespace gort

namespace {

// tau(k) = (1-k) e^-k + k^2 E1(k); E1 by the NAG S13AAF Chebyshev fits (prospect_DB.f90:100-141)
double plate_transmission(double k)
{
    static const double c_lo[17] = {
        -3.60311230482612224e-13, 3.46348526554087424e-12, -2.99627399604128973e-11,
        2.57747807106988589e-10, -2.09330568435488303e-9, 1.59501329936987818e-8,
        -1.13717900285428895e-7, 7.55292885309152956e-7, -4.64980751480619431e-6,
        2.63830365675408129e-5, -1.37089870978830576e-4, 6.47686503728103400e-4,
        -2.76060141343627983e-3, 1.05306034687449505e-2, -3.57191348753631956e-2,
        1.07774527938978692e-1, -2.96997075145080963e-1};
    static const double c_hi[17] = {
        -1.62806570868460749e-12, -8.95400579318284288e-13, -4.08352702838151578e-12,
        -1.45132988248537498e-11, -8.35086918940757852e-11, -2.13638678953766289e-10,
        -1.10302431467069770e-9, -3.67128915633455484e-9, -1.66980544304104726e-8,
        -6.11774386401295125e-8, -2.70306163610271497e-7, -1.05565006992891261e-6,
        -4.72090467203711484e-6, -1.95076375089955937e-5, -9.16450482931221453e-5,
        -4.05892130452128677e-4, -2.14213055000334718e-3};
    if (k <= 0.0) return 1;
    if (k > 85.0) return 0;
    double yy;
    if (k <= 4.0) {
        const double xx = 0.5 * k - 1.0;
        yy = c_lo[0];
        for (int j = 1; j < 17; ++j) yy = yy * xx + c_lo[j];
        yy = (yy * xx + 8.64664716763387311e-1) * xx + 7.42047691268006429e-1;
        yy = yy - std::log(k);
    } else {
        const double xx = 14.5 / (k + 3.25) - 1.0;
        yy = c_hi[0];
        for (int j = 1; j < 17; ++j) yy = yy * xx + c_hi[j];
        yy = ((yy * xx - 1.06374875116569657e-2) * xx - 8.50699154984571871e-2) * xx + 9.23755307807784058e-1;
        yy = std::exp(-k) * yy / k;
    }
    return (1.0 - k) * std::exp(-k) + k * k * yy;
}

}  // namespace

extern "C" int gort_prospect_d(double N, double Cab, double Car, double Anth, double Cbrown,
                               double Cw, double Cm, double *RT)
{
    if (!RT) return gort::fail(GORT_EINVAL, "gort_prospect_d: null output");
    const InterfaceTables &it = interface_tables();
    const float *nr_t = gort_prospect_coeffs;
    const float *kCab = nr_t + GORT_NBANDS, *kCar = kCab + GORT_NBANDS, *kAnth = kCar + GORT_NBANDS;
    const float *kBrown = kAnth + GORT_NBANDS, *kCw = kBrown + GORT_NBANDS, *kCm = kCw + GORT_NBANDS;
    for (int i = 0; i < GORT_NBANDS; ++i) {
        const double nr = nr_t[i];
        const double k = (Cab * kCab[i] + Car * kCar[i] + Anth * kAnth[i] + Cbrown * kBrown[i]
                          + Cw * kCw[i] + Cm * kCm[i]) / N;
        const double tau = plate_transmission(k);
        // one elementary layer (Allen et al. 1969)
        const double t12 = it.t12[i], talf = it.talf[i];
        const double ralf = 1. - talf, r12 = 1. - t12;
        const double t21 = t12 / (nr * nr), r21 = 1 - t21;
        double denom = 1 - r21 * r21 * (tau * tau);
        const double Ta = talf * tau * t21 / denom;
        const double Ra = ralf + r21 * tau * Ta;
        const double t = t12 * tau * t21 / denom;
        const double r = r12 + r21 * tau * t;
        // N-1 further layers (Stokes 1862)
        const double D = std::sqrt((1. + r + t) * (1. + r - t) * (1. - r + t) * (1. - r - t));
        const double rq = r * r, tq = t * t;
        const double a = (1. + rq - tq + D) / (2 * r);
        const double b = (1. - rq + tq + D) / (2 * t);
        const double bNm1 = std::pow(b, (N - 1));
        const double bN2 = bNm1 * bNm1, a2 = a * a;
        denom = a2 * bN2 - 1.;
        double Rsub = a * (bN2 - 1.) / denom;
        double Tsub = bNm1 * (a2 - 1.) / denom;
        if (r + t >= 1.0) {                          // zero absorption
            Tsub = t / (t + (1. - t) * (N - 1));
            Rsub = 1 - Tsub;
        }
        denom = 1 - Rsub * r;
        RT[GORT_NBANDS + i] = Ta * Tsub / denom;
        RT[i] = Ra + Ta * Rsub * t / denom;
    }
    return GORT_OK;
}

extern "C" int gort_spectra(const gort_leaf_soil *s, const double *wl, int nw,
                            double *rsoil, double *rleaf, double *tleaf)
{
    if (!s || !wl || nw < 0 || !rsoil || !rleaf || !tleaf)
        return gort::fail(GORT_EINVAL, "gort_spectra: bad argument");
    // range errors: soil is checked first, as in main() (gortt.c:224 then :227)
    for (int i = 0; i < nw; ++i)
        if (!(wl[i] >= 400 && wl[i] <= 2500))                 // NaN included
            return gort::fail(GORT_ERANGE, "gortt_price_soil: wavlength out of range (400-2500)");
    if (s->use_alb_soil) {
        for (int i = 0; i < nw; ++i) rsoil[i] = s->alb_soil;
    } else {
        int rc = gort_price_soil(wl, nw, s->rsl, rsoil);
        if (rc) return rc;
    }
    if (s->use_alb_leaf) {
        for (int i = 0; i < nw; ++i) rleaf[i] = tleaf[i] = s->alb_leaf / 2.0;
        return GORT_OK;
    }
    std::vector<double> RT(2 * GORT_NBANDS);
    gort_prospect_d(s->N, s->Cab, s->Car, s->Anth, s->Cbrown, s->Cw, s->Cm, RT.data());
    for (int i = 0; i < nw; ++i) {
        const int upper = (int)(1 + (wl[i] - 400.0) / 1.0);
        const int lower = (int)((wl[i] - 400.0) / 1.0);
        // the reference keeps the interpolation weight in a C float (gortt.c:1338,1364)
        const float fraction = (float)((float)(wl[i] - 400.0) / 1.0 - lower);
        const float omf = 1 - fraction;
        const double ru = upper > GORT_NBANDS - 1 ? 0.0 : RT[upper];
        const double tu = upper > GORT_NBANDS - 1 ? 0.0 : RT[upper + GORT_NBANDS];
        rleaf[i] = RT[lower] * omf + ru * fraction;
        tleaf[i] = RT[lower + GORT_NBANDS] * omf + tu * fraction;
    }
    return GORT_OK;
}

// ------------------------------------------------------------------ Gauss-Legendre

extern "C" void gort_gauleg(double x1, double x2, double *x, double *w, int n)
{
    const int m = (n + 1) / 2;
    const double xm = 0.5 * (x2 + x1), xl = 0.5 * (x2 - x1);
    for (int i = 0; i < m; ++i) {
        double z = std::cos(3.141592654 * (i + 0.75) / (n + 0.5)), z1, pp;
        do {                                   // Newton on P_n, tolerance 3e-11 as the reference
            double p1 = 1.0, p2 = 0.0;
            for (int j = 1; j <= n; ++j) {
                const double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            z1 = z;
            z = z1 - p1 / pp;
        } while (std::fabs(z - z1) > 3.0e-11);
        x[i] = xm - xl * z;
        x[n - 1 - i] = xm + xl * z;
        w[i] = 2.0 * xl / ((1.0 - z * z) * pp * pp);
        w[n - 1 - i] = w[i];
    }
}

// ------------------------------------------------------------------ "%f" formatter

// printf("%f") is where the text boundary spends its time (hundreds of MB of digits per
// second of GPU work).  For |v| < 4e9 the value times 1e6 is formed exactly as a double
// plus an FMA error term, rounded to an integer with ties to even exactly as glibc rounds
// the decimal expansion, and the digits are emitted by hand; anything else goes to snprintf.
namespace {

const char DIGIT_PAIRS[201] =
    "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
    "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";

// round(|v| * 1e6) with ties to even on the EXACT product: av * 1e6 = p + e with p < 2^52, so ulp(p) <= 0.5 and
// |e| <= ulp(p)/2; p - n is exact and a multiple of ulp(p): unless it is exactly +-0.5 the error term cannot move the
// value across a rounding boundary; at +-0.5 the sign of e decides, e == 0 is a true tie
#define GORT_ROUND_E6_BODY                                    \
    const double p = av * 1.0e6;                              \
    const double e = __builtin_fma(av, 1.0e6, -p);            \
    double n = __builtin_nearbyint(p);                        \
    const double a = p - n;                                   \
    if (a == 0.5) { if (e > 0.0) n += 1.0; }                  \
    else if (a == -0.5) { if (e < 0.0) n -= 1.0; }            \
    return (unsigned long long)n;

inline unsigned long long round_e6_generic(double av) { GORT_ROUND_E6_BODY }
// the same with hardware FMA and ROUNDSD instead of two libm calls per value (every host of an MI355X has them;
// checked once at run time)
__attribute__((target("fma,sse4.1"))) inline unsigned long long round_e6_fma(double av) { GORT_ROUND_E6_BODY }

inline char *emit_f6(unsigned long long q, bool neg, char *o)
{
    const unsigned long long ip = q / 1000000ULL;
    const unsigned frac = (unsigned)(q - ip * 1000000ULL);
    if (neg) *o++ = '-';
    if (ip < 10) {
        *o++ = (char)('0' + ip);
    } else {
        char tmp[24];
        int k = 0;
        unsigned long long t = ip;
        do { tmp[k++] = (char)('0' + t % 10); t /= 10; } while (t);
        while (k) *o++ = tmp[--k];
    }
    *o++ = '.';
    const unsigned hi = frac / 10000u, lo = frac - hi * 10000u, mid = lo / 100u;
    std::memcpy(o, DIGIT_PAIRS + 2 * hi, 2);
    std::memcpy(o + 2, DIGIT_PAIRS + 2 * mid, 2);
    std::memcpy(o + 4, DIGIT_PAIRS + 2 * (lo - mid * 100u), 2);
    return o + 6;
}

template <bool FMA> inline int format_f6_one(double v, char *dst)
{
    if (v != v) { std::memcpy(dst, "-nan", 4); return 4; }
    const double av = std::fabs(v);
    if (!(av < 4.0e9)) return std::snprintf(dst, 352, "%f", v);
    const unsigned long long q = FMA ? round_e6_fma(av) : round_e6_generic(av);
    return (int)(emit_f6(q, std::signbit(v), dst) - dst);
}

__attribute__((target("fma,sse4.1"))) long format_row_fma(const double *v, long n, char *dst)
{
    char *o = dst;
    for (long i = 0; i < n; ++i) { o += format_f6_one<true>(v[i], o); *o++ = ' '; }
    return (long)(o - dst);
}

long format_row_generic(const double *v, long n, char *dst)
{
    char *o = dst;
    for (long i = 0; i < n; ++i) { o += format_f6_one<false>(v[i], o); *o++ = ' '; }
    return (long)(o - dst);
}

bool cpu_has_fma()
{
    static const bool has = __builtin_cpu_supports("fma") && __builtin_cpu_supports("sse4.1");
    return has;
}

}  // namespace

extern "C" int gort_format_f6(double v, char *dst)
{
    char tmp[360];
    const long n = cpu_has_fma() ? format_row_fma(&v, 1, tmp) : format_row_generic(&v, 1, tmp);
    std::memcpy(dst, tmp, (size_t)(n - 1));          // without the separator
    return (int)(n - 1);
}

extern "C" long gort_format_f6_row(const double *v, long n, char *dst, size_t cap)
{
    if (!v || !dst || n < 0) return gort::fail(GORT_EINVAL, "gort_format_f6_row: bad argument");
    // a value below 4e9 takes at most 18 bytes with its separator; larger ones (snprintf) up to 318: go one by one
    // when the buffer is not roomy enough for the worst case of the whole row
    if (cap >= (size_t)n * 24) {
        bool small = true;
        for (long i = 0; i < n && small; ++i) small = !(std::fabs(v[i]) >= 4.0e9);      // NaN counts as small
        if (small) return cpu_has_fma() ? format_row_fma(v, n, dst) : format_row_generic(v, n, dst);
    }
    long used = 0;
    for (long i = 0; i < n; ++i) {
        if (cap - (size_t)used < 360) return gort::fail(GORT_EINVAL, "gort_format_f6_row: buffer too small");
        used += gort_format_f6(v[i], dst + used);
        dst[used++] = ' ';
    }
    return used;
}

// ------------------------------------------------------------- probability LUT text

extern "C" long gort_lut_format(const gort_canopy *c, char *buf, size_t cap)
{
    if (!c || !buf) return gort::fail(GORT_EINVAL, "gort_lut_format: bad argument");
    size_t n = 0;
    for (int j = 0; j < 90; ++j) {
        int k = std::snprintf(buf + n, cap - n, "%d %0.40f %0.40f\n", j, c->p_n0[j], c->epgap[j]);
        if (k < 0 || (size_t)k >= cap - n) return gort::fail(GORT_EINVAL, "gort_lut_format: buffer too small");
        n += (size_t)k;
    }
    int k = std::snprintf(buf + n, cap - n, "-1 %0.40f %0.40f\n", c->k_open, c->k_openep);
    if (k < 0 || (size_t)k >= cap - n) return gort::fail(GORT_EINVAL, "gort_lut_format: buffer too small");
    return (long)(n + (size_t)k);
}

// ---- gap-table cache keyed on crown geometry (new surface: generalises -W / -P, gortt.c:123-146) ----

namespace {
struct GapKey { double r, b, h1, h2, lambda, favd; int32_t q08; };

GapKey gap_key_of(const gort_canopy *c)
{
    GapKey k;
    std::memset(&k, 0, sizeof k);                            // padding bytes too: the key is hashed as bytes
    k.r = c->r;  k.b = c->b;  k.h1 = c->h1;  k.h2 = c->h2;  k.lambda = c->lambda;  k.favd = c->favd;
    k.q08 = c->use_q08 ? 1 : 0;
    return k;
}

// FNV-1a over the bit patterns of everything a cache file holds: damaged table rows are a miss, not a wrong hit
uint64_t gap_tables_sum(const double *p_n0, const double *epgap, double k_open, double k_openep)
{
    uint64_t h = 1469598103934665603ull;
    auto eat = [&](const double *v, int n) {
        const unsigned char *p = reinterpret_cast<const unsigned char *>(v);
        for (size_t i = 0; i < sizeof(double) * (size_t)n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    };
    eat(p_n0, GORT_NTH);
    eat(epgap, GORT_NTH);
    eat(&k_open, 1);
    eat(&k_openep, 1);
    return h;
}

std::string gap_cache_path(const char *dir, uint64_t key)
{
    char name[64];
    std::snprintf(name, sizeof name, "/gap-%016llx.lut", (unsigned long long)key);
    return std::string(dir) + name;
}
}  // namespace

extern "C" uint64_t gort_canopy_key(const gort_canopy *c)
{
    if (!c) return 0;
    const GapKey k = gap_key_of(c);
    const unsigned char *p = reinterpret_cast<const unsigned char *>(&k);
    uint64_t h = 1469598103934665603ull;                     // FNV-1a
    for (size_t i = 0; i < sizeof k; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

extern "C" int gort_lut_cache_store(const char *dir, const gort_canopy *c)
{
    if (!dir || !c) return gort::fail(GORT_EINVAL, "gort_lut_cache_store: bad argument");
    const std::string path = gap_cache_path(dir, gort_canopy_key(c));
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    FILE *fp = std::fopen(tmp.c_str(), "w");
    if (!fp) return gort::fail(GORT_EIO, "gort_lut_cache_store: cannot write %s", tmp.c_str());
    for (int j = 0; j < 90; ++j) std::fprintf(fp, "%d %a %a\n", j, c->p_n0[j], c->epgap[j]);
    std::fprintf(fp, "-1 %a %a\n", c->k_open, c->k_openep);
    // p_n0[90] / epgap[90] are not part of the -W format (the reference reads them back as 0); kept here so that a
    // cache hit returns every bit gort_gap_probabilities computed
    const GapKey k = gap_key_of(c);
    std::fprintf(fp, "# gort-gap-lut 1 %a %a %a %a %a %a %d %a %a %016llx\n", k.r, k.b, k.h1, k.h2, k.lambda, k.favd, (int)k.q08,
                 c->p_n0[90], c->epgap[90], (unsigned long long)gap_tables_sum(c->p_n0, c->epgap, c->k_open, c->k_openep));
    const bool ok = std::fflush(fp) == 0 && !std::ferror(fp);
    std::fclose(fp);
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) {
        std::remove(tmp.c_str());
        return gort::fail(GORT_EIO, "gort_lut_cache_store: cannot write %s", path.c_str());
    }
    return GORT_OK;
}

extern "C" int gort_lut_cache_load(const char *dir, gort_canopy *c)
{
    if (!dir || !c) return gort::fail(GORT_EINVAL, "gort_lut_cache_load: bad argument");
    const std::string path = gap_cache_path(dir, gort_canopy_key(c));
    FILE *fp = std::fopen(path.c_str(), "r");
    if (!fp) return 1;
    double pn[GORT_NTH] = {0}, ep[GORT_NTH] = {0}, ko = 0, koe = 0;
    int rows = 0, j;
    double x1, x2;
    bool closed = false;
    while (std::fscanf(fp, "%d %lf %lf", &j, &x1, &x2) == 3) {
        if (j >= 0 && j < 90) { pn[j] = x1; ep[j] = x2; ++rows; }
        else if (j < 0) { ko = x1; koe = x2; closed = true; }
    }
    GapKey f;
    std::memset(&f, 0, sizeof f);
    int version = 0, q08 = 0;
    double pn90 = 0, ep90 = 0;
    unsigned long long sum = 0;
    const int got = std::fscanf(fp, " # gort-gap-lut %d %lf %lf %lf %lf %lf %lf %d %lf %lf %16llx", &version, &f.r, &f.b, &f.h1, &f.h2,
                                &f.lambda, &f.favd, &q08, &pn90, &ep90, &sum);
    std::fclose(fp);
    pn[90] = pn90;
    ep[90] = ep90;
    f.q08 = q08;
    const GapKey k = gap_key_of(c);
    // another geometry with the same hash, a damaged or a foreign file: no entry
    if (got != 11 || version != 1 || rows != 90 || !closed || std::memcmp(&f, &k, sizeof k) != 0 ||
        sum != gap_tables_sum(pn, ep, ko, koe))
        return 1;
    for (int t = 0; t < 90; ++t) { c->p_n0[t] = pn[t]; c->epgap[t] = ep[t]; }
    c->p_n0[90] = pn90;
    c->epgap[90] = ep90;
    c->k_open = ko;
    c->k_openep = koe;
    return GORT_OK;
}

extern "C" int gort_lut_read(const char *path, gort_canopy *c)
{
    FILE *fp = path ? std::fopen(path, "r") : nullptr;
    if (!fp) return gort::fail(GORT_EIO, "error opening probability file: %s", path ? path : "(null)");
    for (int t = 0; t < GORT_NTH; ++t) c->p_n0[t] = c->epgap[t] = 0.0;   // calloc'd in the reference
    c->k_open = c->k_openep = 0.0;
    int j;
    double x1, x2;
    while (std::fscanf(fp, "%d %lf %lf", &j, &x1, &x2) == 3) {
        if (j >= 0) {
            if (j < GORT_NTH) { c->p_n0[j] = x1; c->epgap[j] = x2; }      // the reference would write out of bounds
        } else {
            c->k_open = x1;
            c->k_openep = x2;
        }
    }
    std::fclose(fp);
    return GORT_OK;
}
