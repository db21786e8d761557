// gort_math.h -- the fp64 elementary functions of the geometry stage, written for the arguments that stage has.
//
// Round 3 measured 2790 fp64 VALU instructions per angle line, most of them in the device library's general-purpose
// sin / cos / tan / exp / log / atan / acos (150, 150, 180, 42, 98, 83, 95 instructions each, with huge-argument
// reduction and every special case) and in ~30 correctly rounded divisions (12 each).  The geometry's arguments are
// bounded by construction - zeniths in [0, pi/2], azimuths folded into [0, 2 pi], gap probabilities in [0, 1],
// exponents that are optical depths - so the functions here do ONE Cody-Waite reduction step, evaluate the classic
// minimax kernels (the coefficient sets of Sun's fdlibm, whose forms they follow) with explicit FMAs, and replace
// division by the hardware reciprocal + Newton steps.  Each is within ~1 ulp of the correctly rounded result over the
// range it documents (tests/test_math_kernels.py checks them on the CPU against mpmath: the header compiles for the
// host too, with the three hardware primitives spelled in C); outside that range they either fall back to the device
// library (sincos) or produce the IEEE special value (exp, log, quot), so that NaN / inf patterns are those of libm.
//
// Every function pins its own arithmetic (#pragma clang fp contract(off); the fused operations are the ones written as
// fma_): what a function returns must not depend on what the compiler finds to contract in the kernel it is inlined into -
// kernels that must write the same bits (tests/test_stream_forms.py, test_energy_forms.py) inline them in different places.
//
// Reference call sites these serve: gortt_brdf.c:23-100 (overlap: sqrt, acos, sin), :118-238 (Kc: exp, acos, cos, tan),
// :638-702 (hot spot: log, exp, sqrt), gortt.c:581-588 (primed angles: atan, tan, cos).
#ifndef GORT_MATH_H
#define GORT_MATH_H

#include <cmath>
#include <cstdint>
#include <cstring>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GM_FN __host__ __device__ __forceinline__
#else
#define GM_FN static inline
#endif

namespace gort {
namespace gm {

// ---- the four hardware primitives (host spelling for the CPU test of the kernels) ----
GM_FN double hw_rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(x);
#else
    double r = 1.0 / x;                             // cut to 24 bits, about as coarse as the hardware's estimate: the
    uint64_t u;                                     // Newton steps must do the work
    std::memcpy(&u, &r, 8);
    u &= ~((uint64_t(1) << 29) - 1);
    std::memcpy(&r, &u, 8);
    return r;
#endif
}
GM_FN double hw_ldexp(double x, int e)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ldexp(x, e);
#else
    return std::ldexp(x, e);
#endif
}
GM_FN double hw_frexp(double x, int &e)              // mantissa in [0.5, 1)
{
#if defined(__HIP_DEVICE_COMPILE__)
    e = __builtin_amdgcn_frexp_exp(x);
    return __builtin_amdgcn_frexp_mant(x);
#else
    return std::frexp(x, &e);
#endif
}
GM_FN double hw_rsq(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(x);
#else
    double r = 1.0 / std::sqrt(x);
    uint64_t u;
    std::memcpy(&u, &r, 8);
    u &= ~((uint64_t(1) << 29) - 1);
    std::memcpy(&r, &u, 8);
    return r;
#endif
}
GM_FN double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// a b + c with c a CONSTANT of the polynomial.  Left to itself the compiler forms v_fmac_f64 (dst = a b + dst) and builds
// every constant in a vector register pair first - two v_mov_b32 per Horner step, half as many issue cycles again as
// the step itself.  The constant bus takes one scalar pair per instruction, so the step is spelled with its constant in
// scalar registers (two s_mov_b32 on the scalar unit, in the shadow of the vector pipe).
GM_FN double fma_c(double a, double b, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
#else
    return __builtin_fma(a, b, c);
#endif
}

// 1 / b to ~1 ulp: estimate + one third-order step (e = 1 - b r; r (1 + e + e^2)), as stream_reciprocal()
GM_FN double recip(double b)
{
#pragma clang fp contract(off)
    const double r = hw_rcp(b);
    const double e = fma_(-b, r, 1.0);
    return fma_(r, fma_(e, e, e), r);
}

// a / b within one ulp (reciprocal, product, one residual correction); zero / infinite / NaN operands as IEEE
GM_FN double quot(double a, double b)
{
#pragma clang fp contract(off)
    const double r = recip(b);
    const double q = a * r;
    const double q1 = fma_(fma_(-b, q, a), r, q);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_div_fixup(q1, b, a);
#else
    return (b == 0.0 || !std::isfinite(b) || !std::isfinite(a) || !std::isfinite(q1)) ? a / b : q1;
#endif
}

// a / b for finite b != 0 (no special-case pass: a zero or infinite b gives NaN where IEEE gives inf or 0)
GM_FN double quot_finite(double a, double b)
{
#pragma clang fp contract(off)
    const double r = recip(b);
    const double q = a * r;
    return fma_(fma_(-b, q, a), r, q);
}

// quot_finite(a, b) with r = recip(b) formed ahead: the same product and residual correction, so the same bits - for
// denominators that many quotients share (a LUT row's t1 under its 361 azimuth nodes' overlaps: 7 of a division's 10 issue slots)
GM_FN double quot_finite_with(double a, double b, double recip_of_b)
{
#pragma clang fp contract(off)
    const double q = a * recip_of_b;
    return fma_(fma_(-b, q, a), recip_of_b, q);
}

// a / C correctly rounded for a constant C whose reciprocal Y = RN(1 / C) the caller supplies: two Markstein steps
// (q + (a - C q) Y, each residual exact in an FMA; the first makes q faithful, the second rounds it correctly).  What
// `x * PI / 180.0` of the reference's main() needs: the radians must be the reference's to the bit (a zenith of exactly
// 90 degrees sits on the last node of the gap tables, gortt.c:872-915).
GM_FN double div_by_constant(double a, double C, double Y)
{
#pragma clang fp contract(off)
    double q = a * Y;
    q = fma_(fma_(-C, q, a), Y, q);
    return fma_(fma_(-C, q, a), Y, q);
}

// ---- square roots: the hardware's reciprocal-root estimate + two coupled Newton steps (g -> sqrt w, h -> 1 / (2 sqrt w))
// and one residual correction of g; no input scaling (nothing here is below 2^-700 without being 0)
GM_FN void root_steps(double w, double &g, double &h)
{
#pragma clang fp contract(off)
    const double y = hw_rsq(w);
    g = w * y;
    h = 0.5 * y;
    double e = fma_(-h, g, 0.5);
    g = fma_(g, e, g);
    h = fma_(h, e, h);
    e = fma_(-h, g, 0.5);
    g = fma_(g, e, g);
    h = fma_(h, e, h);
}
// sqrt(w): 0 -> 0, inf -> inf, negative -> NaN, NaN -> NaN
GM_FN double sqrt_(double w)
{
#pragma clang fp contract(off)
    double g, h;
    root_steps(w, g, h);
    g = fma_(fma_(-g, g, w), h, g);
    return (w == 0.0 || w == __builtin_inf()) ? w : g;
}
// sqrt(w) and 1 / sqrt(w) for finite w > 0 (the secant and cosine of a primed zenith from 1 + tan^2)
GM_FN void root_and_inverse(double w, double &root, double &inv)
{
#pragma clang fp contract(off)
    double g, h;
    root_steps(w, g, h);
    root = fma_(fma_(-g, g, w), h, g);
    inv = h + h;
}

// ---- sine and cosine -------------------------------------------------------------------------------------------------
// |x| <= 2^18: k = round(x 2/pi), r = x - k pi/2 in two FMAs (pi/2 = HI + LO: the FMA forms x - k HI exactly before it
// rounds), then the kernels on [-pi/4, pi/4].  Beyond that (a zenith of 1e6 degrees is a legal input line) the caller
// takes the device library.
constexpr double TWO_OVER_PI = 6.36619772367581382433e-01;
constexpr double PIO2_HI = 1.57079632679489655800e+00;
constexpr double PIO2_LO = 6.12323399573676603587e-17;
constexpr double SINCOS_MAX = 262144.0;

GM_FN double sin_kernel(double r, double z)
{
#pragma clang fp contract(off)
    double p = fma_c(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    p = fma_c(z, p, 2.75573137070700676789e-06);
    p = fma_c(z, p, -1.98412698298579493134e-04);
    p = fma_c(z, p, 8.33333333332248946124e-03);
    p = fma_c(z, p, -1.66666666666666324348e-01);
    return fma_(z * r, p, r);
}
GM_FN double cos_kernel(double z)
{
#pragma clang fp contract(off)
    double p = fma_c(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    p = fma_c(z, p, -2.75573143513906633035e-07);
    p = fma_c(z, p, 2.48015872894767294178e-05);
    p = fma_c(z, p, -1.38888888888741095749e-03);
    p = fma_c(z, p, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    return w + fma_(z * z, p, (1.0 - w) - hz);
}
// the reduced argument and the quadrant
GM_FN double reduce_pio2(double x, int &n)
{
#pragma clang fp contract(off)
    const double k = __builtin_rint(x * TWO_OVER_PI);
    n = (int)k;
    return fma_(-k, PIO2_LO, fma_(-k, PIO2_HI, x));
}
GM_FN void sincos_reduced(double x, double &s, double &c)
{
#pragma clang fp contract(off)
    int n;
    const double r = reduce_pio2(x, n), z = r * r;
    const double sk = sin_kernel(r, z), ck = cos_kernel(z);
    const double a = (n & 1) ? ck : sk, b = (n & 1) ? sk : ck;
    s = (n & 2) ? -a : a;
    c = ((n + 1) & 2) ? -b : b;
}
GM_FN double cos_reduced(double x)
{
#pragma clang fp contract(off)
    int n;
    const double r = reduce_pio2(x, n), z = r * r;
    const double v = (n & 1) ? sin_kernel(r, z) : cos_kernel(z);
    return ((n + 1) & 2) ? -v : v;
}

// ---- exp: k = round(x / ln 2), r = x - k ln 2 (two FMAs), Taylor polynomial of degree 13 on |r| <= ln2 / 2 (truncation
// 2e-17), scaled by 2^k with the hardware's ldexp (gradual underflow and overflow come out of it).  For x <= 709 (the
// geometry's exponents are optical depths and Kuusk's bounded hot-spot term): -inf -> 0, NaN -> NaN.
GM_FN double exp_(double x)
{
#pragma clang fp contract(off)
    const double k = __builtin_rint(x * 1.44269504088896338700e+00);
    const double r = fma_(-k, 2.31904681384629955842e-17, fma_(-k, 6.93147180559945286227e-01, x));
    double p = fma_c(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
    p = fma_c(p, r, 1.0 / 39916800.0);
    p = fma_c(p, r, 1.0 / 3628800.0);
    p = fma_c(p, r, 1.0 / 362880.0);
    p = fma_c(p, r, 1.0 / 40320.0);
    p = fma_c(p, r, 1.0 / 5040.0);
    p = fma_c(p, r, 1.0 / 720.0);
    p = fma_c(p, r, 1.0 / 120.0);
    p = fma_c(p, r, 1.0 / 24.0);
    p = fma_c(p, r, 1.0 / 6.0);
    p = fma_(p, r, 0.5);
    p = fma_(p, r, 1.0);
    p = fma_(p, r, 1.0);
    // a k far below the subnormals is cut (ldexp underflows to 0 by itself); only where |x| is so large that r is
    // no longer a remainder (2^51 and up) must the result be forced.  (+inf in gives NaN: no caller has it.)
    const double y = hw_ldexp(p, (int)__builtin_fmax(k, -4000.0));
    return x < -1e15 ? 0.0 : y;
}

// ---- log: x = 2^e m, m in [sqrt(1/2), sqrt(2)); f = m - 1, s = f / (2 + f); log(1 + f) = 2 s + s^3 ... as fdlibm's
// e_log.c (its seven coefficients).  0 -> -inf, negative -> NaN, inf -> inf, NaN -> NaN; subnormal arguments are fine
// (the gap probabilities at the horizon are ~1e-65, at worst 0).
GM_FN double log_(double x)
{
#pragma clang fp contract(off)
    int e;
    double m = hw_frexp(x, e);                      // [0.5, 1)
    const bool low = m < 7.07106781186547524401e-01;
    m = low ? 2.0 * m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * recip(2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma_c(w, fma_c(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma_c(w, fma_c(w, fma_c(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                                6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    double y = fma_(dk, 6.93147180369123816490e-01, -((hfsq - fma_(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
    y = x == 0.0 ? -__builtin_inf() : y;
    y = x < 0.0 ? __builtin_nan("") : y;
    y = x == __builtin_inf() ? x : y;
    return y;                                       // NaN in: frexp keeps it, every step propagates it
}

// ---- atan for any finite x (the primed zenith, atan((b/r) tan theta)): fdlibm's s_atan.c - four break points, one
// division, eleven coefficients.  Its five ranges differ in t = (A |x| + B) / (C |x| + D) and in the angle atan(c_i) that
// is added back (hi + lo): six numbers per range, fetched from a table by the range's index instead of being selected
// out of literals (four nested selects of four doubles were 40 of the function's 99 instructions).
struct AtanRange { double A, B, C, D, hi, lo; };
#if defined(__HIP_DEVICE_COMPILE__)
__device__
#endif
static const AtanRange ATAN_RANGES[5] = {
    {1.0, 0.0, 0.0, 1.0, 0.0, 0.0},                                                             // |x| < 7/16: t = x
    {2.0, -1.0, 1.0, 2.0, 4.63647609000806093515e-01, 2.26987774529616870924e-17},              // < 11/16: (2x-1)/(2+x), atan(1/2)
    {1.0, -1.0, 1.0, 1.0, 7.85398163397448278999e-01, 3.06161699786838301793e-17},              // < 19/16: (x-1)/(x+1), pi/4
    {1.0, -1.5, 1.5, 1.0, 9.82793723247329054082e-01, 1.39033110312309984516e-17},              // < 39/16: (x-1.5)/(1+1.5x), atan(3/2)
    {0.0, -1.0, 1.0, 0.0, 1.57079632679489655800e+00, 6.12323399573676603587e-17},              // beyond: -1/x, pi/2
};
GM_FN double atan_(double x)
{
#pragma clang fp contract(off)
    const double ax = __builtin_fabs(x);
    const int id = (ax >= 0.4375) + (ax >= 0.6875) + (ax >= 1.1875) + (ax >= 2.4375);
    const AtanRange g = ATAN_RANGES[id];
    const double t = fma_(ax, g.A, g.B) * recip(fma_(ax, g.C, g.D));
    const double z = t * t, w = z * z;
    double s1 = fma_c(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02);
    s1 = fma_c(w, s1, 6.66107313738753120669e-02);
    s1 = fma_c(w, s1, 9.09088713343650656196e-02);
    s1 = fma_c(w, s1, 1.42857142725034663711e-01);
    s1 = fma_c(w, s1, 3.33333333333329318027e-01);
    double s2 = fma_c(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02);
    s2 = fma_c(w, s2, -7.69187620504482999495e-02);
    s2 = fma_c(w, s2, -1.11111104054623557880e-01);
    s2 = fma_c(w, s2, -1.99999999998764832476e-01);
    const double y = g.hi - ((t * fma_(z, s1, w * s2) - g.lo) - t);
    return __builtin_copysign(ax > 1e300 ? PIO2_HI : y, x);     // the reciprocal of inf is not 0 here; NaN stays NaN
}

// ---- acos on [-1, 1] (the callers clamp): fdlibm's e_acos.c - a rational R(z) = z P(z) / Q(z), |x| < 1/2 directly,
// otherwise through sqrt((1 -+ x) / 2).
GM_FN double acos_rational(double z)
{
#pragma clang fp contract(off)
    double p = fma_c(z, 3.47933107596021167570e-05, 7.91534994289814532176e-04);
    p = fma_c(z, p, -4.00555345006794114027e-02);
    p = fma_c(z, p, 2.01212532134862925881e-01);
    p = fma_c(z, p, -3.25565818622400915405e-01);
    p = fma_c(z, p, 1.66666666666666657415e-01);
    double q = fma_c(z, 7.70381505559019352791e-02, -6.88283971605453293030e-01);
    q = fma_c(z, q, 2.02094576023350569471e+00);
    q = fma_c(z, q, -2.40339491173441421878e+00);
    q = fma_(z, q, 1.0);
    return z * p * recip(q);
}
// acos on [0, 1]: the overlap function's argument (h/b) t2 / t1, clamped (gortt_brdf.c:83-90).  For x >= 1/2 the root of
// z = (1 - x) / 2 comes from the coupled Newton steps as g + (z - g^2) h, root and rounding error apart, which is what
// fdlibm's 2 (s + (R s + c)) wants; x = 1 -> 0.
GM_FN double acos_unit(double x)
{
#pragma clang fp contract(off)
    const bool mid = x < 0.5;
    const double z = mid ? x * x : 0.5 * (1.0 - x);
    const double R = acos_rational(z);
    const double y_mid = PIO2_HI - (x - fma_(-x, R, PIO2_LO));
    double g, h;
    root_steps(z, g, h);
    const double c = fma_(-g, g, z) * h;
    const double y_pos = 2.0 * (g + fma_(R, g, c));
    const double y = mid ? y_mid : y_pos;
    return x == 1.0 ? 0.0 : y;                      // z = 0: the estimate of 1 / sqrt(0) is inf
}
GM_FN double acos_(double x)
{
#pragma clang fp contract(off)
    const double ax = __builtin_fabs(x);
    const bool mid = ax < 0.5;
    // one evaluation of R and one root serve all three ranges
    const double z = mid ? x * x : 0.5 * (1.0 - ax);
    const double R = acos_rational(z);
    const double s = sqrt_(z);                      // unused (but harmless) for |x| < 1/2
    // |x| < 1/2:  pi/2 - (x - (pio2_lo - x R))
    const double y_mid = PIO2_HI - (x - fma_(-x, R, PIO2_LO));
    // x <= -1/2:  pi - 2 (s + (R s - pio2_lo))
    const double y_neg = 3.14159265358979311600e+00 - 2.0 * (s + fma_(R, s, -PIO2_LO));
    // x >= 1/2:   2 (s + R s) with the root's rounding error put back: c = (z - s^2) / (2 s)
    const double c = fma_(-s, s, z) * recip(s + s);
    const double y_pos = 2.0 * (s + fma_(R, s, c));
    double y = mid ? y_mid : (x < 0.0 ? y_neg : y_pos);
    y = ax == 1.0 ? (x < 0.0 ? 3.14159265358979311600e+00 : 0.0) : y;      // s = 0: 0 / 0 in c
    return y;
}

}  // namespace gm
}  // namespace gort
#endif
