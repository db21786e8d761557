// Host build of gort_amd/csrc/gort_math.h for tests/test_math_kernels.py: the same kernels, the three hardware
// primitives spelled in C (the reciprocal estimate deliberately only float-accurate).  Test infrastructure.
#include "gort_math.h"

#define ARRAY_FN(name, expr)                                                   \
    extern "C" void name(const double *x, double *y, long n)                   \
    {                                                                          \
        for (long i = 0; i < n; ++i) { const double v = x[i]; y[i] = (expr); } \
    }
using namespace gort::gm;
ARRAY_FN(gm_exp, exp_(v))
ARRAY_FN(gm_log, log_(v))
ARRAY_FN(gm_atan, atan_(v))
ARRAY_FN(gm_acos, acos_(v))
ARRAY_FN(gm_acos_unit, acos_unit(v))
ARRAY_FN(gm_div180, div_by_constant(v, 180.0, 1.0 / 180.0))
ARRAY_FN(gm_cos, cos_reduced(v))
ARRAY_FN(gm_recip, recip(v))
ARRAY_FN(gm_sqrt, sqrt_(v))
extern "C" void gm_sincos(const double *x, double *s, double *c, long n)
{
    for (long i = 0; i < n; ++i) sincos_reduced(x[i], s[i], c[i]);
}
extern "C" void gm_quot(const double *a, const double *b, double *y, long n)
{
    for (long i = 0; i < n; ++i) y[i] = quot(a[i], b[i]);
}
extern "C" void gm_quot_finite(const double *a, const double *b, double *y, long n)
{
    for (long i = 0; i < n; ++i) y[i] = quot_finite(a[i], b[i]);
}
extern "C" void gm_root_and_inverse(const double *x, double *r, double *i, long n)
{
    for (long k = 0; k < n; ++k) root_and_inverse(x[k], r[k], i[k]);
}
