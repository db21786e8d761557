// gort_spectra.hip -- leaf and soil spectra on the device, for ensembles.
//
// The single-canopy `gortt` path keeps PROSPECT-D and the Price soil model on the host
// (gort_host.cpp) as north_star prescribes.  An ensemble (BASELINE config 5: 1000
// members with their own N, Cab, Cw, Cm, rsl1 ...) needs 1000 x 2101 plate-model
// evaluations per update; one thread per (member, band) removes that serial host stage.
// Same equations as gort_prospect_d / gort_price_soil (prospect_DB.f90:94-189,
// gortt.c:1286-1374), with the band-only interface transmissivities tav(90), tav(40)
// precomputed on the host; device exp/log/pow differ from glibc by ulps, which the plate
// model amplifies to ~1e-13 (1e-8 at exactly zero absorption) - far inside 1e-5.
#include <hip/hip_runtime.h>

#include "gort_internal.h"

namespace gort {
namespace {

// tau(k) = (1-k) e^-k + k^2 E1(k), E1 from the NAG S13AAF Chebyshev fits (prospect_DB.f90:100-141)
__device__ double plate_tau(double k)
{
    if (k <= 0.0) return 1.0;
    if (k > 85.0) return 0.0;
    double yy;
    if (k <= 4.0) {
        const double x = 0.5 * k - 1.0;
        yy = -3.60311230482612224e-13;
        yy = yy * x + 3.46348526554087424e-12;   yy = yy * x - 2.99627399604128973e-11;
        yy = yy * x + 2.57747807106988589e-10;   yy = yy * x - 2.09330568435488303e-9;
        yy = yy * x + 1.59501329936987818e-8;    yy = yy * x - 1.13717900285428895e-7;
        yy = yy * x + 7.55292885309152956e-7;    yy = yy * x - 4.64980751480619431e-6;
        yy = yy * x + 2.63830365675408129e-5;    yy = yy * x - 1.37089870978830576e-4;
        yy = yy * x + 6.47686503728103400e-4;    yy = yy * x - 2.76060141343627983e-3;
        yy = yy * x + 1.05306034687449505e-2;    yy = yy * x - 3.57191348753631956e-2;
        yy = yy * x + 1.07774527938978692e-1;    yy = yy * x - 2.96997075145080963e-1;
        yy = (yy * x + 8.64664716763387311e-1) * x + 7.42047691268006429e-1;
        yy = yy - log(k);
    } else {
        const double x = 14.5 / (k + 3.25) - 1.0;
        yy = -1.62806570868460749e-12;
        yy = yy * x - 8.95400579318284288e-13;   yy = yy * x - 4.08352702838151578e-12;
        yy = yy * x - 1.45132988248537498e-11;   yy = yy * x - 8.35086918940757852e-11;
        yy = yy * x - 2.13638678953766289e-10;   yy = yy * x - 1.10302431467069770e-9;
        yy = yy * x - 3.67128915633455484e-9;    yy = yy * x - 1.66980544304104726e-8;
        yy = yy * x - 6.11774386401295125e-8;    yy = yy * x - 2.70306163610271497e-7;
        yy = yy * x - 1.05565006992891261e-6;    yy = yy * x - 4.72090467203711484e-6;
        yy = yy * x - 1.95076375089955937e-5;    yy = yy * x - 9.16450482931221453e-5;
        yy = yy * x - 4.05892130452128677e-4;    yy = yy * x - 2.14213055000334718e-3;
        yy = ((yy * x - 1.06374875116569657e-2) * x - 8.50699154984571871e-2) * x + 9.23755307807784058e-1;
        yy = exp(-k) * yy / k;
    }
    return (1.0 - k) * exp(-k) + k * k * yy;
}

// leaf reflectance / transmittance of native band `ib` (0..2100)
__device__ void prospect_band(const gort_leaf_soil &p, const float *__restrict__ coef, const double *__restrict__ t12v,
                              const double *__restrict__ talfv, int ib, double &R, double &T)
{
    const int NB = GORT_NBANDS;
    const double nr = coef[ib];
    const double k = (p.Cab * coef[NB + ib] + p.Car * coef[2 * NB + ib] + p.Anth * coef[3 * NB + ib] +
                      p.Cbrown * coef[4 * NB + ib] + p.Cw * coef[5 * NB + ib] + p.Cm * coef[6 * NB + ib]) / p.N;
    const double tau = plate_tau(k);
    const double t12 = t12v[ib], talf = talfv[ib];
    const double ralf = 1. - talf, r12 = 1. - t12;
    const double t21 = t12 / (nr * nr), r21 = 1 - t21;
    double denom = 1 - r21 * r21 * (tau * tau);
    const double Ta = talf * tau * t21 / denom;
    const double Ra = ralf + r21 * tau * Ta;
    const double t = t12 * tau * t21 / denom;
    const double r = r12 + r21 * tau * t;
    const double D = sqrt((1. + r + t) * (1. + r - t) * (1. - r + t) * (1. - r - t));
    const double rq = r * r, tq = t * t;
    const double a = (1. + rq - tq + D) / (2 * r);
    const double b = (1. - rq + tq + D) / (2 * t);
    const double bNm1 = pow(b, (p.N - 1));
    const double bN2 = bNm1 * bNm1, a2 = a * a;
    denom = a2 * bN2 - 1.;
    double Rsub = a * (bN2 - 1.) / denom;
    double Tsub = bNm1 * (a2 - 1.) / denom;
    if (r + t >= 1.0) {
        Tsub = t / (t + (1. - t) * (p.N - 1));
        Rsub = 1 - Tsub;
    }
    denom = 1 - Rsub * r;
    T = Ta * Tsub / denom;
    R = Ra + Ta * Rsub * t / denom;
}

// spectra[m][3][nw] = rsoil, rleaf, tleaf of member m at wavelengths wl[nw]
__global__ __launch_bounds__(256) void member_spectra_kernel(const gort_leaf_soil *__restrict__ leaf, int nw,
                                                              const double *__restrict__ wl,
                                                              const float *__restrict__ coef,
                                                              const double *__restrict__ t12v,
                                                              const double *__restrict__ talfv,
                                                              const double *__restrict__ eof,
                                                              double *__restrict__ spectra)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw) return;
    const long m = blockIdx.y;
    const gort_leaf_soil &p = leaf[m];
    double *__restrict__ out = spectra + m * 3 * nw;
    const double w = wl[i];
    // Price soil (gortt.c:1305-1320)
    double rs;
    if (p.use_alb_soil) {
        rs = p.alb_soil;
    } else {
        const int upper = (int)(1. + (w - 400) / 5.0), lower = (int)((w - 400) / 5.0);
        const double fr = (w - 400.) / 5.0 - lower;
        const double *v1 = eof, *v2 = eof + 421, *v3 = eof + 842, *v4 = eof + 1263;
        const double lo = p.rsl[0] * v1[lower] + p.rsl[1] * v2[lower] + p.rsl[2] * v3[lower] + p.rsl[3] * v4[lower];
        const double up = upper > 420 ? 0.0
            : p.rsl[0] * v1[upper] + p.rsl[1] * v2[upper] + p.rsl[2] * v3[upper] + p.rsl[3] * v4[upper];
        rs = lo * (1 - fr) + up * fr;
    }
    out[i] = rs;
    // PROSPECT-D + linear interpolation with a float weight (gortt.c:1355-1368)
    if (p.use_alb_leaf) {
        out[nw + i] = out[2 * nw + i] = p.alb_leaf / 2.0;
        return;
    }
    const int upper = (int)(1 + (w - 400.0) / 1.0), lower = (int)((w - 400.0) / 1.0);
    const float fraction = (float)((float)(w - 400.0) / 1.0 - lower);
    const float omf = 1 - fraction;
    double Rl, Tl, Ru = 0.0, Tu = 0.0;
    prospect_band(p, coef, t12v, talfv, lower, Rl, Tl);
    if (fraction != 0.0f && upper <= GORT_NBANDS - 1) prospect_band(p, coef, t12v, talfv, upper, Ru, Tu);
    out[nw + i] = Rl * omf + Ru * fraction;
    out[2 * nw + i] = Tl * omf + Tu * fraction;
}

}  // namespace

int launch_member_spectra(const gort_leaf_soil *leaf_dev, int n_members, int nw, const double *wl_dev,
                          const float *coef_dev, const double *t12_dev, const double *talf_dev,
                          const double *eof_dev, double *spectra_dev, void *stream)
{
    if (n_members <= 0 || nw <= 0) return GORT_OK;
    if (n_members > 65535) return fail(GORT_EINVAL, "member_spectra: %d members in one launch (max 65535)", n_members);
    hipLaunchKernelGGL(member_spectra_kernel, dim3((nw + 255) / 256, n_members), dim3(256), 0, (hipStream_t)stream,
                       leaf_dev, nw, wl_dev, coef_dev, t12_dev, talf_dev, eof_dev, spectra_dev);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(GORT_ENODEVICE, "member_spectra_kernel: %s", hipGetErrorString(err));
    return GORT_OK;
}

}  // namespace gort
