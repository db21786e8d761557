// gort_tables.hip -- the small tables the expansion kernels read: the band-only two-stream closed forms
// L[member][11][nw] (gortt_brdf.c:348-634 hoisted out of the per-sample loop) with, behind the last member, the first
// member's twelve StreamBand constants per band ([nw][12], what the flat stream kernel keeps per lane) and, for LUT grids, the five
// (sun zenith, band) terms sun[q][5][nw] of every sun row.
#include <hip/hip_runtime.h>

#include "gort_device.h"
#include "gort_internal.h"

namespace gort {
namespace {

// Two-stream closed forms that depend on the band only (gortt_brdf.c:348-634 hoisted)
// blockIdx.y = ensemble member: canopy[m], spectra[m][3][nw] (rsoil, rleaf, tleaf) -> L[m][11][nw]
__global__ __launch_bounds__(256) void lambda_table_kernel(const gort_canopy *__restrict__ canopies, int nw,
                                                            const double *__restrict__ spectra,
                                                            double *__restrict__ Lall, StreamBand *__restrict__ stream_bands)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw) return;
    const long m = blockIdx.y;
    const gort_canopy &c = canopies[m];
    const double *__restrict__ sp = spectra + m * 3 * nw;
    double *__restrict__ L = Lall + m * L_NSLOT * nw;
    const double rs = sp[i], rl = sp[nw + i], tl = sp[2 * nw + i];
    const double omega = rl + tl;
    const double gam = sqrt(1 - omega);
    const double Rff = (1.0 - gam) / (1.0 + gam);
    const double Tff = exp(-(2.0 * gam * c.k * c.elai));
    const double RT = Rff * Tff;
    const double tff = Tff * (1. - Rff * Rff) / (1. - RT * RT);
    const double pff = Rff * (1. - Tff * Tff) / (1. - RT * RT);
    const double kopen = c.k_open + c.k_openep;
    const double tpff = tpff_of(tff, kopen);
    const double gfun = -(4.0 / 9.0) * (rl - tl) / omega;
    const double mgk = (rs / (1.0 - rs * pff)) * (tpff - c.k_open);
    L[L_GAMMA * nw + i] = gam;
    L[L_OMEGA * nw + i] = omega;
    L[L_RFF * nw + i] = Rff;
    L[L_TFF * nw + i] = Tff;
    L[L_tFF * nw + i] = tff;
    L[L_PFF * nw + i] = pff;
    L[L_RS * nw + i] = rs;
    L[L_MGK * nw + i] = mgk;
    L[L_ZF * nw + i] = (tpff - c.k_openep) * rs;
    L[L_TF * nw + i] = tpff * mgk;
    L[L_B * nw + i] = (1.0 - omega) * omega * (1.0 - gfun);
    // the band constants of the stream family's sample, ready-made for the flat stream kernel (single-canopy streams:
    // member 0): the same stream_band() of the same eleven numbers every other stream kernel evaluates per sample
    if (m == 0 && stream_bands) stream_bands[i] = stream_band(load_band(L, nw, i));
}

// the twelve StreamBand constants per band of every member, out[m][nw] (the line kernel's scalar loads; the engine's own
// table behind L is the first member's)
__global__ __launch_bounds__(256) void member_stream_bands_kernel(const double *__restrict__ Lall, int nw, StreamBand *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw) return;
    const long m = blockIdx.y;
    out[m * nw + i] = stream_band(load_band(Lall + m * L_NSLOT * nw, nw, i));
}

// sun[q - q_begin][5][nw] with q = member * nsza + isza: the "sun rows" of an ensemble are (member, sun zenith)
__global__ __launch_bounds__(256) void sun_table_kernel(const gort_canopy *__restrict__ canopies,
                                                         const double *__restrict__ Lall, int nw, gort_grid g,
                                                         int q_begin, int n_q, double *__restrict__ sun)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int js = blockIdx.y;
    if (i >= nw || js >= n_q) return;
    const int q = q_begin + js;
    const int member = q / g.nsza, isza = q - member * g.nsza;
    const gort_canopy &c = canopies[member];
    const double *__restrict__ L = Lall + (long)member * L_NSLOT * nw;
    // sun-only scalars exactly as geometry_core derives them for "vza phi sza 0"
    double vza, sza, saa, raa;
    normalise_angles(0.0, 0.0, g.sza0 + isza * g.dsza, 0.0, vza, sza, saa, raa);
    const SunScalars s = sun_from_zenith(c, sza);
    const SunTerms b = sun_terms(L, nw, i, s, c.k_open, c.k_openep);
    double *o = sun + (long)js * 5 * nw;
    o[0 * nw + i] = b.C0;
    o[1 * nw + i] = b.B;
    o[2 * nw + i] = b.Z;
    o[3 * nw + i] = b.G;
    o[4 * nw + i] = b.T;
}

}  // namespace

int launch_lambda_table(const gort_canopy *canopies_dev, int n_members, int nw, const double *spectra_dev,
                        double *L_dev, void *stream)
{
    if (nw <= 0 || n_members <= 0) return GORT_OK;
    hipLaunchKernelGGL(lambda_table_kernel, dim3((nw + 255) / 256, n_members), dim3(256), 0, (hipStream_t)stream,
                       canopies_dev, nw, spectra_dev, L_dev, reinterpret_cast<StreamBand *>(L_dev + stream_band_table_offset(nw, n_members)));
    return check_launch("lambda_table_kernel");
}

int launch_member_stream_bands(const double *L_dev, int n_members, int nw, double *bands_dev, void *stream)
{
    if (nw <= 0 || n_members <= 0) return GORT_OK;
    if (n_members > 65535) return fail(GORT_EINVAL, "band tables: %d members in one launch (max 65535)", n_members);
    hipLaunchKernelGGL(member_stream_bands_kernel, dim3((nw + 255) / 256, n_members), dim3(256), 0, (hipStream_t)stream, L_dev, nw,
                       reinterpret_cast<StreamBand *>(bands_dev));
    return check_launch("member_stream_bands_kernel");
}

int launch_sun_table(const gort_canopy *canopies_dev, const double *L_dev, int nw, const gort_grid &g, int q_begin,
                     int q_end, double *sun_dev, void *stream)
{
    const int n = q_end - q_begin;
    if (n <= 0 || nw <= 0) return GORT_OK;
    if (n > 65535) return fail(GORT_EINVAL, "sun_table: %d (member, sun zenith) rows in one launch (max 65535)", n);
    hipLaunchKernelGGL(sun_table_kernel, dim3((nw + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, canopies_dev,
                       L_dev, nw, g, q_begin, n, sun_dev);
    return check_launch("sun_table_kernel");
}

}  // namespace gort
