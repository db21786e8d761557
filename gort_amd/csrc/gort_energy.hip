// gort_energy.hip -- spectral albedo, vegetation and soil absorption per sun direction
// (replaces gortt_energy / gortt_albedo, gortt_albedo.c:7-138).
#include <hip/hip_runtime.h>

#include "gort_geometry.h"

namespace gort {
namespace {

// One 512-thread workgroup per angle line = the 32 x 16 Gauss-Legendre nodes of the
// viewing hemisphere (gortt_albedo.c:89-134), one node per thread.  By linearity of
// rsurf in the five angle coefficients the quadrature is applied to the coefficients
// (wavefront shuffle + LDS reduction), then every band costs 5 FMAs:
//   albedo(band) = sum_k [sum_nodes w_node a_k(node)] b_k(sun zenith, band).
constexpr int ENERGY_THREADS = 512;

__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// blockIdx.y = ensemble member (its canopy and band tables); the nA angle lines are shared by the members;
// energy[member][nA][nw][3]
__global__ __launch_bounds__(ENERGY_THREADS) void energy_kernel(const gort_canopy *__restrict__ canopies,
                                                                 const double *__restrict__ Lall, int nw,
                                                                 const double *__restrict__ angles,
                                                                 const double *__restrict__ nodes,   // [512][3] vaa, vza, weight
                                                                 double *__restrict__ energy_all)
{
    __shared__ double s_part[5][ENERGY_THREADS / 64];
    __shared__ double s_abar[5];
    __shared__ double s_sun[6];
    const long member = blockIdx.y;
    const gort_canopy &c = canopies[member];
    const double *__restrict__ L = Lall + member * L_NSLOT * nw;
    double *__restrict__ energy = energy_all + member * (long)gridDim.x * nw * 3;
    const long a = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    double vza, sza, saa, raa;
    normalise_angles(angles[4 * a], angles[4 * a + 1], angles[4 * a + 2], angles[4 * a + 3], vza, sza, saa, raa);
    // node geometry (gortt_albedo.c:91-105): vaa = pi + pi x_i in (0, 2pi); vza = acos(x_j)
    {
#pragma clang fp contract(off)
        const double vaa = nodes[3 * tid];
        raa = saa - vaa;
        raa = fabs((raa - 2 * PI * (int)(0.5 + raa * INV_PI * 0.5)));
    }
    vza = nodes[3 * tid + 1];
    const double w = nodes[3 * tid + 2];
    GeomOut g;
    geometry_core(c, vza, sza, raa, g);
    double rec[GORT_COEF_STRIDE];
    store_coef(rec, c, g);
    double part[5] = {w * rec[A_C], w * rec[A_B], w * rec[A_Z], w * rec[A_G], w * rec[A_T]};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        part[k] = wave_sum(part[k]);
        if (lane == 0) s_part[k][wave] = part[k];
    }
    if (tid == 0) {
        s_sun[0] = g.sun.fd;  s_sun[1] = g.sun.mu;  s_sun[2] = g.sun.t0;
        s_sun[3] = g.sun.tp0; s_sun[4] = g.sun.eps; s_sun[5] = g.sun.pn0;
    }
    __syncthreads();
    if (tid < 5) {
        double t = 0.0;
        for (int q = 0; q < ENERGY_THREADS / 64; ++q) t += s_part[tid][q];
        s_abar[tid] = t;
    }
    __syncthreads();
    SunScalars s;
    s.fd = s_sun[0];  s.mu = s_sun[1];  s.t0 = s_sun[2];  s.tp0 = s_sun[3];  s.eps = s_sun[4];  s.pn0 = s_sun[5];
    const double aC = s_abar[0], aB = s_abar[1], aZ = s_abar[2], aG = s_abar[3], aT = s_abar[4];
    for (int i = tid; i < nw; i += ENERGY_THREADS) {
        const SunTerms b = sun_terms(L, nw, i, s, c.k_open, c.k_openep);
        const double albedo = dot5(aC, aB, aZ, aG, aT, b.C0, b.B, b.Z, b.G, b.T);
        const double rs = L[L_RS * nw + i];
        // energy balance, Lambertian background (gortt_albedo.c:39-52)
        const double Fu2 = b.G * s.pn0 + b.Z * (1. - s.pn0);
        const double Fd2 = s.pn0 + b.Z * (1. - s.pn0) / rs;
        double *o = energy + (a * nw + i) * 3;
        o[0] = albedo;
        o[1] = 1. - albedo - Fd2 + Fu2;
        o[2] = Fd2 - Fu2;
    }
}

}  // namespace

int launch_energy(const gort_canopy *canopies_dev, int n_members, const double *L_dev, int nw,
                  const double *angles_dev, long nA, const double *nodes_dev, double *energy_dev, void *stream)
{
    if (nA <= 0 || nw <= 0 || n_members <= 0) return GORT_OK;
    if (n_members > 65535) return fail(GORT_EINVAL, "energy: %d members in one launch (max 65535)", n_members);
    hipLaunchKernelGGL(energy_kernel, dim3((unsigned)nA, (unsigned)n_members), dim3(ENERGY_THREADS), 0,
                       (hipStream_t)stream, canopies_dev, L_dev, nw, angles_dev, nodes_dev, energy_dev);
    return check_launch("energy_kernel");
}

}  // namespace gort
