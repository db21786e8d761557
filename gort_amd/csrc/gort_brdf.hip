// gort_brdf.hip -- per-sample GORT BRDF and hemispherical albedo on gfx950.
//
// The reference evaluates, per (angle line, wavelength), a tree of ~40 tiny functions
// (gortt.c:385-578 + gortt_brdf.c) that recompute the same exponentials dozens of
// times.  Here the work is factored by what each term depends on:
//
//   wavelength only      L[11][nw]          lambda_table_kernel      (once per spectra set)
//   angle tuple only     coef[nA][16]       geometry_*_kernel        (Kc,Kg,Kt,Kz, hot spot, ...)
//   (sun zenith, band)   C0,B,Z,G,T         sun_terms()              (5 numbers)
//   sample               rsurf = aC*C0 + aB*B + aZ*Z + aG*G + aT*T   (5 FMAs, 8 B stored)
//
// which is an exact regrouping of gortt.c:484-557 (no approximation; rounding differs
// at the 1e-16 level).  The LUT kernel keeps the five (sun zenith, band) numbers of its
// bands in registers, reads the five angle coefficients through the scalar cache and
// does nothing else but FMA + store: it is bound by HBM write bandwidth (8 B/sample).
//
// No MFMA: K=5 is not a matrix-core shape and the kernel is store-bound anyway.
#include <hip/hip_runtime.h>
#include <vector>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gort_device.h"

namespace gort {
namespace {

// mutual-shadowing overlap O(theta_s', theta_v', phi)  (gortt_brdf.c:23-100)
// Unfused on purpose: for equal primed zeniths and phi = 0 the reference gets d = t^2 + t^2 - 2 t t = 0
// EXACTLY; an FMA leaves 1e-16 t^2 of product rounding in d, sqrt turns it into 1e-8 t, and Kc at 89/89 deg
// moves by 3e-8.  All geometry below keeps plain IEEE multiply/add for the same reason.
__device__ inline double overlap(double hb, const Primed &s, const Primed &v, double cphi, double sphi)
{
#pragma clang fp contract(off)
    const double d = s.t * s.t + v.t * v.t - 2.0 * s.t * v.t * cphi;
    const double D = sqrt(ref_max(0.0, d));
    const double x = s.t * v.t * sphi;
    const double t2 = sqrt(D * D + x * x);
    const double t1 = s.sec + v.sec;
    double cos_t = hb * t2 / t1;
    cos_t = ref_max(-1.0, cos_t);
    cos_t = ref_min(1.0, cos_t);
    const double t = acos(cos_t);
    return ref_max(0.0, (t - sin(t) * cos_t) * t1 / PI);
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
};

// Restates the azimuth-independent parts of gortt_kg/gortt_kc/gortt_kc_fFbeta (gortt_brdf.c:7-238),
// gortt_set_zenith_dependant_probabilities (gortt.c:872-915) and gortt_kuusk (gortt_brdf.c:638-702).
__device__ void row_terms(const gort_canopy &c, double vza, double sza, RowTerms &r)
{
#pragma clang fp contract(off)
    const double ell = c.b / c.r;
    sincos(vza, &r.sin_vz, &r.cos_vz);
    sincos(sza, &r.sin_sz, &r.cos_sz);
    r.v = prime(ell, r.sin_vz / r.cos_vz);
    r.s = prime(ell, r.sin_sz / r.cos_sz);
    const Primed &v = r.v, &s = r.s;
    r.cov = c.lambda * PI * c.rr;                         // lambda pi r^2
    r.hb = c.h / c.b;
    r.t1 = s.sec + v.sec;
    const double cov = r.cov, t1 = r.t1;

    // principal-plane overlaps (Kc is interpolated between phi = 0 and pi, gortt_brdf.c:143-159)
    const double O_0 = overlap(r.hb, s, v, 1.0, 0.0);
    const double O_pi = overlap(r.hb, s, v, -1.0, 1.2246467991473532e-16);   // sin(M_PI) in double
    const double Kg0 = exp(-(cov * (t1 - O_0)));
    const double Kgpi = exp(-(cov * (t1 - O_pi)));

    const double xs = cov * s.sec, xv = cov * v.sec;
    r.es = exp(-xs);
    r.ev = exp(-xv);
    const double Mi = 1.0 - (1.0 - r.es) / xs;
    const double Mv = 1.0 - (1.0 - r.ev) / xv;
    const double theta_Mi = acos(1.0 - 2.0 * Mi);
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
        const double F = Gc / Gam;
        const double M = 1.0 - (1.0 - Kgq) / (c.lambda * Gam);
        const double PiMi = (1 - cos(theta_Mi * (1 - (s.ang - v.ang * cphi) / PI))) / 2.0;
        const double PvMv = Mv - (1.0 - cos(v.ang * cphi - s.ang)) / 2.0;
        // phi = pi lies in (90,270) deg -> Po = PvMv; phi = 0 -> by steepness (gortt_brdf.c:219-221)
        const double Po = (q == 1) ? PvMv : (view_steeper ? PiMi : PvMv);
        const double f = F * (1.0 - Gv * (PvMv + PiMi - Po) / Gc) / (1.0 - M);
        fF[q] = f * F;
    }
    r.fF0 = fF[0];
    r.fFpi = fF[1];

    if (c.use_user_beta) {
        r.beta = c.beta;
    } else if (s.ang < 0.000000001) {
        r.beta = 0.0;
    } else {
        const double Dd = c.r * (1.0 / tan(s.ang / 2.0));
        const double dh = (c.h2 - c.h1) / Dd;
        const double lg = c.lambda * Gv;
        r.beta = lg / (lg + dh) * (1.0 - exp(-lg - dh)) / (1.0 - exp(-lg));
    }

    // zenith-dependent gap probabilities; path lengths of Kuusk's hot spot
    double pn0_v;
    r.sun = sun_scalars(c, sza, r.cos_sz, s);
    r.eps_s = r.sun.eps;
    gap_lookup(c, vza, pn0_v, r.eps_v);
    r.kf = c.k * c.favd;
    r.ls = -log(r.eps_s) / r.kf;
    r.lv = -log(r.eps_v) / (0.5 * c.favd);
    r.h1 = (r.ls * r.lv) > 0.0 ? sqrt(r.ls * r.lv) : 0.0;

}

// The azimuth-dependent rest: overlap and Kg at the actual azimuth, the interpolated Kc, the other
// proportions (gortt.c:424-449) and the hot spot.
__device__ void finish_angle(const gort_canopy &c, const RowTerms &r, double raa, GeomOut &o)
{
#pragma clang fp contract(off)
    const Primed &v = r.v, &s = r.s;
    double sin_r, cos_r;
    sincos(raa, &sin_r, &cos_r);
    const double O_r = overlap(r.hb, s, v, cos_r, sin_r);
    const double Kg = exp(-(r.cov * (r.t1 - O_r)));
    const double ph_r = v.c * s.c + v.s * s.s * cos_r;
    const double F_r = (r.Gv * 0.5 * (1.0 + ph_r)) / (PI * c.rr * (r.t1 - O_r));

    double frac = raa / PI;
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
            const double lsv = sqrt(q2);
            h2 = (1.0 - exp(-lsv / c.r)) / (lsv / c.r);
        }
    }
    const double kuusk = r.eps_s * r.eps_v * exp(r.kf * r.h1 * h2);

    o.Kc = Kc;  o.Kg = Kg;  o.Kt = Kt;  o.Kz = Kz;  o.Kpg = Kpg;  o.Kpz = Kpz;
    o.A = kuusk / (2.0 * s.c * v.c);
    o.sun = r.sun;
}

// areal proportions + hot spot for one normalised geometry
__device__ void geometry_core(const gort_canopy &c, double vza, double sza, double raa, GeomOut &o)
{
    RowTerms r;
    row_terms(c, vza, sza, r);
    finish_angle(c, r, raa, o);
}

__device__ inline void store_coef(double *rec, const gort_canopy &c, const GeomOut &g)
{
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

// layout 0: the classic record (five coefficients, sun scalars, component-spectra extras: CoefSlot);
// layout 1: the LineTerms of the stream family's regrouped sample (gort_device.h) for the wide stream kernels, which
//           read them through the scalar cache once per 128 samples and must not spend VALU work on deriving them
// FUSED (few bands, no component spectra): the line's samples are formed right here from the record in registers -
// the same functions the narrow expansion kernels apply to the stored record, so the same bits - and no record is
// written: one launch instead of two (a 181-line x 1-band call is launch latency and nothing else), and for long
// narrow streams no 128 B of record written and read back per 8 B of result.
template <bool FUSED>
__global__ __launch_bounds__(256) void geometry_stream_kernel(const gort_canopy *__restrict__ canopy,
                                                               const double *__restrict__ angles, long nA,
                                                               double *__restrict__ coef, double *__restrict__ K,
                                                               int layout, const double *__restrict__ L, int nw,
                                                               double *__restrict__ rsurf)
{
    const long a = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nA) return;
    // blockIdx.z = ensemble member: its canopy, its nA records (the angle lines are shared)
    const long member = blockIdx.z;
    const gort_canopy &c = canopy[member];
    double vza, sza, saa, raa;
    normalise_angles(angles[4 * a], angles[4 * a + 1], angles[4 * a + 2], angles[4 * a + 3], vza, sza, saa, raa);
    GeomOut g;
    geometry_core(c, vza, sza, raa, g);
    if (FUSED) {
        double rec[GORT_COEF_STRIDE];
        store_coef(rec, c, g);
        const LineTerms l = line_terms_of_record(rec, c.k_openep, c.k_open);
        const double *__restrict__ Lm = L + member * L_NSLOT * nw;
        double *__restrict__ o = rsurf + (member * nA + a) * nw;
        for (int i = 0; i < nw; ++i) o[i] = stream_sample(l, stream_band(load_band(Lm, nw, i)));
    } else if (layout == 0) {
        store_coef(coef + (member * nA + a) * GORT_COEF_STRIDE, c, g);
    } else {
        double rec[GORT_COEF_STRIDE];
        store_coef(rec, c, g);
        const LineTerms l = line_terms_of_record(rec, c.k_openep, c.k_open);
        double *o = coef + (member * nA + a) * GORT_COEF_STRIDE;
        o[0] = l.alpha;  o[1] = l.P1;  o[2] = l.P2;  o[3] = l.Q1;  o[4] = l.Q2;  o[5] = l.Q3;  o[6] = l.Q4;  o[7] = l.Q5;
        o[8] = l.Q6;  o[9] = l.mu;  o[10] = l.t0;  o[11] = l.omtp0;  o[12] = l.m2;  o[13] = 0.0;  o[14] = 0.0;  o[15] = 0.0;
    }
    if (K) {
        double *k = K + 4 * (member * nA + a);
        k[0] = g.Kc;  k[1] = g.Kg;  k[2] = g.Kt;  k[3] = g.Kz;
    }
}

// grid nodes generated from indices; identical to streaming "vza phi sza 0" (SURVEY 8d, C3).
// One workgroup per GEOM_ROWS LUT rows, a row = (member, sun zenith, view zenith): the azimuth-independent terms
// of each row are evaluated ONCE into LDS, the rows side by side on the first lanes of one wavefront (the ~25
// transcendentals of a row are a serial chain: four rows cost the issue time of one), then the lanes walk the
// GEOM_ROWS x nphi azimuth nodes.  With 361 nodes per row this removes ~70 % of the transcendentals of the
// per-tuple form.
constexpr int GEOM_ROW_THREADS = 128;
constexpr int GEOM_ROWS = 4;
// mode 0: full stream records (GORT_COEF_STRIDE doubles per node); 1: compact 64-B records for the LUT kernel;
// 2: FUSED for grids of a few bands (BASELINE config 3 is one band): the node's samples are formed right here from
// its coefficients - no 128-B record per node written and read back (383 MB each way for the hemisphere grid,
// more than the arithmetic costs) - with the same sun_terms()/dot5() as the two-kernel path: same bits.
#ifndef GORT_GEOM_WAVES
#define GORT_GEOM_WAVES 3      // 168 VGPRs instead of 171: a third wave per SIMD, C3 133 -> 125 us; 4 would spill to scratch
#endif
__global__ __launch_bounds__(GEOM_ROW_THREADS) __attribute__((amdgpu_waves_per_eu(GORT_GEOM_WAVES)))
void geometry_grid_kernel(const gort_canopy *__restrict__ canopies,
                                                                          gort_grid g, long row_begin, long n_rows,
                                                                          double *__restrict__ coef, int compact,
                                                                          const double *__restrict__ Lall, int nw,
                                                                          double *__restrict__ rsurf)
{
    __shared__ RowTerms s_row[GEOM_ROWS];
    __shared__ int s_member[GEOM_ROWS];
    __shared__ double s_vza_deg[GEOM_ROWS], s_sza_deg[GEOM_ROWS];
    const long rows_per_member = (long)g.nsza * g.nvza;
    const long first = (long)blockIdx.x * GEOM_ROWS;                   // first row of this block, relative to row_begin
    const int rows_here = n_rows - first < GEOM_ROWS ? (int)(n_rows - first) : GEOM_ROWS;
    if ((int)threadIdx.x < rows_here) {
        const long grow = row_begin + first + threadIdx.x;
        const long member = grow / rows_per_member;
        const long row = grow - member * rows_per_member;
        const int isza = (int)(row / g.nvza), ivza = (int)(row % g.nvza);
        const double vza_deg = g.vza0 + ivza * g.dvza, sza_deg = g.sza0 + isza * g.dsza;
        double vza, sza, saa, raa;
        normalise_angles(vza_deg, g.phi0, sza_deg, 0.0, vza, sza, saa, raa);
        row_terms(canopies[member], vza, sza, s_row[threadIdx.x]);
        s_member[threadIdx.x] = (int)member;
        s_vza_deg[threadIdx.x] = vza_deg;
        s_sza_deg[threadIdx.x] = sza_deg;
    }
    __syncthreads();
    const int nodes = rows_here * g.nphi;
    for (int n = threadIdx.x; n < nodes; n += GEOM_ROW_THREADS) {
        const int r = n / g.nphi, l = n - r * g.nphi;
        const gort_canopy &c = canopies[s_member[r]];
        double vza, sza, saa, raa;
        normalise_angles(s_vza_deg[r], g.phi0 + l * g.dphi, s_sza_deg[r], 0.0, vza, sza, saa, raa);
        GeomOut o;
        finish_angle(c, s_row[r], raa, o);
        const long i = first * g.nphi + n;
        if (compact == 2) {
            double rec[GORT_COEF_STRIDE];
            store_coef(rec, c, o);
            const double *__restrict__ L = Lall + (long)s_member[r] * L_NSLOT * nw;
            const SunScalars sun = load_sun(rec);
            for (int b = 0; b < nw; ++b) {
                const SunTerms t = sun_terms(L, nw, b, sun, c.k_open, c.k_openep);
                rsurf[i * nw + b] = dot5(rec[A_C], rec[A_B], rec[A_Z], rec[A_G], rec[A_T], t.C0, t.B, t.Z, t.G, t.T);
            }
        } else if (compact) {
            // LUT path: only the five expansion coefficients, one 64-B record per node
            double rec[GORT_COEF_STRIDE];
            store_coef(rec, c, o);
            double2 *dst = reinterpret_cast<double2 *>(coef + i * 8);
            dst[0] = make_double2(rec[A_C], rec[A_B]);
            dst[1] = make_double2(rec[A_Z], rec[A_G]);
            dst[2] = make_double2(rec[A_T], 0.0);
            dst[3] = make_double2(0.0, 0.0);
        } else {
            store_coef(coef + i * GORT_COEF_STRIDE, c, o);
        }
    }
}

// ------------------------------------------------------- wavelength-only table

// Two-stream closed forms that depend on the band only (gortt_brdf.c:348-634 hoisted)
// blockIdx.y = ensemble member: canopy[m], spectra[m][3][nw] (rsoil, rleaf, tleaf) -> L[m][11][nw]
__global__ __launch_bounds__(256) void lambda_table_kernel(const gort_canopy *__restrict__ canopies, int nw,
                                                            const double *__restrict__ spectra,
                                                            double *__restrict__ Lall)
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
    const double tpff = tff * (1.0 - kopen) + kopen;
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
}


// ------------------------------------------------ stream expansion (any angles)

// one thread per (angle line, band); consecutive lanes = consecutive bands
template <bool WITH_SCOMP>
__global__ __launch_bounds__(256) void expand_stream_kernel(const gort_canopy *__restrict__ canopy,
                                                             const double *__restrict__ L, int nw,
                                                             const double *__restrict__ coef, long n_samples,
                                                             double *__restrict__ rsurf, double *__restrict__ scomp,
                                                             int grid_form)
{
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_samples) return;
    const long a = idx / nw;
    const int i = (int)(idx - a * nw);
    // blockIdx.z = ensemble member: canopy, band table, records and output of that member (n_samples each)
    const long member = blockIdx.z;
    canopy += member;
    L += member * L_NSLOT * nw;
    const double *rec = coef + a * GORT_COEF_STRIDE;
    if (member) {                                   // uniform branch; single-canopy launches skip the division
        rec += member * (n_samples / nw) * GORT_COEF_STRIDE;
        idx += member * n_samples;
    }
    const SunScalars s = load_sun(rec);
    const BandTerms t = load_band(L, nw, i);
    SunTerms b;
    if (WITH_SCOMP || grid_form) b = sun_terms(t, s, canopy->k_open, canopy->k_openep);
    // grid_form: a few-band LUT through this kernel belongs to the LUT family (five terms, dot5); streams use the
    // stream family's regrouped sample, like every other stream kernel
    if (grid_form) rsurf[idx] = dot5(rec[A_C], rec[A_B], rec[A_Z], rec[A_G], rec[A_T], b.C0, b.B, b.Z, b.G, b.T);
    else rsurf[idx] = stream_sample(line_terms_of_record(rec, canopy->k_openep, canopy->k_open), stream_band(t));
    if (WITH_SCOMP) {
        double4 o;
        o.x = b.C0 + rec[C_FDA] * b.B + rec[C_KPZ] * b.Z + rec[C_KPG] * b.G;    // C
        o.y = b.G;
        o.z = b.T;
        o.w = b.Z;
        reinterpret_cast<double4 *>(scomp)[idx] = o;
    }
}

// Band-major form for wide spectra: thread = band, keeps its 11 band terms in registers and walks
// STREAM_LINES angle lines; the line's record (5 coefficients + 6 sun scalars) is workgroup-uniform and
// comes through the scalar cache.  The flat form above re-reads 88 B of band terms per 8 B written.
constexpr int STREAM_LINES = 32;
template <bool WITH_SCOMP>
__global__ __launch_bounds__(256) void expand_stream_bands_kernel(const gort_canopy *__restrict__ canopy,
                                                                   const double *__restrict__ L, int nw,
                                                                   const double *__restrict__ coef, long nA,
                                                                   double *__restrict__ rsurf,
                                                                   double *__restrict__ scomp, int grid_form)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const long a0 = (long)blockIdx.y * STREAM_LINES;
    const long a1 = a0 + STREAM_LINES < nA ? a0 + STREAM_LINES : nA;
    const bool live = i < nw;
    const BandTerms t = load_band(L, nw, live ? i : 0);
    const StreamBand sb = stream_band(t);
    const double ko = canopy->k_open, kep = canopy->k_openep;
    for (long a = a0; a < a1; ++a) {
        const double *__restrict__ rec = coef + a * GORT_COEF_STRIDE;
        const SunScalars s = load_sun(rec);
        SunTerms b;
        if (WITH_SCOMP || grid_form) b = sun_terms(t, s, ko, kep);
        const double v = grid_form ? dot5(rec[A_C], rec[A_B], rec[A_Z], rec[A_G], rec[A_T], b.C0, b.B, b.Z, b.G, b.T)
                                   : stream_sample(line_terms_of_record(rec, kep, ko), sb);
        if (live) {
            rsurf[a * nw + i] = v;
            if (WITH_SCOMP) {
                double4 o;
                o.x = b.C0 + rec[C_FDA] * b.B + rec[C_KPZ] * b.Z + rec[C_KPG] * b.G;    // C
                o.y = b.G;
                o.z = b.T;
                o.w = b.Z;
                reinterpret_cast<double4 *>(scomp)[a * nw + i] = o;
            }
        }
    }
}

// ------------------------------------------------------------ LUT (grid) path

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

// The hot kernel.  One workgroup per (sun zenith, view zenith) row of the LUT.
//   * the row's angle coefficients (nphi x 8 doubles, written compactly by
//     geometry_grid_kernel) are staged once into LDS with 16-B coalesced loads, so the
//     azimuth loop never waits on HBM latency;
//   * thread `tid` owns bands tid, tid+THREADS, ... (NP passes) and keeps their five sun
//     terms in registers for the whole row;
//   * per azimuth node: 5 workgroup-uniform LDS reads (broadcast), 5 FMAs per band, and
//     8 B per lane / 512 contiguous bytes per wave-instruction streamed to
//     lut[row][phi][band] (optionally non-temporal).
constexpr int GRID_COEF_STRIDE = 8;     // compact record: A_C..A_T + 3 pad = 64 B
template <int THREADS, int NP, bool NT>
__global__ __launch_bounds__(THREADS) void expand_grid_kernel(const double *__restrict__ sun, int isza_base,
                                                               const double *__restrict__ coef, int nw, int nvza,
                                                               int nphi, long row_begin, double *__restrict__ lut)
{
    extern __shared__ double s_coef[];      // [nphi][GRID_COEF_STRIDE]
    const long row = row_begin + blockIdx.x;
    const int isza = (int)(row / nvza) - isza_base;
    const int tid = threadIdx.x;
    {
        const double2 *__restrict__ src =
            reinterpret_cast<const double2 *>(coef + (long)blockIdx.x * nphi * GRID_COEF_STRIDE);
        double2 *dst = reinterpret_cast<double2 *>(s_coef);
        for (int q = tid; q < nphi * (GRID_COEF_STRIDE / 2); q += THREADS) dst[q] = src[q];
    }
    const double *__restrict__ b = sun + (long)isza * 5 * nw;
    double bC[NP], bB[NP], bZ[NP], bG[NP], bT[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int i = tid + p * THREADS;
        const bool ok = i < nw;
        bC[p] = ok ? b[0 * nw + i] : 0.0;
        bB[p] = ok ? b[1 * nw + i] : 0.0;
        bZ[p] = ok ? b[2 * nw + i] : 0.0;
        bG[p] = ok ? b[3 * nw + i] : 0.0;
        bT[p] = ok ? b[4 * nw + i] : 0.0;
    }
    __syncthreads();
    double *__restrict__ out = lut + (long)blockIdx.x * nphi * nw;
    // launch_expand_grid picks NP = ceil(nw/THREADS): passes 0..NP-2 are full, only the last is ragged
    const bool last_ok = tid + (NP - 1) * THREADS < nw;
#pragma unroll 2
    for (int l = 0; l < nphi; ++l, out += nw) {
        const double *a = s_coef + l * GRID_COEF_STRIDE;
        const double aC = a[A_C], aB = a[A_B], aZ = a[A_Z], aG = a[A_G], aT = a[A_T];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = tid + p * THREADS;
            const double v = dot5(aC, aB, aZ, aG, aT, bC[p], bB[p], bZ[p], bG[p], bT[p]);
            if (p < NP - 1 || last_ok) {
                if (NT) __builtin_nontemporal_store(v, out + i);
                else out[i] = v;
            }
        }
    }
}

// The hot kernel, aligned form.  The LUT slab is one contiguous array of
// n_total = angles x nw doubles.  HBM3E on this part sustains its write rate only when
// every wave store covers whole 128-B lines: a plain 8-B-per-lane fill of the 50 GB slab
// runs at 6.5 TB/s when its wave stores are 512-B aligned and at 3.2 TB/s when they start
// 8 B off (tools/probes/store_probe.hip) - and with nw = 2101 (odd) any band-major mapping is
// 8-B aligned at best.  So the slab is cut into 1-KiB chunks (128 doubles) aligned in
// ABSOLUTE address, one chunk per wave-step (16 B per lane, one global_store_dwordx4),
// and a wave takes chunks c, c + W, c + 2W, ... where the stride W (in chunks) is a
// multiple of nw / gcd(nw, 128).  Then 128 W is a multiple of nw, so a lane keeps the
// SAME two bands for its whole life (their five (sun zenith, band) terms stay in
// registers, reloaded only when the lane's angle crosses into the next sun zenith) and
// advances its angle by da = 128 W / nw per step.
//
// The slab is worked through in PANELS of K steps x W waves: wave (panel, w) writes chunks
// panel K W + w + k W, k < K, so that a panel is K W consecutive chunks and each XCD's write
// window stays compact (K = 6, W = 2101: 12 MiB) instead of combing through the whole slab
// (7.1 against 7.9-9.5 ms for the 50 GB slab, and far less dependent on where the slab lies
// physically - DESIGN.md 5.1).  Waves are short-lived, hence the lean prologue below.
//
// Everything that moves per step is wave-uniform and lives in SGPRs: the chunk's output
// address and the address of the angle record, whose five coefficients arrive through
// the scalar cache (tools/probes/store_probe2.hip: one scalar record per 1-KiB step costs 2 %,
// per 512-B step 30 %).  In ~6 % of the waves (nw = 2101) the band index wraps inside
// the chunk, i.e. the chunk spans two angles: those waves fetch both records and each
// element picks its own.  The coefficient buffer carries one pad record in front and a
// tail pad so that the record prefetch needs no bounds logic.

// The steps of one wave.  What a lane carries is its sun terms b (20 VGPRs) and its bands; everything that
// moves is wave-uniform: the output chunk, the record address and the sun zenith of the chunk's first angle
// (isza_w, rem_w = angle index within the zenith).  An element is in the chunk's first angle or (`second`,
// only in WRAP waves) in the next one, so whether its sun zenith changes at a step follows from the scalar
// tracker: ch0 for first-angle elements, ch1 for second-angle ones - in the 94 % of waves without a wrap the
// reload of the sun terms is a wave-uniform branch.  Validity is also scalar but for the two edges of the
// slab: at step 0 of chunk 0 the elements below first_off lie in front of it, at step last_step (the slab's
// last chunk; -1 if this wave never gets there) those above last_off lie behind it.
template <int DEPTH, bool NT, bool WRAP>
__device__ __forceinline__ void flat_loop(double (&b)[EPL][5], const int (&band)[EPL], const bool (&second)[EPL],
                                          int isza_w, int rem_w, int first_off, int last_step, int last_off,
                                          const double *__restrict__ sun, int isza_base, int nw, int angles_per_sza,
                                          int da, long step, int k_wave, const double *__restrict__ rec_w,
                                          double *__restrict__ out_w, int lane)
{
    const long rec_step = (long)da * GRID_COEF_STRIDE;
    double rA[DEPTH][5], rB[WRAP ? DEPTH : 1][5];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const double *r = rec_w + (long)d * rec_step;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            rA[d][q] = r[q];
            if (WRAP) rB[d][q] = r[GRID_COEF_STRIDE + q];
        }
    }
    rec_w += (long)DEPTH * rec_step;
    for (int k = 0; k < k_wave; k += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kk = k + d;
            double v[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const double vA = dot5(rA[d][A_C], rA[d][A_B], rA[d][A_Z], rA[d][A_G], rA[d][A_T], b[j][0], b[j][1],
                                       b[j][2], b[j][3], b[j][4]);
                if (WRAP) {
                    const double vB = dot5(rB[d][A_C], rB[d][A_B], rB[d][A_Z], rB[d][A_G], rB[d][A_T], b[j][0], b[j][1],
                                           b[j][2], b[j][3], b[j][4]);
                    v[j] = second[j] ? vB : vA;
                } else {
                    v[j] = vA;
                }
            }
            if (kk < k_wave) {
                double *o = out_w + EPL * lane;
                const bool front = kk == 0 && first_off > 0, back = kk == last_step;
                if (!front && !back) {
                    dbl2 vv;
                    vv.x = v[0];
                    vv.y = v[1];
                    if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<dbl2 *>(o));
                    else *reinterpret_cast<dbl2 *>(o) = vv;
                } else {
#pragma unroll
                    for (int j = 0; j < EPL; ++j) {
                        const int off = EPL * lane + j;
                        if (!(front && off < first_off) && !(back && off > last_off)) o[j] = v[j];
                    }
                }
            }
            out_w += step;
            // refill this slot with the record(s) DEPTH steps ahead (the tail pad makes them always readable)
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                rA[d][q] = rec_w[q];
                if (WRAP) rB[d][q] = rec_w[GRID_COEF_STRIDE + q];
            }
            rec_w += rec_step;
            // next step's sun zenith (scalar), and the sun terms of the elements that cross into it
            const int isza_old = isza_w + ((WRAP && rem_w == angles_per_sza - 1) ? 1 : 0), isza_old0 = isza_w;
            rem_w += da;
            while (rem_w >= angles_per_sza) { rem_w -= angles_per_sza; ++isza_w; }
            const int isza_new = isza_w + ((WRAP && rem_w == angles_per_sza - 1) ? 1 : 0);
            const bool ch0 = isza_w != isza_old0, ch1 = WRAP && isza_new != isza_old;
            if (kk + 1 < k_wave && (ch0 || ch1)) {
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const bool sec = WRAP && second[j];
                    const bool behind = kk + 1 == last_step && EPL * lane + j > last_off;    // never stored again
                    if ((sec ? ch1 : ch0) && !behind) {
                        const double *bp = sun + (long)((sec ? isza_new : isza_w) - isza_base) * 5 * nw + band[j];
#pragma unroll
                        for (int q = 0; q < 5; ++q) b[j][q] = bp[(long)q * nw];
                    }
                }
            }
        }
    }
}

// 60 VGPRs but 106 SGPRs (the records of DEPTH steps live there): 7 waves/SIMD.  Forcing 8 with
// amdgpu_waves_per_eu spills 50 SGPRs into VGPR lanes and changes nothing (6.8 ms either way, ab_occ.log).
template <int DEPTH, bool NT>
__global__ __launch_bounds__(256) void expand_flat_kernel(const double *__restrict__ sun, int isza_base,
                                                           const double *__restrict__ coef, int nw,
                                                           int angles_per_sza, long angle0, long n_total, int shift,
                                                           long stride_chunks, int da, int steps_per_wave,
                                                           FastDiv div_stride, FastDiv div_nw, FastDiv div_aps,
                                                           double *__restrict__ lut, int xcd_mode,
                                                           XcdDuty duty, long useful_blocks,
                                                           int *__restrict__ xcd_slots)
{
    // A wave of a short panel lives for a few microseconds and the launch is bound by how many such lives fit
    // on a CU, not by HBM alone: every scalar-memory round trip in the prologue shows in the kernel's time.
    // Left alone the compiler fetches kernel arguments where they are first used, in five or six round trips;
    // naming them here makes it one batch of s_loads.
#define GORT_ARG_NOW(x) asm volatile("" ::"s"(x))
    GORT_ARG_NOW(nw);  GORT_ARG_NOW(angles_per_sza);  GORT_ARG_NOW(angle0);  GORT_ARG_NOW(n_total);  GORT_ARG_NOW(shift);
    GORT_ARG_NOW(stride_chunks);  GORT_ARG_NOW(da);  GORT_ARG_NOW(steps_per_wave);  GORT_ARG_NOW(xcd_mode);
    GORT_ARG_NOW(duty.w8);  GORT_ARG_NOW(duty.q);  GORT_ARG_NOW(useful_blocks);  GORT_ARG_NOW(isza_base);
    GORT_ARG_NOW(div_stride.mul);  GORT_ARG_NOW(div_stride.sh);  GORT_ARG_NOW(div_nw.mul);  GORT_ARG_NOW(div_nw.sh);
    GORT_ARG_NOW(div_aps.mul);  GORT_ARG_NOW(div_aps.sh);
    GORT_ARG_NOW(sun);  GORT_ARG_NOW(lut);      // not `coef`: naming it here costs 40 VGPRs
#undef GORT_ARG_NOW
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long block = xcd_logical_block(xcd_mode, duty, useful_blocks, xcd_slots);
    if (block < 0) return;
    // panels of steps_per_wave x stride chunks: wave (panel, w) takes chunks panel*K*stride + w + k*stride, k < K
    // All of the wave's index arithmetic is wave-uniform, 32-bit and free of run-time divisions (the host
    // checks that waves, chunks and 2 step stay below 2^31).
    const unsigned wave = (unsigned)(block * 4 + wave_in_block);              // scalar
    const unsigned stride = (unsigned)stride_chunks;
    const unsigned panel = fast_div(wave, div_stride);
    const unsigned w_in_panel = wave - panel * stride;
    const int lane = threadIdx.x & 63;
    const long step = stride_chunks * CHUNK;       // elements per step = da * nw
    // chunk index at step 0, counted from the aligned chunk that holds element 0 (the slab starts `shift`
    // elements into chunk 0), and the chunk / offset of the slab's last element
    const long c0 = (long)panel * steps_per_wave * stride_chunks + w_in_panel;
    const long last = n_total - 1 + shift;
    const long last_chunk = last / CHUNK;
    const int last_off = (int)(last % CHUNK);
    if (c0 > last_chunk) return;
    const long e0 = c0 * CHUNK - shift;            // element index of the chunk start at step 0 (< 0 only for chunk 0)
    // angle and band of the chunk start at step 0: a panel starts a whole number of angles into the slab,
    // and the rest, taken one step ahead to stay positive, is below 2 step
    const unsigned local = w_in_panel * CHUNK + (unsigned)step - (unsigned)shift;
    const unsigned a_loc = fast_div(local, div_nw);
    const int band_w = (int)(local - a_loc * (unsigned)nw);
    const long a_w = (long)panel * steps_per_wave * da + a_loc - da;          // scalar, >= -1
    // steps until the wave's chunk passes the end of the slab (only the last panel's waves run out)
    const long rel = last_chunk - c0;
    const bool runs_out = rel < (long)steps_per_wave * stride_chunks;
    int k_wave = steps_per_wave, last_step = -1;
    if (runs_out) {
        const unsigned k_last = fast_div((unsigned)rel, div_stride);
        k_wave = (int)k_last + 1;
        if ((unsigned)rel == k_last * stride) last_step = (int)k_last;        // ends in the slab's last chunk
    }
    // sun zenith of the angle a_w
    const long A0 = angle0 + a_w;                                             // >= -1
    int isza_w = -1, rem_w = angles_per_sza - 1;
    if (A0 >= 0) {
        const long q = A0 < (1L << 31) ? (long)fast_div((unsigned)A0, div_aps) : A0 / angles_per_sza;
        isza_w = (int)q;
        rem_w = (int)(A0 - q * angles_per_sza);
    }
    const int first_off = c0 == 0 ? shift : 0;

    double b[EPL][5];
    int band[EPL];
    bool second[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int off = EPL * lane + j;
        band[j] = band_w + off;
        second[j] = band[j] >= nw;                                            // nw >= CHUNK on this path: one wrap at most
        if (second[j]) band[j] -= nw;
        // sun zenith of this element's angle at step 0; rows outside the table belong to elements that are not
        // stored at step 0 (in front of the slab: the first crossing loads them; behind it: never)
        const int isza = isza_w + ((second[j] && rem_w == angles_per_sza - 1) ? 1 : 0);
        const bool live = isza >= isza_base && !(last_step == 0 && off > last_off);
        const double *bp = sun + (long)(isza - isza_base) * 5 * nw + band[j];
#pragma unroll
        for (int q = 0; q < 5; ++q) b[j][q] = live ? bp[(long)q * nw] : 0.0;
    }
    const double *rec_w = coef + a_w * GRID_COEF_STRIDE;                      // may point at the front pad record
    double *out_w = lut + e0;
    // a wave either never or always has its band wrap inside the chunk (bands are fixed per lane)
    if (band_w + CHUNK - 1 >= nw)
        flat_loop<DEPTH, NT, true>(b, band, second, isza_w, rem_w, first_off, last_step, last_off, sun, isza_base, nw,
                                   angles_per_sza, da, step, k_wave, rec_w, out_w, lane);
    else
        flat_loop<DEPTH, NT, false>(b, band, second, isza_w, rem_w, first_off, last_step, last_off, sun, isza_base, nw,
                                    angles_per_sza, da, step, k_wave, rec_w, out_w, lane);
}

// The aligned flat form for ARBITRARY angle lines (every line has its own sun zenith): the chunking, the
// band-preserving stride, the PANELS (K steps x W waves, each XCD one contiguous run of panels) and the slab-edge
// handling of expand_flat_kernel; but a lane keeps the 12 band constants of its two bands in registers and forms
// p_df, t'_df and the sample per step from the line's 13 LineTerms (records in layout 1, scalar loads, one step
// ahead): ~24 instructions + one fp64 division per sample (gort_device.h, stream family).
// coef: stream records (GORT_COEF_STRIDE doubles), one pad record in front, tail pad behind.
// one step of a wave: the samples of the chunk from the record(s) `rec`, stored with the slab-edge handling
template <bool NT, bool WRAP>
__device__ __forceinline__ void flat_stream_step(const StreamBand (&t)[EPL], const bool (&second)[EPL],
                                                 const double (&rec)[WRAP ? 2 : 1][LINE_NTERMS], bool front, bool back,
                                                 int first_off, int last_off, double *__restrict__ o, int lane)
{
    double v[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        double vv[2];
#pragma unroll
        for (int w = 0; w < (WRAP ? 2 : 1); ++w) {
            double pdf, tpdf;
            sun_pair(t[j], rec[w][9], rec[w][10], rec[w][11], rec[w][12], pdf, tpdf);
            vv[w] = stream_sample(rec[w][0], rec[w][1], rec[w][2], rec[w][3], rec[w][4], rec[w][5], rec[w][6], rec[w][7],
                                  rec[w][8], t[j], pdf, tpdf);
        }
        v[j] = (WRAP && second[j]) ? vv[1] : vv[0];
    }
    if (!front && !back) {
        dbl2 x;
        x.x = v[0];
        x.y = v[1];
        if (NT) __builtin_nontemporal_store(x, reinterpret_cast<dbl2 *>(o));
        else *reinterpret_cast<dbl2 *>(o) = x;
    } else {
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int off = EPL * lane + j;
            if (!(front && off < first_off) && !(back && off > last_off)) o[j] = v[j];
        }
    }
}

template <bool WRAP>
__device__ __forceinline__ void load_line_terms(double (&rec)[WRAP ? 2 : 1][LINE_NTERMS], const double *__restrict__ p)
{
#pragma unroll
    for (int q = 0; q < LINE_NTERMS; ++q) {
        rec[0][q] = p[q];
        if (WRAP) rec[WRAP ? 1 : 0][q] = p[GORT_COEF_STRIDE + q];
    }
}

// The steps of a wave.  One record set: the compiler issues the scalar loads of step k+1 behind the arithmetic of step
// k and the wave waits for them at the top of the next step; the other waves of the SIMD (6 at 78 VGPRs) fill that
// gap.  A hand-made double buffer (loads of step k+1 in front of the arithmetic of step k, two SGPR sets) cost a wave
// of occupancy and ran 30 % SLOWER (4.54 against 3.44 ms for 1 048 576 lines): this kernel lives on thread-level
// parallelism.
template <bool NT, bool WRAP>
__device__ __forceinline__ void flat_stream_loop(const StreamBand (&t)[EPL], const bool (&second)[EPL], int first_off,
                                                 int last_step, int last_off, int da, long step, int k_wave,
                                                 const double *__restrict__ rec_w, double *__restrict__ out_w, int lane)
{
    const long rec_step = (long)da * GORT_COEF_STRIDE;
    double *o = out_w + EPL * lane;
    double r[WRAP ? 2 : 1][LINE_NTERMS];
    for (int kk = 0; kk < k_wave; ++kk) {
        load_line_terms<WRAP>(r, rec_w);
        rec_w += rec_step;
        flat_stream_step<NT, WRAP>(t, second, r, kk == 0 && first_off > 0, kk == last_step, first_off, last_off, o, lane);
        o += step;
    }
}

// 70 VGPRs, 7 waves/SIMD.  Forcing 8 (amdgpu_waves_per_eu) spills 68 B per lane to scratch and halves the rate.
template <bool NT>
__global__ __launch_bounds__(256) void expand_flat_stream_kernel(const double *__restrict__ L, int nw,
                                                                  const double *__restrict__ coef, long n_total,
                                                                  int shift, long stride_chunks, int da,
                                                                  int steps_per_wave, FastDiv div_stride, FastDiv div_nw,
                                                                  double *__restrict__ out, int xcd_mode,
                                                                  XcdDuty duty, long useful_blocks,
                                                                  int *__restrict__ xcd_slots,
                                                                  const int *__restrict__ direct_flag)
{
    // behind the grouped form (gort_stream.hip) this kernel runs only when that one gave the stream up
    if (direct_flag && *direct_flag == -1) return;          // -1 = clear
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long block = xcd_logical_block(xcd_mode, duty, useful_blocks, xcd_slots);
    if (block < 0) return;
    // panels of steps_per_wave x stride chunks: wave (panel, w) takes chunks panel*K*stride + w + k*stride, k < K
    // (index arithmetic as in expand_flat_kernel: wave-uniform, 32-bit, divisions by multiply-shift)
    const unsigned wave = (unsigned)(block * 4 + wave_in_block);
    const unsigned stride = (unsigned)stride_chunks;
    const unsigned panel = fast_div(wave, div_stride);
    const unsigned w_in_panel = wave - panel * stride;
    const int lane = threadIdx.x & 63;
    const long step = stride_chunks * CHUNK;                 // elements per step = da * nw
    const long c0 = (long)panel * steps_per_wave * stride_chunks + w_in_panel;
    const long last = n_total - 1 + shift;
    const long last_chunk = last / CHUNK;
    const int last_off = (int)(last % CHUNK);
    if (c0 > last_chunk) return;
    const long e0 = c0 * CHUNK - shift;                      // element index of the chunk start at step 0 (< 0 only for chunk 0)
    const unsigned local = w_in_panel * CHUNK + (unsigned)step - (unsigned)shift;
    const unsigned a_loc = fast_div(local, div_nw);
    const int band_w = (int)(local - a_loc * (unsigned)nw);
    const long a_w = (long)panel * steps_per_wave * da + a_loc - da;          // line of the chunk start, >= -1
    const long rel = last_chunk - c0;
    int k_wave = steps_per_wave, last_step = -1;
    if (rel < (long)steps_per_wave * stride_chunks) {       // only the last panel's waves run out of slab
        const unsigned k_last = fast_div((unsigned)rel, div_stride);
        k_wave = (int)k_last + 1;
        if ((unsigned)rel == k_last * stride) last_step = (int)k_last;        // ends in the slab's last chunk
    }
    const int first_off = c0 == 0 ? shift : 0;
    StreamBand t[EPL];
    bool second[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        int band = band_w + EPL * lane + j;
        second[j] = band >= nw;                              // nw >= CHUNK on this path: one wrap at most
        if (second[j]) band -= nw;
        t[j] = stream_band(load_band(L, nw, band));
    }
    const double *rec_w = coef + a_w * GORT_COEF_STRIDE;     // may point at the front pad record
    double *out_w = out + e0;
    if (band_w + CHUNK - 1 >= nw)
        flat_stream_loop<NT, true>(t, second, first_off, last_step, last_off, da, step, k_wave, rec_w, out_w, lane);
    else
        flat_stream_loop<NT, false>(t, second, first_off, last_step, last_off, da, step, k_wave, rec_w, out_w, lane);
}

// ------------------------------------------------------------- albedo / energy

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

inline int check_launch(const char *what)
{
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(GORT_ENODEVICE, "%s: %s", what, hipGetErrorString(err));
    return GORT_OK;
}

}  // namespace

// ------------------------------------------------------------------- launchers

int launch_lambda_table(const gort_canopy *canopies_dev, int n_members, int nw, const double *spectra_dev,
                        double *L_dev, void *stream)
{
    if (nw <= 0 || n_members <= 0) return GORT_OK;
    hipLaunchKernelGGL(lambda_table_kernel, dim3((nw + 255) / 256, n_members), dim3(256), 0, (hipStream_t)stream,
                       canopies_dev, nw, spectra_dev, L_dev);
    return check_launch("lambda_table_kernel");
}

int launch_geometry_stream(const gort_canopy *canopy_dev, const double *angles_dev, long nA, double *coef_dev,
                           double *K_dev, int layout, void *stream)
{
    if (nA <= 0) return GORT_OK;
    hipLaunchKernelGGL(geometry_stream_kernel<false>, dim3((unsigned)((nA + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       canopy_dev, angles_dev, nA, coef_dev, K_dev, layout, (const double *)nullptr, 0, (double *)nullptr);
    return check_launch("geometry_stream_kernel");
}

// few bands, no component spectra: one fused launch (GORT_STREAM_FUSE=0 keeps the two-kernel path, for tests)
bool stream_fuses(int nw, bool want_scomp)
{
    const char *v = getenv("GORT_STREAM_FUSE");             // read per call: the tests switch it inside one process
    return !(v && atoi(v) == 0) && !want_scomp && nw > 0 && nw <= 16;
}

// geometry and samples of a few-band stream in one launch (no records): rsurf_dev[nA][nw]
int launch_geometry_stream_fused(const gort_canopy *canopy_dev, const double *L_dev, int nw, const double *angles_dev,
                                 long nA, double *rsurf_dev, double *K_dev, void *stream)
{
    if (nA <= 0 || nw < 0 || (nw == 0 && !K_dev)) return GORT_OK;      // nw = 0: the proportions K alone
    hipLaunchKernelGGL(geometry_stream_kernel<true>, dim3((unsigned)((nA + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       canopy_dev, angles_dev, nA, (double *)nullptr, K_dev, 0, L_dev, nw, rsurf_dev);
    return check_launch("geometry_stream_kernel<fused>");
}

int launch_geometry_grid(const gort_canopy *canopy_dev, const gort_grid &g, long row_begin, long row_end,
                         double *coef_dev, bool compact, void *stream)
{
    const long rows = row_end - row_begin;
    if (rows <= 0) return GORT_OK;
    hipLaunchKernelGGL(geometry_grid_kernel, dim3((unsigned)((rows + GEOM_ROWS - 1) / GEOM_ROWS)), dim3(GEOM_ROW_THREADS), 0,
                       (hipStream_t)stream, canopy_dev, g, row_begin, rows, coef_dev, compact ? 1 : 0,
                       (const double *)nullptr, 0, (double *)nullptr);
    return check_launch("geometry_grid_kernel");
}

int launch_geometry_grid_fused(const gort_canopy *canopy_dev, const double *L_dev, int nw, const gort_grid &g, long row_begin,
                               long row_end, double *rsurf_dev, void *stream)
{
    const long rows = row_end - row_begin;
    if (rows <= 0 || nw <= 0) return GORT_OK;
    hipLaunchKernelGGL(geometry_grid_kernel, dim3((unsigned)((rows + GEOM_ROWS - 1) / GEOM_ROWS)), dim3(GEOM_ROW_THREADS), 0,
                       (hipStream_t)stream, canopy_dev, g, row_begin, rows, (double *)nullptr, 2, L_dev, nw, rsurf_dev);
    return check_launch("geometry_grid_kernel (fused)");
}

// nA angle lines for each of n_members members (member-major records and outputs); one thread per sample
int launch_members_stream(const gort_canopy *canopies_dev, int n_members, const double *L_dev, int nw,
                          const double *angles_dev, long nA, double *coef_dev, double *rsurf_dev, void *stream)
{
    const long n = nA * nw;
    if (n <= 0 || n_members <= 0) return GORT_OK;
    if (n_members > 65535) return fail(GORT_EINVAL, "members stream: %d members in one launch (max 65535)", n_members);
    hipStream_t s = (hipStream_t)stream;
    if (stream_fuses(nw, false)) {
        hipLaunchKernelGGL(geometry_stream_kernel<true>, dim3((unsigned)((nA + 255) / 256), 1, (unsigned)n_members), dim3(256), 0, s,
                           canopies_dev, angles_dev, nA, (double *)nullptr, (double *)nullptr, 0, L_dev, nw, rsurf_dev);
        return check_launch("geometry_stream_kernel<fused>");
    }
    hipLaunchKernelGGL(geometry_stream_kernel<false>, dim3((unsigned)((nA + 255) / 256), 1, (unsigned)n_members), dim3(256), 0, s,
                       canopies_dev, angles_dev, nA, coef_dev, (double *)nullptr, 0, (const double *)nullptr, 0, (double *)nullptr);
    int rc = check_launch("geometry_stream_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(expand_stream_kernel<false>, dim3((unsigned)((n + 255) / 256), 1, (unsigned)n_members), dim3(256), 0,
                       s, canopies_dev, L_dev, nw, coef_dev, n, rsurf_dev, (double *)nullptr, 0);
    return check_launch("expand_stream_kernel");
}

static bool stream_uses_flat(int nw, long nA, bool want_scomp);
static int launch_expand_stream_flat(const gort_canopy *canopy_dev, const double *L_dev, int nw,
                                     const double *coef_dev, long nA, double *rsurf_dev, int *xcd_slots_dev,
                                     const int *direct_flag_dev, hipStream_t s);

bool expand_stream_workspace(int nw, long nA, bool want_scomp, size_t *ws_bytes, size_t *sun_bytes)
{
    *ws_bytes = 0;
    *sun_bytes = 0;
    if (!stream_uses_flat(nw, nA, want_scomp)) return false;
    stream_group_workspace(nw, nA, ws_bytes, sun_bytes);
    return true;
}

int launch_expand_stream(const gort_canopy *canopy_dev, const double *L_dev, int nw, const double *angles_dev,
                         const double *coef_dev, long nA, double *rsurf_dev, double *scomp_dev, int *xcd_slots_dev,
                         void *group_ws_dev, double *group_sun_dev, void *stream, void *coef_ready_event, bool grid_form)
{
    const long n = nA * nw;
    if (n <= 0) return GORT_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool grouped = !grid_form && stream_uses_flat(nw, nA, scomp_dev != nullptr) && angles_dev && group_ws_dev && group_sun_dev;
    if (coef_ready_event && !grouped &&
        hipStreamWaitEvent(s, (hipEvent_t)coef_ready_event, 0) != hipSuccess)
        return fail(GORT_ENODEVICE, "stream expansion: cannot wait for the geometry kernel");
    if (!grid_form && stream_uses_flat(nw, nA, scomp_dev != nullptr)) {
        // wide streams (records in layout 1): lines grouped by sun zenith where the stream allows it, per-line sun terms otherwise
        const int *direct_flag = nullptr;
        if (grouped) {
            const int rc = launch_expand_stream_grouped(canopy_dev, L_dev, nw, angles_dev, coef_dev, nA, rsurf_dev,
                                                        group_ws_dev, group_sun_dev, stream, coef_ready_event, &direct_flag);
            if (rc) return rc;
        }
        return launch_expand_stream_flat(canopy_dev, L_dev, nw, coef_dev, nA, rsurf_dev, xcd_slots_dev, direct_flag, s);
    }
    const long groups = (nA + STREAM_LINES - 1) / STREAM_LINES;
    if (nw >= 64 && groups <= 65535) {
        const dim3 grid((unsigned)((nw + 255) / 256), (unsigned)groups), block(256);
        if (scomp_dev)
            hipLaunchKernelGGL(expand_stream_bands_kernel<true>, grid, block, 0, s, canopy_dev, L_dev, nw, coef_dev, nA,
                               rsurf_dev, scomp_dev, grid_form ? 1 : 0);
        else
            hipLaunchKernelGGL(expand_stream_bands_kernel<false>, grid, block, 0, s, canopy_dev, L_dev, nw, coef_dev,
                               nA, rsurf_dev, scomp_dev, grid_form ? 1 : 0);
        return check_launch("expand_stream_bands_kernel");
    }
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (scomp_dev)
        hipLaunchKernelGGL(expand_stream_kernel<true>, grid, block, 0, s, canopy_dev, L_dev, nw, coef_dev, n, rsurf_dev,
                           scomp_dev, grid_form ? 1 : 0);
    else
        hipLaunchKernelGGL(expand_stream_kernel<false>, grid, block, 0, s, canopy_dev, L_dev, nw, coef_dev, n, rsurf_dev,
                           scomp_dev, grid_form ? 1 : 0);
    return check_launch("expand_stream_kernel");
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

// Tuning knobs for the LUT expansion (read once):
//   GORT_EXPAND_VARIANT  flat (default) | row     kernel form, see the two kernels above
//   GORT_EXPAND_DEPTH    1 | 2 | 4                coefficient records in flight per lane (flat)
//   GORT_EXPAND_NT       1 | 0                    non-temporal stores
//   GORT_EXPAND_WAVES    target wave stride of expand_flat_kernel in chunks = waves per panel (rounded to a
//                        band-preserving multiple of nw/gcd(nw,128))
//   GORT_EXPAND_STEPS    steps per wave = panel height; 0 = one panel, every wave strides through the whole slab
//   GORT_EXPAND_XCD      0 | 1 | 2                XCD mapping, see below; default automatic
//   GORT_STREAM_WAVES    waves per panel of expand_flat_stream_kernel; GORT_STREAM_STEPS its steps per wave
// Measured on the 50.25 GB metric slab, four slabs held at once per run (profiles/r01/tune_panels*.log):
//   whole-slab strides (steps 0, stride 33616)   8.1-8.3 ms, 9.5 ms on some allocations
//   panels of 6 steps x 2101 waves, XCD mode 1   7.04-7.30 ms (6.9-7.1 TB/s), 7.8 ms on some allocations
//   the same with 4 steps                        7.8 ms on every allocation (prologue-bound)
//   the same, XCD mode 0 (interleaved)           8.2 ms
//   XCD mode 2 with panels                       one returning atomic per workgroup costs ~190 ns on its
//                                                counter's line: 9.2 ms at 16 steps before the counters were
//                                                spread over 8 lines, then fine from 8 steps up
// The spread between allocations of one size is a property of where the slab lies physically (the same slab
// is slow or fast at any offset and for the whole run); short panels narrow it from 18 % to 10 %.
struct ExpandTuning {
    bool flat = true, nt = true;
    // GORT_EXPAND_XCD: 0 = logical blocks interleaved over the XCDs, 1 = one contiguous range per XCD assuming
    // round-robin dispatch, 2 = the same by the real XCC_ID through per-XCD slot counters;
    // -1 = automatic: 1 where dispatch is round-robin over the XCDs (probed once per engine), else 2
    int xcd_mode = -1;
    int depth = 2;
    int steps = -1;             // -1 = automatic: 6 with the static mapping, 16 with slot counters (fewer atomics)
    long waves = 2048;
    // the per-line stream kernel is VALU bound with a heavy prologue (24 band constants per lane): long waves.  Panels
    // of 64 steps x 33616 waves (2.2 GB per panel; streams up to 131 072 lines are ONE panel): 65 536 lines 232-290 us
    // whatever the shape, 1 048 576 lines 3.76 ms against 3.83 (one panel) and 4.0-5.0 (8..16 steps), tools/probes/perline_sweep.sh
    long stream_waves = 16808;      // GORT_STREAM_WAVES: waves per panel of the per-line stream kernel (rounded like `waves`)
    int stream_steps = 64;          // GORT_STREAM_STEPS: steps per wave = panel height of the per-line stream kernel
    ExpandTuning()
    {
        if (const char *v = getenv("GORT_EXPAND_VARIANT")) flat = strcmp(v, "row") != 0;
        if (const char *v = getenv("GORT_EXPAND_NT")) nt = atoi(v) != 0;
        if (const char *v = getenv("GORT_EXPAND_DEPTH")) depth = atoi(v);
        if (const char *v = getenv("GORT_EXPAND_WAVES")) waves = atol(v);
        if (const char *v = getenv("GORT_STREAM_WAVES")) stream_waves = atol(v);
        if (const char *v = getenv("GORT_STREAM_STEPS")) stream_steps = atoi(v);
        if (stream_steps < 1) stream_steps = 1;
        if (const char *v = getenv("GORT_EXPAND_XCD")) xcd_mode = atoi(v);
        if (const char *v = getenv("GORT_EXPAND_STEPS")) steps = atoi(v);
        if (xcd_mode < -1 || xcd_mode > 2) xcd_mode = -1;
        if (depth != 1 && depth != 2 && depth != 4) depth = 2;
        if (steps > 0) steps = (steps + depth - 1) / depth * depth;        // whole groups of DEPTH
        if (waves < 64) waves = 64;
        if (stream_waves < 64) stream_waves = 64;
    }
};

template <bool NT>
static int launch_expand_rows(int np, dim3 grid, size_t lds, hipStream_t s, const double *sun_dev, int isza_base,
                              const double *coef_dev, int nw, int nvza, int nphi, long row_begin, double *lut_dev)
{
#define GORT_EXPAND_CASE(N)                                                                                    \
    case N:                                                                                                    \
        hipLaunchKernelGGL((expand_grid_kernel<256, N, NT>), grid, dim3(256), lds, s, sun_dev, isza_base,     \
                           coef_dev, nw, nvza, nphi, row_begin, lut_dev);                                      \
        return GORT_OK;
    switch (np) {
        GORT_EXPAND_CASE(1)
        GORT_EXPAND_CASE(2)
        GORT_EXPAND_CASE(3)
        GORT_EXPAND_CASE(4)
        GORT_EXPAND_CASE(5)
        GORT_EXPAND_CASE(6)
        GORT_EXPAND_CASE(7)
        GORT_EXPAND_CASE(8)
        GORT_EXPAND_CASE(9)
    default:
        return fail(GORT_EINVAL, "expand_grid: nw=%d needs %d passes of 256 threads (max 9)", nw, np);
    }
#undef GORT_EXPAND_CASE
}

static long gcd_long(long a, long b)
{
    while (b) { const long t = a % b; a = b; b = t; }
    return a;
}

static const ExpandTuning &tuning()
{
    static const ExpandTuning t;
    return t;
}

// XCD mapping of a flat launch: without slot counters only the static forms are possible
static int resolve_xcd_mode(const int *xcd_slots_dev)
{
    const int m = tuning().xcd_mode;
    if (m < 0) return xcd_slots_dev ? 2 : 1;
    return (m == 2 && !xcd_slots_dev) ? 1 : m;
}

bool expand_wants_xcd_slots(bool dispatch_round_robin)
{
    const int m = tuning().xcd_mode;
    return m == 2 || (m < 0 && !dispatch_round_robin);
}

static FastDiv make_fast_div(unsigned d)
{
    FastDiv f;
    f.sh = 0;
    while ((1ull << f.sh) < d) ++f.sh;
    f.mul = (unsigned)((1ull << (31 + f.sh)) / d + 1);
    return f;
}

// Grid and ranges of a flat launch over `useful` logical blocks.  Mode 1: XCD x owns q w[x] blocks, q =
// ceil(useful / sum w), and gets through them in 32 q workgroup slots whatever its weight (the overshoot of at
// most sum w blocks falls off the end of the last range).
static long plan_xcd_duty(int xcd_mode, long useful, const int *weights, XcdDuty &duty)
{
    long sumw = 0;
    duty.w8 = 0;
    for (int x = 0; x < 8; ++x) {
        int w = weights ? weights[x] : 32;
        w = w < 8 ? 8 : (w > 32 ? 32 : w);
        duty.w8 |= (unsigned long long)w << (8 * x);
        sumw += w;
    }
    duty.q = (useful + sumw - 1) / sumw;
    return xcd_mode == 1 ? 8 * 32 * duty.q : useful;
}

// Host-side check of the index arithmetic the flat kernels rely on (no GPU needed; tests/test_host_abi.py):
// fast_div against '/', and the duty mapping as a bijection of the launch's workgroups onto the logical blocks.
// Returns 0, or the line of the first failed check.
int selftest_index_math()
{
    unsigned long long rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    const unsigned divisors[] = {1, 2, 3, 7, 128, 361, 2101, 2100, 4202, 32851, 65535, 65536, 1000003, 0x7fffffffu};
    for (unsigned d : divisors) {
        const FastDiv f = make_fast_div(d);
        const unsigned edge[] = {0u, 1u, d - 1, d, d + 1, 2 * d - 1 < 0x7fffffffu ? 2 * d - 1 : 0u, 0x7fffffffu, 0x7ffffffeu};
        for (unsigned n : edge) if (n <= 0x7fffffffu && fast_div(n, f) != n / d) return __LINE__;
        for (int k = 0; k < 200000; ++k) {
            const unsigned n = (unsigned)(next() >> 33);
            if (fast_div(n, f) != n / d) return __LINE__;
        }
    }
    for (int trial = 0; trial < 300; ++trial) {
        int w[8];
        for (int x = 0; x < 8; ++x) w[x] = trial == 0 ? 32 : (trial == 1 ? (x & 1 ? 27 : 32) : 8 + (int)(next() % 25));
        const long useful = trial < 2 ? 2044799 / (trial + 1) / 100 : 1 + (long)(next() % 20000);
        XcdDuty duty;
        const long grid = plan_xcd_duty(1, useful, w, duty);
        std::vector<unsigned char> seen((size_t)useful, 0);
        long hit = 0;
        for (long b = 0; b < grid; ++b) {
            const long blk = duty_logical_block(b, duty, useful);
            if (blk < 0) continue;
            if (blk >= useful || seen[(size_t)blk]) return __LINE__;
            seen[(size_t)blk] = 1;
            ++hit;
        }
        if (hit != useful) return __LINE__;
        XcdDuty plain;
        if (plan_xcd_duty(0, useful, w, plain) != useful || plan_xcd_duty(2, useful, nullptr, plain) != useful) return __LINE__;
    }
    return 0;
}

namespace {
// the store pattern of expand_flat_kernel (panels of K x W chunks, 16-B non-temporal stores) with the XCD
// mapping of mode 1; one workgroup in 64 reports when it started and ended
__global__ __launch_bounds__(256) void xcd_pattern_kernel(double *__restrict__ slab, long chunks, int K, unsigned W,
                                                          XcdDuty duty, long useful,
                                                          unsigned long long *__restrict__ t_first,
                                                          unsigned long long *__restrict__ t_last)
{
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long block = xcd_logical_block(1, duty, useful, nullptr);
    if (block < 0) return;
    const int xd = blockIdx.x & 7;
    const bool reports = t_first && (block & 63) == 0;
    if (reports && threadIdx.x == 0) atomicMin(&t_first[xd], wall_clock64());
    const unsigned wave = (unsigned)(block * 4 + wib);
    const unsigned panel = wave / W, w = wave - panel * W;
    const long c0 = (long)panel * K * W + w;
    const int lane = threadIdx.x & 63;
    dbl2 v;
    v.x = 0.0;
    v.y = 0.0;
    for (int k = 0; k < K; ++k) {
        const long c = c0 + (long)k * W;
        if (c < chunks) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(slab + c * CHUNK + EPL * lane));
    }
    __syncthreads();
    if (reports && threadIdx.x == 0) atomicMax(&t_last[xd], wall_clock64());
}
}  // namespace

int calibrate_xcd_weights(void *stream, double *slab, long n_doubles, int weights[8], double *pattern_gbs)
{
    if (pattern_gbs) *pattern_gbs = 0.0;
    for (int x = 0; x < 8; ++x) weights[x] = 32;
    // whole aligned chunks inside the slab
    const uintptr_t addr = reinterpret_cast<uintptr_t>(slab);
    const long skip = (long)(((addr + 1023) & ~(uintptr_t)1023) - addr) / 8;
    const long chunks = (n_doubles - skip) / CHUNK;
    constexpr int K = 6;
    constexpr unsigned W = 2101;
    if (chunks < 64L * K * W) return GORT_OK;                  // too small to say anything
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *dev = nullptr;
    if (hipMalloc(&dev, 16 * sizeof(*dev)) != hipSuccess) return fail(GORT_ENOMEM, "xcd calibration: hipMalloc failed");
    const long panels = (chunks + (long)K * W - 1) / ((long)K * W);
    const long useful = (panels * W + 3) / 4;
    int rc = GORT_OK;
    // ONE pass with equal weights: the rates of the XCDs while all of them run.  Iterating on the result drives
    // the slow XCDs' weights further down (to 25/32), which suits this bare store pattern but not the LUT kernel
    // (tools/probes/weights_sweep.py: 27-28 is its optimum, 25 already loses half the gain).
    for (int iter = 0; iter < 1 && rc == GORT_OK; ++iter) {
        XcdDuty duty;
        const long grid = plan_xcd_duty(1, useful, weights, duty);
        unsigned long long host[16];
        for (int x = 0; x < 8; ++x) { host[x] = ~0ull; host[8 + x] = 0; }
        hipError_t err = hipMemcpyAsync(dev, host, sizeof(host), hipMemcpyHostToDevice, s);
        if (err == hipSuccess) {
            hipLaunchKernelGGL(xcd_pattern_kernel, dim3((unsigned)grid), dim3(256), 0, s, slab + skip, chunks, K, W, duty,
                               useful, dev, dev + 8);
            err = hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, s);
        }
        if (err == hipSuccess) err = hipStreamSynchronize(s);
        if (err != hipSuccess) { rc = fail(GORT_ENODEVICE, "xcd calibration: %s", hipGetErrorString(err)); break; }
        unsigned long long t0 = ~0ull;
        for (int x = 0; x < 8; ++x) if (host[x] < t0) t0 = host[x];
        // blocks per tick of every XCD in this run; the next weights are proportional to it
        double rate[8], rmax = 0.0;
        bool ok = true;
        for (int x = 0; x < 8; ++x) {
            if (host[8 + x] <= t0) { ok = false; break; }
            rate[x] = (double)weights[x] / (double)(host[8 + x] - t0);
            if (rate[x] > rmax) rmax = rate[x];
        }
        if (!ok) break;                                        // an XCD reported nothing: leave the weights alone
        if (pattern_gbs) {
            // the rate of the bare store pattern with equal shares, first start to last end (wall clock ticks)
            unsigned long long t1 = 0;
            for (int x = 0; x < 8; ++x) if (host[8 + x] > t1) t1 = host[8 + x];
            int khz = 0, dev = 0;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0 && t1 > t0)
                *pattern_gbs = (double)chunks * 1024.0 / ((double)(t1 - t0) / khz * 1e-3) / 1e9;
        }
        for (int x = 0; x < 8; ++x) {
            int w = (int)(32.0 * rate[x] / rmax + 0.5);
            weights[x] = w < 16 ? 16 : (w > 32 ? 32 : w);
        }
    }
    // What is being measured is a trait of the device - the XCDs of one XCC_ID parity write ~15 % slower than the
    // others on every MI355X seen so far (the odd XCC_IDs in every standalone probe, the even dispatch slots in one process) - under a few % of
    // run-to-run noise, and a weight that is off by one
    // costs more than it gains (tools/probes/weights_sweep.py).  So the eight results are averaged within each parity.
    if (rc == GORT_OK) {
        double mean[2] = {0.0, 0.0};
        for (int x = 0; x < 8; ++x) mean[x & 1] += 0.25 * weights[x];
        const double top = mean[0] > mean[1] ? mean[0] : mean[1];
        for (int x = 0; x < 8; ++x) {
            int w = (int)(32.0 * mean[x & 1] / top + 0.5);
            weights[x] = w < 16 ? 16 : (w > 32 ? 32 : w);
        }
    }
    (void)hipFree(dev);
    return rc;
}

// Are workgroups b, b+8, b+16, ... of a launch placed on one XCD each (round-robin dispatch, the documented
// behaviour of the multi-XCD dispatcher)?  Then the static mapping of the flat kernels is exact and needs no
// atomics.  A profiler or a partition mode may change the pattern, hence the probe rather than an assumption.
namespace {
__global__ void xcd_probe_kernel(int *__restrict__ xcc_of_block)
{
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc_of_block[blockIdx.x] = (int)(x & 7);
    }
}
}  // namespace

int probe_xcd_dispatch(void *stream, int *round_robin)
{
    constexpr int NB = 4096;
    *round_robin = 0;
    int *dev = nullptr;
    if (hipMalloc(&dev, sizeof(int) * NB) != hipSuccess) return fail(GORT_ENOMEM, "xcd probe: hipMalloc failed");
    int host[NB];
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(xcd_probe_kernel, dim3(NB), dim3(256), 0, s, dev);
    hipError_t err = hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, s);
    if (err == hipSuccess) err = hipStreamSynchronize(s);
    (void)hipFree(dev);
    if (err != hipSuccess) return fail(GORT_ENODEVICE, "xcd probe: %s", hipGetErrorString(err));
    unsigned seen = 0;
    for (int b = 0; b < 8; ++b) seen |= 1u << host[b];
    bool ok = seen == 0xffu;
    for (int b = 8; b < NB && ok; ++b) ok = host[b] == host[b & 7];
    *round_robin = ok ? 1 : 0;
    return GORT_OK;
}

// chunk stride (in 1-KiB chunks) of the flat kernel: a multiple of nw/gcd(nw,CHUNK) close to the wave target
static long flat_stride(int nw, long chunks, long target)
{
    const long unit = nw / gcd_long(nw, CHUNK);
    long mult = (target + unit / 2) / unit;
    if (mult < 1) mult = 1;
    long stride = unit * mult;
    if (stride > chunks) stride = unit * ((chunks + unit - 1) / unit);       // tiny slab: one step per wave
    return stride;
}

// records the LUT kernel may read past the last angle (prefetch depth x angles per step, + wrap, + slack)
long expand_grid_tail_pad_records(int nw, long n_total)
{
    if (!tuning().flat) return 0;
    const long stride = flat_stride(nw, (n_total + 2 * CHUNK - 2) / CHUNK, tuning().waves);
    const long da = stride * CHUNK / nw;
    return 12 * da + 9;     // the k loop runs in groups of DEPTH <= 4 and prefetches DEPTH steps ahead (+1: wrap record)
}

int launch_expand_grid(const double *sun_dev, int isza_base, const double *coef_dev, int nw, int nvza, int nphi,
                       long row_begin, long row_end, double *lut_dev, int *xcd_slots_dev, const int *xcd_weights,
                       void *stream)
{
    const long rows = row_end - row_begin;
    if (rows <= 0) return GORT_OK;
    const ExpandTuning &tune = tuning();
    hipStream_t s = (hipStream_t)stream;
    if (!tune.flat) {
        const int np = (nw + 255) / 256;
        const size_t lds = sizeof(double) * GRID_COEF_STRIDE * (size_t)nphi;
        if (lds > 64 * 1024) return fail(GORT_EINVAL, "expand_grid: nphi=%d too large for the LDS staging buffer", nphi);
        const dim3 grid((unsigned)rows);
        int rc = tune.nt ? launch_expand_rows<true>(np, grid, lds, s, sun_dev, isza_base, coef_dev, nw, nvza, nphi, row_begin, lut_dev)
                         : launch_expand_rows<false>(np, grid, lds, s, sun_dev, isza_base, coef_dev, nw, nvza, nphi, row_begin, lut_dev);
        if (rc) return rc;
        return check_launch("expand_grid_kernel");
    }
    // flat, absolutely aligned form
    const long n_total = rows * nphi * (long)nw;
    const int shift = (int)((reinterpret_cast<uintptr_t>(lut_dev) / sizeof(double)) % CHUNK);
    const long chunks = (n_total + shift + CHUNK - 1) / CHUNK;
    const long stride = flat_stride(nw, chunks, tune.waves);
    const int xcd_mode = resolve_xcd_mode(xcd_slots_dev);
    // panel height: short panels keep the eight write windows compact; with slot counters every workgroup
    // pays a returning atomic, so there the panels are taller (fewer workgroups)
    int steps = tune.steps;
    if (steps < 0) steps = xcd_mode == 2 ? 16 : (6 + tune.depth - 1) / tune.depth * tune.depth;
    const long panels = steps > 0 ? (chunks + (long)steps * stride - 1) / ((long)steps * stride) : 1;
    if (steps == 0) steps = 1 << 30;
    if (panels * stride >= (1L << 31) || chunks >= (1L << 31) || stride * CHUNK >= (1L << 30))
        return fail(GORT_EINVAL, "expand_grid: slab of %ld chunks in %ld waves is beyond the kernel's 32-bit indices",
                    chunks, panels * stride);
    const int da = (int)(stride * CHUNK / nw);          // angles per step (the stride is a multiple of nw/gcd(nw,CHUNK))
    const long useful = (panels * stride + 3) / 4;
    XcdDuty duty;
    const long nblocks = plan_xcd_duty(xcd_mode, useful, xcd_weights, duty);
    if (nblocks >= (1L << 31)) return fail(GORT_EINVAL, "expand_grid: %ld workgroups in one launch", nblocks);
    const dim3 grid((unsigned)nblocks);
    const int angles_per_sza = nvza * nphi;
    const long angle0 = row_begin * nphi;
#define GORT_FLAT(D, N)                                                                                           \
    hipLaunchKernelGGL((expand_flat_kernel<D, N>), grid, dim3(256), 0, s, sun_dev, isza_base, coef_dev, nw,      \
                       angles_per_sza, angle0, n_total, shift, stride, da, steps, make_fast_div((unsigned)stride),   \
                       make_fast_div((unsigned)nw), make_fast_div((unsigned)angles_per_sza), lut_dev,            \
                       xcd_mode, duty, useful, xcd_slots_dev)
    if (tune.nt) {
        if (tune.depth == 1) GORT_FLAT(1, true); else if (tune.depth == 2) GORT_FLAT(2, true); else GORT_FLAT(4, true);
    } else {
        if (tune.depth == 1) GORT_FLAT(1, false); else if (tune.depth == 2) GORT_FLAT(2, false); else GORT_FLAT(4, false);
    }
#undef GORT_FLAT
    return check_launch("expand_flat_kernel");
}

// ---- aligned flat form of the stream expansion: large, wide streams without component spectra ----
static bool stream_uses_flat(int nw, long nA, bool want_scomp)
{
    return tuning().flat && !want_scomp && nw >= CHUNK && nA * (long)nw >= (1L << 22);
}

// panel shape of the per-line stream kernel: stride W (chunks) and steps K per wave
static void stream_panel_shape(int nw, long chunks, long *stride, int *steps)
{
    const ExpandTuning &tune = tuning();
    // small streams: fewer waves, so that a wave still has ~6 steps to spread its prologue (24 band constants per lane)
    // over - 3000 lines: 20 us with 8404 waves, 36 us with 33616; 8192 lines: 37 against 43 (tools/probes/perline_small.sh)
    long target = tune.stream_waves;
    if (chunks / 6 < target) target = chunks / 6 < 4202 ? 4202 : chunks / 6;
    *stride = flat_stride(nw, chunks, target);
    *steps = tune.stream_steps;
}

// readable records the stream expansion may touch behind the last line (the caller also keeps ONE in front)
long expand_stream_tail_pad_records(int nw, long nA)
{
    if (!stream_uses_flat(nw, nA, false)) return 0;
    long stride;
    int steps;
    stream_panel_shape(nw, (nA * (long)nw + 2 * CHUNK - 2) / CHUNK, &stride, &steps);
    return 2 * (stride * CHUNK / nw) + 4;       // one step of prefetch (da lines) + wrap record + slack
}

static int launch_expand_stream_flat(const gort_canopy *canopy_dev, const double *L_dev, int nw,
                                     const double *coef_dev, long nA, double *rsurf_dev, int *xcd_slots_dev,
                                     const int *direct_flag_dev, hipStream_t s)
{
    (void)canopy_dev;
    const ExpandTuning &tune = tuning();
    const long n_total = nA * (long)nw;
    const int shift = (int)((reinterpret_cast<uintptr_t>(rsurf_dev) / sizeof(double)) % CHUNK);
    const long chunks = (n_total + shift + CHUNK - 1) / CHUNK;
    long stride;
    int steps;
    stream_panel_shape(nw, chunks, &stride, &steps);
    const long panels = (chunks + (long)steps * stride - 1) / ((long)steps * stride);
    if (panels * stride >= (1L << 31) || chunks >= (1L << 31) || stride * CHUNK >= (1L << 30))
        return fail(GORT_EINVAL, "stream expansion: %ld chunks in %ld waves is beyond the kernel's 32-bit indices", chunks,
                    panels * stride);
    const int da = (int)(stride * CHUNK / nw);
    const int xcd_mode = resolve_xcd_mode(xcd_slots_dev);
    const long useful = (panels * stride + 3) / 4;
    XcdDuty duty;
    // equal XCD shares: this kernel is VALU bound, the duty weights of the LUT kernel (28:32) change nothing here
    // (tried: 28:32, 32:28, 30:32 against equal, 65 536 and 1 048 576 lines)
    const long nblocks = plan_xcd_duty(xcd_mode, useful, nullptr, duty);
    if (nblocks >= (1L << 31)) return fail(GORT_EINVAL, "stream expansion: %ld workgroups in one launch", nblocks);
    const dim3 grid((unsigned)nblocks);
    if (tune.nt)
        hipLaunchKernelGGL(expand_flat_stream_kernel<true>, grid, dim3(256), 0, s, L_dev, nw, coef_dev, n_total, shift, stride,
                           da, steps, make_fast_div((unsigned)stride), make_fast_div((unsigned)nw), rsurf_dev, xcd_mode, duty,
                           useful, xcd_slots_dev, direct_flag_dev);
    else
        hipLaunchKernelGGL(expand_flat_stream_kernel<false>, grid, dim3(256), 0, s, L_dev, nw, coef_dev, n_total, shift, stride,
                           da, steps, make_fast_div((unsigned)stride), make_fast_div((unsigned)nw), rsurf_dev, xcd_mode, duty,
                           useful, xcd_slots_dev, direct_flag_dev);
    return check_launch("expand_flat_stream_kernel");
}

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
