// gort_gap.hip -- Pn / EPgap / KOpen gap-probability integrals on gfx950.
//
// One 256-thread workgroup per canopy member.  The four live products of the
// reference's gortt_gap_probabilities (gortt_pn_kopen.c:7-129; what `gortt -W`
// writes, gortt.c:123-128) are produced in five barrier-separated phases whose
// intermediates live in LDS (about 32 KiB per workgroup):
//
//   1  V_gamma(h,t) -> P(n=0|h,t)       15 x 91 crown-projection volumes  (:24-32,149-323)
//   2  ES(0,t); capsule volumes          91 mean chords, 13 x 91 vol()     (:534-645, :665-924)
//   3  Poisson crown-count mixture       13 x 90 (s', t) pairs x 30 crowns  (:457-527)
//   4  EPgap(0,t) = sum over s'          90 zeniths                         (:1083-1125)
//   5  KOpen / KOpenEP trapezoid over the 91 zeniths: one term per lane,
//      wavefront shuffle reduction, cross-wave combine through LDS        (:329-391)
//
// Only what the outputs depend on is evaluated: no vb/fb/t_open
// tables, path-length statistics for h=0 only, and vol() once per (s',t) instead of
// once per crown count.  The path-length histogram of the reference is not
// materialised: EPgap(0,t) = sum_{s',n} P_n P_s' exp(-bin(s) ds tau') is the same sum
// regrouped, with bin(s) = (int)(s/ds+0.5) evaluated exactly as the reference does.
//
// fp64 throughout; compiled with -ffp-contract=off so that the quantities feeding
// integer truncations (loop trip counts, histogram bins) are formed by the same IEEE
// operations as on the CPU.  The Q08 closed form (gortt_pn_kopen.c:1144-1200) is the
// `use_q08` branch of the same kernel.
#include <hip/hip_runtime.h>

#include "gort_internal.h"

namespace gort {
namespace {

constexpr int GAP_THREADS = 256;
constexpr double PI = 3.14159265358979323846;

struct Crown {          // scalars of one member, in registers
    double r, rr, rrr, lv_p, tau_p, ds, dz_p, h1_p, h2_p;
};

__device__ inline double sec_(double x) { return 1.0 / cos(x); }

// area of the disc of radius r on the x <= x_cut side of a chord        (:285-305)
__device__ double disc_left_of(double r, double x_cut)
{
    const double whole = PI * r * r;
    const double ax = fabs(x_cut);
    const double ang = acos(ax / r) * 2.0;
    const double sector = whole * ang / (2.0 * PI);
    const double tri = ax * sqrt(r * r - x_cut * x_cut);
    return x_cut > 0.0 ? whole - (sector - tri) : sector - tri;
}

// horizontal cross-section at height z of the crown volume projected along zenith t
// (transformed space; a circle, an ellipse, or a circle/ellipse mix)      (:170-282,309-323)
__device__ double projected_section(const Crown &c, double t, double sin_t, double h, double z)
{
    if (z < h - c.r) return 0.0;
    const double lo = h - c.r * sin_t, hi = h + c.r * sin_t;
    if (z <= lo) {
        const double a = c.rr - (h - z) * (h - z);
        const double rp = a <= 0 ? 0 : sqrt(a);
        return PI * rp * rp;
    }
    if (z > lo && z < hi) {
        const double dzh = h - z;
        const double rp = sqrt(c.rr - dzh * dzh);
        const double ct = cos(t);
        const double x_cc = dzh * tan(t);
        const double x_p = x_cc / (1.0 - ct * ct);
        const double circ = disc_left_of(rp, x_p - x_cc);
        const double bb = c.r * sec_(t);                       // semi-major axis of the ellipse
        const double ratio = bb / c.r;
        double ell = PI * c.r * c.r;
        ell -= disc_left_of(c.r, x_p / ratio);
        return circ + ell * ratio;
    }
    return PI * c.rr * sec_(t);
}

// V_gamma: midpoint rule over crown-centre heights; the loop variable is accumulated
// in floating point exactly like the reference's (:149-167)
__device__ double projection_volume(const Crown &c, double t, double h)
{
    const double st = sin(t);
    double vol = 0.0;
    // <= 14 steps for any crown gort_canopy_init accepts (dz_p = (z2_p - z1_p) / 14 and h2_p - h1_p < z2_p - z1_p); the
    // guard only keeps a step that rounds to nothing (a crown 1e300 wide) from spinning on the device for ever
    int guard = 0;
    for (double z = c.h1_p + c.dz_p / 2.0; z <= c.h2_p && guard < 64; z += c.dz_p, ++guard)
        vol += projected_section(c, t, st, h, z) * (c.dz_p);
    return vol;
}

// mean chord through one sphere centred at h, above the plane hz          (:566-645)
__device__ double mean_chord(const Crown &c, double hz, double h, double thp, double sin_thp, double cos_thp)
{
    if (hz > h + c.r - 0.0001) return 0.0;
    if (hz < h - c.r + 0.0001) return 4.0 * c.r / 3.0;
    const double v_sphere = 4.0 * PI * c.rrr / 3.0;
    const double zd = fabs(h - hz);
    const double cap_h = c.r - zd;
    const double v_cap = PI * cap_h * cap_h / 3.0 * (3.0 * c.r - cap_h);
    double v = hz > h ? v_cap : v_sphere - v_cap;
    v /= cos_thp;
    const double area = (h < hz) ? projected_section(c, thp, sin_thp, h, h - zd)
                                 : projected_section(c, thp, sin_thp, h, h + zd);
    return v / area;
}

// ES(z0,t): 20-point midpoint mean over crown-centre heights               (:534-563)
__device__ double expected_chord(const Crown &c, double hz, double thp)
{
    const double dh = (c.h2_p - c.h1_p) / (double)GORT_NH_ES;
    const double pcc = 1.0 / (c.h2_p - c.h1_p);
    const double st = sin(thp), ct = cos(thp);
    double es = 0.0;
    int guard = 0;                                           // 20 steps; see projection_volume
    for (double h = c.h1_p + dh / 2.0; h <= c.h2_p && guard < 64; h += dh, ++guard)
        es += mean_chord(c, hz, h, thp, st, ct) * (pcc * dh);
    return es;
}

// ---- volume, below the plane h_b, of the capsule swept from height zh to zs ----

struct Dir { double th, st, ct, tt; };   // zenith (primed) with sin, cos, tan

__device__ inline double wedge_integrand(double x, double b, double r, double tt)   // (:858-872)
{
    const double a1 = tt * (x - b);
    double a3 = (r * r - x * x) - a1 * a1;
    if (fabs(a3) < 0.0000000001) a3 = 0.0;
    return 2.0 * a1 * sqrt(a3);
}

// composite Simpson with 20 double-intervals                                (:811-854)
__device__ double wedge_volume(double b, double r, const Dir &d)
{
    const int m = 20;
    const double a1 = r * r - b * b * d.st * d.st;
    const double x0 = b * (d.st * d.st) + sqrt(a1) * d.ct;
    const double h = .50 * (x0 - b) / (double)m;
    double odd = 0.0, even = 0.0;
    for (int i = 0; i < m; ++i) odd += wedge_integrand(b + (double)(2 * i + 1) * h, b, r, d.tt);
    double v = 4.0 * odd;
    for (int i = 0; i < m - 1; ++i) even += wedge_integrand(b + (double)(2 * (i + 1)) * h, b, r, d.tt);
    v += 2.0 * even;
    v += wedge_integrand(x0, b, r, d.tt);
    v += wedge_integrand(b, b, r, d.tt);
    v *= h / 3.0;
    return v;
}

__device__ inline double lens_sector(double a1, double a2, double r)              // (:796-806)
{
    const double b1 = r * r * a1 - (a1 * a1 * a1) / 3.0;
    const double b2 = r * r * a2 - (a2 * a2 * a2) / 3.0;
    return PI * (b2 - b1) / 2.0;
}

__device__ double cut_hemisphere(double hh, double hh_b, double r, const Dir &d)  // (:771-792)
{
    const double tmp = hh - hh_b;
    const double x = -1.0 * tmp * d.st + sqrt(r * r - tmp * tmp) * d.ct;
    const double b = -tmp / d.st;
    return wedge_volume(b, r, d) + lens_sector(x, r, r);
}

__device__ inline double half_disc_primitive(double x, double r)                  // (:876-886)
{
    return .50 * x * sqrt(r * r - x * x) + .50 * r * r * asin(x / r);
}

__device__ double cut_cylinder(double r, double h1, double h2, double h)          // (:891-924)
{
    const double slope = h / (h2 - h1);
    const double q1 = sqrt(r * r - h1 * h1), q2 = sqrt(r * r - h2 * h2);
    double v = q1 * q1 * q1 - q2 * q2 * q2;
    v /= 3.0;
    v -= h1 * (half_disc_primitive(h2, r) - half_disc_primitive(h1, r));
    v *= 2.0 * slope;
    if (h2 < r) {
        const double phi = acos(h2 / r);
        v += (r * r * phi - r * sin(phi) * h2) * h;
    }
    return v;
}

// seven cases in where the plane h_b cuts the capsule                        (:665-768)
__device__ double capsule_below(const Crown &c, double zh, double zs, const Dir &d, double h_b)
{
    const double r = c.r, rs = r * d.st;
    const double half_ball = (2.0 / 3.0) * PI * c.rrr;
    if ((zh - r) >= h_b) return 0.0;
    if ((zh - rs) >= h_b) {
        const double ht = r - (zh - h_b);
        return (PI / 3.0) * ht * ht * (3.0 * r - ht);
    }
    if ((zh + rs) >= h_b) {
        double sp1 = half_ball;
        sp1 -= cut_hemisphere(zh, h_b, r, d);
        const double h_tt = (h_b - (zh - rs)) / d.ct;
        const double hh1 = (zh - h_b) / d.st;
        double cyl, sp2;
        if (zs - rs >= h_b) {
            cyl = cut_cylinder(r, hh1, r, h_tt);
            sp2 = 0.0;
        } else {
            const double hh2 = (zs - h_b) / d.st;
            const double hh = (zs - zh) / d.ct;
            cyl = cut_cylinder(r, hh1, hh2, hh);
            sp2 = cut_hemisphere(h_b, zs, r, d);
        }
        return sp1 + cyl + sp2;
    }
    if (zs - rs >= h_b) {
        const double th = (h_b - zh) / d.ct;
        const double cyl = PI * r * r * th;
        return half_ball + cyl;
    }
    if (zs + rs >= h_b) {
        const double h_tt = (zs + rs - h_b) / d.ct;
        const double hh1 = (h_b - zs) / d.st;
        const double th = (zs - zh) / d.ct;
        const double cyl = PI * r * r * th - cut_cylinder(r, hh1, r, h_tt);
        const double sp2 = cut_hemisphere(h_b, zs, r, d);
        return cyl + sp2 + half_ball;
    }
    double v0 = PI * c.rr * ((zs - zh) / d.ct);
    v0 += (4.0 / 3.0) * PI * c.rrr;
    if (zs + r >= h_b) {
        const double ht = r - (h_b - zs);
        return v0 - (PI / 3.0) * ht * ht * (3.0 * r - ht);
    }
    return v0;
}

// sum over a wavefront with DPP-free shuffles (64 lanes)
__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(GAP_THREADS) void gap_probabilities_kernel(gort_canopy *members)
{
    __shared__ double s_pn0[GORT_NLAYERS][GORT_NTH];      // P(n=0|h,t)
    __shared__ double s_es[GORT_NTH];                     // ES(0,t)
    __shared__ double s_mean[GORT_NLAYERS][GORT_NTH];     // lv' (vol(h2') - vol(h1')) per (s',t); later partial EPgap
    __shared__ double s_epgap[GORT_NTH];
    __shared__ double s_red[2][GAP_THREADS / 64];

    gort_canopy &m = members[blockIdx.x];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    Crown c;
    c.r = m.r;  c.rr = m.rr;  c.rrr = m.rrr;  c.lv_p = m.lv_p;  c.tau_p = m.tau_p;
    c.ds = m.ds;  c.dz_p = m.dz_p;  c.h1_p = m.h1_p;  c.h2_p = m.h2_p;

    if (m.use_q08) {
        // Lewis' closed form as used in Quaife et al. 2008             (:1144-1200)
        const double cov = PI * m.rr * m.lambda;
        const double l = m.favd * m.b * 4. / 3. * cov;
        const double k2 = 0.348535 * pow(cov, (-1.08069 - 0.0874595 * cov));
        const double k1 = 0.0014166;
        const double a = cov * (exp(k1 * cov * cov) - exp(-k2 * l));
        for (int t = tid; t < GORT_NTH; t += GAP_THREADS) {
            const double ctp = cos(m.theta_p[t]);
            const double p = exp(-cov / ctp);
            s_pn0[0][t] = p;
            s_epgap[t] = exp(-a / ctp) - p;
        }
        __syncthreads();
    } else {
        // ---- phase 1: P(n=0) for every (height, zenith) ----
        for (int i = tid; i < GORT_NLAYERS * GORT_NTH; i += GAP_THREADS) {
            const int t = i / GORT_NLAYERS, h = i - t * GORT_NLAYERS;
            const double vg = projection_volume(c, m.theta_p[t], m.height_p[h]);
            s_pn0[h][t] = exp(-1.0 * c.lv_p * vg);
        }
        // ---- phase 2: ES(0,t) and the expected crown count per (s',t) ----
        const double z0 = m.height_p[0];
        for (int t = tid; t < GORT_NTH - 1; t += GAP_THREADS)
            s_es[t] = expected_chord(c, z0, m.theta_p[t]);
        for (int i = tid; i < (GORT_NLAYERS - 2) * (GORT_NTH - 1); i += GAP_THREADS) {
            const int t = i / (GORT_NLAYERS - 2), sp = 1 + (i - t * (GORT_NLAYERS - 2));
            Dir d;
            d.th = m.theta_p[t];  d.st = sin(d.th);  d.ct = cos(d.th);  d.tt = tan(d.th);
            const double zs = m.height_p[sp];
            double mean = capsule_below(c, z0, zs, d, c.h2_p) - capsule_below(c, z0, zs, d, c.h1_p);
            mean *= c.lv_p;
            s_mean[sp][t] = mean;
        }
        __syncthreads();
        // ---- phase 3: Poisson mixture over the number of crowns penetrated ----
        for (int i = tid; i < (GORT_NLAYERS - 2) * (GORT_NTH - 1); i += GAP_THREADS) {
            const int t = i / (GORT_NLAYERS - 2), sp = 1 + (i - t * (GORT_NLAYERS - 2));
            const double ctp = cos(m.theta_p[t]);
            const double s_p = (m.height_p[sp] - z0) / ctp;
            const double p_sp = s_pn0[sp + 1][t] - s_pn0[sp][t];          // P(s=0) differencing (:40-45)
            const double mean = s_mean[sp][t];
            const double es = s_es[t];
            const double e_m = exp(-mean);
            const double norm = 1.0 - e_m;
            double pw = 1.0, fact = 1.0, acc = 0.0;
            for (int n = 1; n <= GORT_MAXCROWNS; ++n) {
                pw *= mean;
                fact *= (double)n;
                const double P_n = (pw * e_m) / (fact * norm);
                const double s = s_p * (1.0 - exp(-1.0 * (double)n * es / s_p));
                const int bin = (int)(s / c.ds + 0.5);                      // histogram bin (:134-139)
                acc += P_n * exp(-((double)bin * c.ds) * c.tau_p);
            }
            // ordering hazard: s_mean[sp][t] is read above and overwritten here by the same thread only
            s_mean[sp][t] = acc * p_sp;
        }
        __syncthreads();
        // ---- phase 4: EPgap(0,t) ----
        for (int t = tid; t < GORT_NTH; t += GAP_THREADS) {
            double e = 0.0;
            if (t < GORT_NTH - 1)
                for (int sp = GORT_NLAYERS - 2; sp >= 1; --sp) e += s_mean[sp][t];
            s_epgap[t] = e;                                                 // [90] stays 0 (:1099)
        }
        __syncthreads();
    }

    // ---- phase 5: hemispherical openness, trapezoid in theta (:361-375) ----
    double term_o = 0.0, term_e = 0.0;
    if (tid >= 1 && tid < GORT_NTH) {
        const double s1 = sin(2.0 * m.theta[tid]), s0 = sin(2.0 * m.theta[tid - 1]);
        term_o = (s_pn0[0][tid] * s1 + s_pn0[0][tid - 1] * s0) / 2.0 * m.dth;
        term_e = (s_epgap[tid] * s1 + s_epgap[tid - 1] * s0) / 2.0 * m.dth;
    }
    term_o = wave_sum(term_o);
    term_e = wave_sum(term_e);
    if (lane == 0) { s_red[0][wave] = term_o;  s_red[1][wave] = term_e; }
    __syncthreads();
    if (tid == 0) {
        double ko = 0.0, ke = 0.0;
        for (int w = 0; w < GAP_THREADS / 64; ++w) { ko += s_red[0][w];  ke += s_red[1][w]; }
        m.k_open = ko;
        m.k_openep = ke;
    }
    for (int t = tid; t < GORT_NTH; t += GAP_THREADS) {
        m.p_n0[t] = s_pn0[0][t];
        m.epgap[t] = s_epgap[t];
    }
}

}  // namespace

int launch_gap_probabilities(gort_canopy *members_dev, int n_members, void *stream)
{
    if (n_members <= 0) return GORT_OK;
    hipLaunchKernelGGL(gap_probabilities_kernel, dim3((unsigned)n_members), dim3(GAP_THREADS), 0,
                       (hipStream_t)stream, members_dev);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(GORT_ENODEVICE, "gap_probabilities_kernel: %s", hipGetErrorString(err));
    return GORT_OK;
}

}  // namespace gort
