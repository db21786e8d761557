// gort_geometry.hip -- the angle-only stage: one thread per angle line (streams) or one workgroup per four LUT rows
// (grids).  Per angle tuple the kernels evaluate the ~35 fp64 transcendentals of gortt_kg / gortt_kc / gortt_kuusk
// (gort_geometry.h) once and leave a record for the expansion kernels - or, for a few bands, the samples themselves.
//
//   wavelength only      L[11][nw]          lambda_table_kernel (gort_tables.hip)
//   angle tuple only     coef[nA][16]       geometry_*_kernel   (this file)
//   (sun zenith, band)   C0,B,Z,G,T         sun_terms()         (gort_device.h)
//   sample               rsurf = aC*C0 + aB*B + aZ*Z + aG*G + aT*T   (5 FMAs, 8 B stored)
//
// which is an exact regrouping of gortt.c:484-557 (no approximation; rounding differs at the 1e-16 level).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "gort_geometry.h"
#include "gort_stamps.h"

GORT_STAMPS_DEFINE(geometry)

namespace gort {
namespace {

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
                                                               double *__restrict__ rsurf, int proportions_wanted)
{
    // FUSED, 9 ... 16 bands: a wave's 64 lines x nw samples are consecutive doubles of the output, and a lane storing its own
    // line's writes 8 bytes of every nw-th double per instruction; the wave turns them through LDS eight bands at a time instead
    // and lane k stores elements k, k + 64, ... (the few-band LUT's turn, geometry_grid_kernel below: a million lines x 16
    // bands 104 -> 92 us) - so every lane of a wave stays to the end, the lanes behind the stream evaluating its last line
    // and storing nothing
    __shared__ double s_val[FUSED ? 4 : 1][64][8];
    const long a_own = (long)blockIdx.x * blockDim.x + threadIdx.x;
    GORT_STAMPS_BEGIN();
    GORT_STAMP(0);
    if (a_own >= nA && (!FUSED || (a_own & ~63L) >= nA)) return;      // (whole waves behind the stream leave)
    const bool live = a_own < nA;
    const long a = live ? a_own : nA - 1;
    // blockIdx.z = ensemble member: its canopy, its nA records (the angle lines are shared)
    const long member = blockIdx.z;
    const gort_canopy &c = canopy[member];
    double vza, sza, saa, raa;
    normalise_angles(angles[4 * a], angles[4 * a + 1], angles[4 * a + 2], angles[4 * a + 3], vza, sza, saa, raa);
    GORT_STAMP_ANCHOR(raa);
    GORT_STAMP(1);                                           // the angle line is there
    GeomOut g;
    geometry_core(c, vza, sza, raa, g, K == nullptr && proportions_wanted == 0);      // reflectances only: gort_geometry.h, row_terms
    GORT_STAMP_ANCHOR(g.A);
    GORT_STAMP(2);                                           // geometry
    if (FUSED) {
        double rec[GORT_COEF_STRIDE];
        store_coef(rec, c, g);
        const LineTerms l = line_terms_of_record(rec, c.k_openep, c.k_open);
        const double *__restrict__ Lm = L + member * L_NSLOT * nw;
        if (nw <= 8) {                 // up to eight bands a lane stores its own line's (measured: the turn costs what it saves)
            double *__restrict__ o = rsurf + (member * nA + a) * nw;
            if (live)
                for (int i = 0; i < nw; ++i) o[i] = stream_sample(l, stream_band(load_band(Lm, nw, i)));
        } else {
            const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
            double (*val)[8] = s_val[wave];
            const long a_wave = a_own - lane;                                  // the wave's first line
            const int lines_here = nA - a_wave < 64 ? (int)(nA - a_wave) : 64;
            double *__restrict__ o = rsurf + (member * nA + a_wave) * nw;      // its 64 x nw samples
            for (int b0 = 0; b0 < nw; b0 += 8) {
                const int w = nw - b0 < 8 ? nw - b0 : 8;
                for (int b = 0; b < w; ++b) val[lane][b] = stream_sample(l, stream_band(load_band(Lm, nw, b0 + b)));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                int line = w == 8 ? lane >> 3 : lane / w, b = lane - line * w;
                const int line_step = w == 8 ? 8 : 64 / w, band_step = 64 - line_step * w;
                for (int e = 0; e < w; ++e) {                                    // element lane + 64 e of the pass's 64 x w
                    if (line < lines_here) o[(long)line * nw + b0 + b] = val[line][b];
                    b += band_step;
                    line += line_step;
                    if (b >= w) { b -= w;  ++line; }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();                              // before the next pass overwrites the samples
            }
        }
    } else if (layout == 0) {
        store_coef(coef + (member * nA + a) * GORT_COEF_STRIDE, c, g);
    } else {
        double rec[GORT_COEF_STRIDE];
        store_coef(rec, c, g);
        const LineTerms l = line_terms_of_record(rec, c.k_openep, c.k_open);
        double *o = coef + (member * nA + a) * GORT_COEF_STRIDE;
        o[0] = l.alpha;  o[1] = l.am;  o[2] = l.P1m;  o[3] = l.P2m;  o[4] = l.Q1;  o[5] = l.Q2;  o[6] = l.Q3;  o[7] = l.Q4;
        o[8] = l.Q5;  o[9] = l.Q6;  o[10] = l.mu;  o[11] = l.t0;  o[12] = l.t0m;  o[13] = 0.0;  o[14] = 0.0;  o[15] = 0.0;
    }
    if (K && live) {
        double *k = K + 4 * (member * nA + a);
        k[0] = g.Kc;  k[1] = g.Kg;  k[2] = g.Kt;  k[3] = g.Kz;
    }
    GORT_STAMP(3);                                           // samples / record stored
    GORT_STAMPS_END(geometry, a >> 6, (a & 63) == 0);
}

// grid nodes generated from indices; identical to streaming "vza phi sza 0" (SURVEY 8d, C3).
// A workgroup takes GEOM_ROWS LUT rows (4, 6 or 8: geom_rows_per_workgroup) or, in single-member launches, a span of nodes
// (below); a row = (member, sun zenith, view zenith): the azimuth-independent terms of each of its rows are evaluated ONCE into LDS, the rows side by side on the first lanes of one wavefront (the ~25
// transcendentals of a row are a serial chain: four rows cost the issue time of one), then the lanes walk the
// GEOM_ROWS x nphi azimuth nodes.  With 361 nodes per row this removes ~70 % of the transcendentals of the
// per-tuple form.
constexpr int GEOM_ROW_THREADS = 128;
constexpr int GEOM_FUSED_MAX_BANDS = 8;      // the fused form serves grids of up to this many bands (gort_api.hip: grid_rows)
#ifndef GORT_GEOM_SPAN_THREADS
#define GORT_GEOM_SPAN_THREADS 256
#endif
constexpr int GEOM_SPAN_THREADS = GORT_GEOM_SPAN_THREADS;      // threads of a workgroup of the node-partitioned form
constexpr int GEOM_SPAN_ROWS = GEOM_SPAN_THREADS / 32 + 4;          // rows such a workgroup may touch (8 / 12 / 20 at 128 / 256 / 512 threads)
// Azimuth table (round 5).  In a grid of non-negative zeniths a node's relative azimuth - and with it M::sincos(raa), ~75 of
// a node's ~560 instructions - depends on its azimuth index only (gortt.c:240-279: saa = 0, vaa = phi): the waves that wait
// for the row terms anyway fill a table of (raa, sin, cos) per azimuth node in LDS, and the node loop reads it.  The same
// functions on the same numbers: the same bits.
constexpr int GEOM_AZ_TABLE = 384;           // azimuth nodes per row the table holds (the mirrored hemisphere has 181, the full circle 361)

// mode 0: full stream records (GORT_COEF_STRIDE doubles per node); 1: compact 64-B records for the LUT kernel;
// 2: FUSED for grids of a few bands (BASELINE config 3 is one band): the node's samples are formed right here from
// its coefficients - no 128-B record per node written and read back (383 MB each way for the hemisphere grid,
// more than the arithmetic costs) - with the same sun_terms()/dot5() as the two-kernel path: same bits.
#ifndef GORT_GEOM_WAVES
#define GORT_GEOM_WAVES 3      // the row-partitioned form: 131 VGPRs, three waves per SIMD (the node-partitioned one: 120, four)
#endif
// ONE_MEMBER: every row of the launch belongs to one member (any single-canopy grid: BASELINE configs 1, 3 and a rank's
// slab of the metric grid).  Two things follow.  (1) Its canopy and the first band's constants are uniform and are read
// once, ahead of the node loop, where the general form reads them per lane and node (a dozen vector loads and three exposed
// waits per node).  (2) The launch is partitioned by NODES, not rows: workgroup k takes the nodes [k T / G, (k + 1) T / G)
// of the launch's T = rows x per_row and evaluates the terms of the <= GEOM_ROWS rows its span touches - so the launcher
// can choose G = the machine's slots exactly (at 120 VGPRs four waves per SIMD fit: 2048 workgroups), every CU gets the
// same eight workgroups of the same length, and the grid's row count no longer decides how full the machine is
// (the row form fills it 2.7 waves per SIMD deep for the hemisphere, unevenly where four fit: C3 61.7 us against 55.6).
template <int GEOM_ROWS, bool ONE_MEMBER>
__global__ __launch_bounds__(ONE_MEMBER ? GEOM_SPAN_THREADS : GEOM_ROW_THREADS) __attribute__((amdgpu_waves_per_eu(ONE_MEMBER ? 4 : GORT_GEOM_WAVES)))
void geometry_grid_kernel(const gort_canopy *__restrict__ canopies,
                                                                          gort_grid g, long row_begin, long n_rows,
                                                                          double *__restrict__ coef, int compact,
                                                                          const double *__restrict__ Lall, int nw,
                                                                          double *__restrict__ rsurf, int mirror, int az_table,
                                                                          const double *__restrict__ sun_tab, int q_begin)
{
    __shared__ RowTerms s_row[GEOM_ROWS];
    __shared__ double s_az[3][GEOM_AZ_TABLE];
    __shared__ RowScratch s_scr[GEOM_ROWS];
    __shared__ double s_sun_terms[GEOM_ROWS][GEOM_FUSED_MAX_BANDS][5];      // fused form: C0, B, Z, G, T per (row, band)
    __shared__ int s_member[GEOM_ROWS], s_q[GEOM_ROWS];               // a row's member and its sun row member * nsza + isza
    __shared__ double s_vza_deg[GEOM_ROWS], s_sza_deg[GEOM_ROWS];
    const long rows_per_member = (long)g.nsza * g.nvza;
    const long member0 = row_begin / rows_per_member;                  // ONE_MEMBER: the member of every row
    // mirror: the azimuth nodes run once round the full circle from phi0 = 0, and everything below depends on the
    // relative azimuth through cos(raa), sin^2(raa) and the folded raa / pi only (overlap, ph_r, frac, cos xi:
    // gortt_brdf.c:23-100, 118-169, 650-666): node l' = nphi - 1 - l is the mirror image of node l.  Half the nodes
    // are evaluated and each result is written twice (the device's cos of 2 pi - x and of x differ in the last place,
    // as the reference's do: the images agree with their own evaluation to rounding, 1e-15).
    GORT_STAMPS_BEGIN();
    GORT_STAMP(0);
    const int per_row = mirror ? (g.nphi + 1) / 2 : g.nphi;
    long first;                                                         // first row of this block, relative to row_begin
    int rows_here, rel0, rel1;                                          // its nodes, counted from node 0 of row `first`
    if (ONE_MEMBER) {
        const long total = n_rows * per_row;
        const long node0 = (long)blockIdx.x * total / gridDim.x, node1 = ((long)blockIdx.x + 1) * total / gridDim.x;
        first = node0 / per_row;
        rows_here = node1 > node0 ? (int)((node1 - 1) / per_row - first) + 1 : 0;
        rel0 = (int)(node0 - first * per_row);
        rel1 = (int)(node1 - first * per_row);
    } else {
        first = (long)blockIdx.x * GEOM_ROWS;
        rows_here = n_rows - first < GEOM_ROWS ? (int)(n_rows - first) : GEOM_ROWS;
        rel0 = 0;
        rel1 = rows_here * per_row;
    }
    // row i of this block: its member and the two zeniths as the grid's lines type them
    auto row_of = [&](int i, long &member, double &vza_deg, double &sza_deg) {
        const long grow = row_begin + first + i;
        member = grow / rows_per_member;
        const long row = grow - member * rows_per_member;
        const int isza = (int)(row / g.nvza), ivza = (int)(row % g.nvza);
        vza_deg = g.vza0 + ivza * g.dvza;
        sza_deg = g.sza0 + isza * g.dsza;
    };
    constexpr int THREADS = ONE_MEMBER ? GEOM_SPAN_THREADS : GEOM_ROW_THREADS;
    if (az_table && (int)threadIdx.x >= 64)                                 // the waves that would wait for the row terms
        for (int l = (int)threadIdx.x - 64; l < per_row; l += THREADS - 64) {
            double vza, sza, saa, raa, sn, cs;
            normalise_angles(0.0, g.phi0 + l * g.dphi, 0.0, 0.0, vza, sza, saa, raa);
            FastMath::sincos(raa, sn, cs);
            s_az[0][l] = raa;
            s_az[1][l] = sn;
            s_az[2][l] = cs;
        }
    if ((int)threadIdx.x >= 64 && (int)threadIdx.x - 64 < rows_here) {     // on the second wave: the first one is busy below
        const int i = (int)threadIdx.x - 64;
        long member;
        double vza_deg, sza_deg;
        row_of(i, member, vza_deg, sza_deg);
        s_member[i] = (int)member;
        {
            const long grow = row_begin + first + i;
            s_q[i] = (int)(member * g.nsza + (grow - member * rows_per_member) / g.nvza);
        }
        s_vza_deg[i] = vza_deg;
        s_sza_deg[i] = sza_deg;
    }
    // the rows' azimuth-independent terms, split over lanes (gort_geometry.h); a LUT holds reflectances only
    row_terms_split(rows_here, s_row, s_scr, true, [&](int i, const gort_canopy *&c, double &vza, double &sza) {
        long member;
        double vza_deg, sza_deg, saa, raa;
        row_of(i, member, vza_deg, sza_deg);
        normalise_angles(vza_deg, g.phi0, sza_deg, 0.0, vza, sza, saa, raa);
        c = &canopies[ONE_MEMBER ? member0 : member];
    });
    GORT_STAMP(1);                                           // row terms
    // ONE_MEMBER: what the node loop reads of the canopy and (fused form) the first band's constants, once, ahead of the loop
    // fused form: the five (sun zenith, band) terms of the sample depend on the row and the band only - once per (row, band)
    // here instead of once per node (sun_terms() is ~50 of a node's ~560 instructions)
    if (compact == 2 && nw <= GEOM_FUSED_MAX_BANDS) {
        if ((int)threadIdx.x < rows_here * nw) {
            const int r = (int)threadIdx.x / nw, b = (int)threadIdx.x - r * nw;
            const long member = ONE_MEMBER ? member0 : (long)s_member[r];
            const gort_canopy &c = canopies[member];
            const SunTerms t = sun_terms(Lall + member * L_NSLOT * nw, nw, b, s_row[r].sun, c.k_open, c.k_openep);
            double *o = s_sun_terms[r][b];
            o[0] = t.C0;  o[1] = t.B;  o[2] = t.Z;  o[3] = t.G;  o[4] = t.T;
        }
        __syncthreads();
    }
    // a node's azimuth-dependent rest: from the table where there is one
    auto finish_node = [&](const gort_canopy &c, int r, int l, GeomOut &o) {
        if (az_table) {
            finish_angle(c, s_row[r], s_az[0][l], s_az[1][l], s_az[2][l], o);
        } else {
            double vza, sza, saa, raa;
            normalise_angles(s_vza_deg[r], g.phi0 + l * g.dphi, s_sza_deg[r], 0.0, vza, sza, saa, raa);
            finish_angle(c, s_row[r], raa, o);
        }
    };
    // what a wave's 64 lanes hand each other before they store (compact records: four lanes to a record; few-band samples: the
    // lanes' rows side by side) and where each lane's node - and its image - lies in the output
    __shared__ dbl2 s_rec[THREADS / 64][64][4];
    __shared__ long s_at[THREADS / 64][64][2];
    static_assert(GEOM_FUSED_MAX_BANDS * sizeof(double) == 4 * sizeof(dbl2), "the fused form's samples take the records' place");
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    if (compact == 2 && nw > 1) {
        // Few-band LUT: a node's nw samples are nw consecutive doubles, and a lane storing its own node's writes 8 bytes of
        // every nw-th double per instruction - every cache line of the wave's span nw times, an eighth of it each time
        // (12.5 us per band for the hemisphere).  The wave turns its samples through LDS instead, eight bands of its 64
        // nodes at a time: lane k stores elements k, k + 64, ... of the 64 x 8, consecutive lanes consecutive doubles
        // wherever consecutive nodes are consecutive in the output (a row's nodes are; its images run backwards, node by
        // node): 2.2 us per band.  Up to eight bands the five (sun zenith, band) terms of the workgroup's rows sit in LDS;
        // beyond, they come from the LUT path's table sun_tab[q - q_begin][5][nw] (launch_sun_table) - through the scalar
        // cache where the wave's nodes share a row (two waves of three do, at 181 nodes per row), per lane where they do not.
        double (*val)[GEOM_FUSED_MAX_BANDS] = reinterpret_cast<double (*)[GEOM_FUSED_MAX_BANDS]>(&s_rec[wave][0][0]);
        const bool wide = nw > GEOM_FUSED_MAX_BANDS;
        // element lane + 64 e of a pass of w bands: node (lane + 64 e) / w, band (lane + 64 e) % w
        auto store_pass = [&](int b0, int w) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int node = w == GEOM_FUSED_MAX_BANDS ? lane >> 3 : lane / w, b = lane - node * w;
            const int node_step = w == GEOM_FUSED_MAX_BANDS ? 8 : 64 / w, band_step = 64 - node_step * w;
            for (int e = 0; e < w; ++e) {
                const double x = val[node][b];
                const long at = s_at[wave][node][0], at2 = s_at[wave][node][1];
                if (at >= 0) rsurf[at * nw + b0 + b] = x;
                if (at2 >= 0) rsurf[at2 * nw + b0 + b] = x;
                b += band_step;
                node += node_step;
                if (b >= w) { b -= w;  ++node; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                                  // before the next pass overwrites the samples
        };
        for (int n0 = rel0 + (wave << 6); n0 < rel1; n0 += THREADS) {       // wave-uniform bounds: every lane takes part in the turn
            const int n = n0 + lane;
            long i = -1, i2 = -1;
            int r = n0 / per_row;                                            // lane 0's row: the lanes behind the span stay in it
            double aC = 0.0, aB = 0.0, aZ = 0.0, aG = 0.0, aT = 0.0;
            if (n < rel1) {
                r = n / per_row;
                const int l = n - r * per_row;
                const gort_canopy &c = canopies[ONE_MEMBER ? member0 : (long)s_member[r]];
                GeomOut o;
                finish_node(c, r, l, o);
                i = (first + r) * g.nphi + l;
                const int l2 = g.nphi - 1 - l;
                i2 = (mirror && l2 != l) ? (first + r) * g.nphi + l2 : -1;
                double rec[GORT_COEF_STRIDE];
                store_coef(rec, c, o);
                aC = rec[A_C];  aB = rec[A_B];  aZ = rec[A_Z];  aG = rec[A_G];  aT = rec[A_T];
            }
            s_at[wave][lane][0] = i;
            s_at[wave][lane][1] = i2;
            if (!wide) {
                for (int b = 0; b < nw; ++b) {
                    const double *t = s_sun_terms[r][b];
                    val[lane][b] = dot5(aC, aB, aZ, aG, aT, t[0], t[1], t[2], t[3], t[4]);
                }
                store_pass(0, nw);
                continue;
            }
            // 9 ... 127 bands: the other way round.  Lanes are BANDS now - of as many nodes at a time as fit the wave (seven
            // nodes of 9 bands, one of 33 ... 64; from 65 bands a lane has a second band 64 further on) - a lane keeps its bands'
            // five (sun zenith, band) terms of its node's row in hand, and the wave walks its 64 nodes G at a time: a node's five
            // coefficients come from the lane that made them (through LDS), five or ten FMAs, and whole rows - nw consecutive
            // doubles each, G of them side by side - leave in one or two instructions; so do the images'.  (Turning eight bands
            // at a time like the narrow form wrote 64 B of every row per pass: 1.8 TB/s at 100 bands; this way 4.8.)
            val[lane][0] = aC;  val[lane][1] = aB;  val[lane][2] = aZ;  val[lane][3] = aG;  val[lane][4] = aT;
            val[lane][5] = (double)r;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                                  // the lanes' coefficients and s_at, before other lanes read them
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int n_here = rel1 - n0 < 64 ? rel1 - n0 : 64;              // wave-uniform
            const int G = nw <= 64 ? 64 / nw : 1;                            // nodes per step
            const int g = nw <= 64 ? lane / nw : 0, b_lo = lane - g * nw, b_hi = lane + 64;
            const bool lo_live = g < G && b_lo < nw, hi_live = nw > 64 && b_hi < nw;
            int r_cur = -1;
            double tl[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, th[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            for (int j0 = 0; j0 < n_here; j0 += G) {
                const int j = j0 + g;
                if (!lo_live || j >= n_here) continue;                       // (a lane without a band, or behind the wave's last node)
                const int r_j = (int)val[j][5];
                if (r_j != r_cur) {                                          // a new row (once or twice per 64 nodes): its sun terms
                    r_cur = r_j;
                    const double *__restrict__ t = sun_tab + (long)(s_q[r_j] - q_begin) * 5 * nw;
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        tl[k] = t[(long)k * nw + b_lo];
                        th[k] = hi_live ? t[(long)k * nw + b_hi] : 0.0;
                    }
                }
                const double cC = val[j][0], cB = val[j][1], cZ = val[j][2], cG = val[j][3], cT = val[j][4];
                const long at = s_at[wave][j][0], at2 = s_at[wave][j][1];
                const double vl = dot5(cC, cB, cZ, cG, cT, tl[0], tl[1], tl[2], tl[3], tl[4]);
                rsurf[at * nw + b_lo] = vl;
                if (at2 >= 0) rsurf[at2 * nw + b_lo] = vl;
                if (hi_live) {
                    const double vh = dot5(cC, cB, cZ, cG, cT, th[0], th[1], th[2], th[3], th[4]);
                    rsurf[at * nw + b_hi] = vh;
                    if (at2 >= 0) rsurf[at2 * nw + b_hi] = vh;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                                  // before the next round overwrites them
        }
        GORT_STAMP(2);
        GORT_STAMPS_END(geometry, (long)blockIdx.x * 4 + (threadIdx.x >> 6), (threadIdx.x & 63) == 0);
        return;
    }
    if (compact == 1) {
        // LUT path: the five expansion coefficients of a node, one 64-B record - and its image's.  A lane holding its record
        // would store it as four 16-B pieces 64 B apart: every store instruction a quarter of 32 cache lines.  The wave
        // turns its 64 records through LDS instead, four lanes to a record: an instruction then writes 16 whole records.
        for (int n0 = rel0 + (wave << 6); n0 < rel1; n0 += THREADS) {       // wave-uniform bounds: every lane takes part in the turn
            const int n = n0 + lane;
            long i = -1, i2 = -1;
            dbl2 piece[4] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
            if (n < rel1) {
                const int r = n / per_row, l = n - r * per_row;
                const gort_canopy &c = canopies[ONE_MEMBER ? member0 : (long)s_member[r]];
                GeomOut o;
                finish_node(c, r, l, o);
                i = (first + r) * g.nphi + l;
                const int l2 = g.nphi - 1 - l;
                i2 = (mirror && l2 != l) ? (first + r) * g.nphi + l2 : -1;
                double rec[GORT_COEF_STRIDE];
                store_coef(rec, c, o);
                piece[0].x = rec[A_C];  piece[0].y = rec[A_B];
                piece[1].x = rec[A_Z];  piece[1].y = rec[A_G];
                piece[2].x = rec[A_T];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) s_rec[wave][lane][p] = piece[p];
            s_at[wave][lane][0] = i;
            s_at[wave][lane][1] = i2;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int q = 16 * p + (lane >> 2);                      // the record this lane stores a quarter of
                    const long at = s_at[wave][q][k];
                    if (at >= 0) reinterpret_cast<dbl2 *>(coef + at * 8)[lane & 3] = s_rec[wave][q][lane & 3];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                                  // before the next round overwrites the records
        }
        GORT_STAMP(2);
        GORT_STAMPS_END(geometry, (long)blockIdx.x * 4 + (threadIdx.x >> 6), (threadIdx.x & 63) == 0);
        return;
    }
    for (int n = rel0 + (int)threadIdx.x; n < rel1; n += THREADS) {
        const int r = n / per_row, l = n - r * per_row;
        const long member = ONE_MEMBER ? member0 : (long)s_member[r];
        const gort_canopy &c = canopies[member];
        GeomOut o;
        finish_node(c, r, l, o);
        const long i = (first + r) * g.nphi + l;
        const int l2 = g.nphi - 1 - l;
        const long i2 = (mirror && l2 != l) ? (first + r) * g.nphi + l2 : -1;       // the image, if the node has one
        if (compact == 2) {
            double rec[GORT_COEF_STRIDE];
            store_coef(rec, c, o);
            for (int b = 0; b < nw; ++b) {
                const double *t = s_sun_terms[r][b];
                const double v = dot5(rec[A_C], rec[A_B], rec[A_Z], rec[A_G], rec[A_T], t[0], t[1], t[2], t[3], t[4]);
                rsurf[i * nw + b] = v;
                if (i2 >= 0) rsurf[i2 * nw + b] = v;
            }
        } else {
            store_coef(coef + i * GORT_COEF_STRIDE, c, o);
            if (i2 >= 0) store_coef(coef + i2 * GORT_COEF_STRIDE, c, o);
        }
    }
    GORT_STAMP(2);                                           // nodes
    GORT_STAMPS_END(geometry, (long)blockIdx.x * 4 + (threadIdx.x >> 6), (threadIdx.x & 63) == 0);
}

}  // namespace

// ------------------------------------------------------------------- launchers

// n_members > 1: blockIdx.z = member, canopy_dev[m], records coef_dev[m][nA][16], proportions K_dev[m][nA][4]
int launch_geometry_stream(const gort_canopy *canopy_dev, int n_members, const double *angles_dev, long nA,
                           double *coef_dev, double *K_dev, int layout, void *stream, bool proportions_wanted)
{
    if (nA <= 0 || n_members <= 0) return GORT_OK;
    if (n_members > 65535) return fail(GORT_EINVAL, "geometry: %d members in one launch (max 65535)", n_members);
    hipLaunchKernelGGL(geometry_stream_kernel<false>, dim3((unsigned)((nA + 255) / 256), 1, (unsigned)n_members), dim3(256), 0,
                       (hipStream_t)stream, canopy_dev, angles_dev, nA, coef_dev, K_dev, layout, (const double *)nullptr, 0,
                       (double *)nullptr, proportions_wanted ? 1 : 0);
    return check_launch("geometry_stream_kernel");
}

// few bands, no component spectra: one fused launch (GORT_STREAM_FUSE=0 keeps the two-kernel path, for tests)
bool stream_fuses(int nw, bool want_scomp)
{
    const char *v = ab_env("GORT_STREAM_FUSE");             // measuring build, read per call: the tests switch it inside one process
    return !(v && atoi(v) == 0) && !want_scomp && nw > 0 && nw <= 16;
}

// geometry and samples of a few-band stream in one launch (no records): rsurf_dev[m][nA][nw]
int launch_geometry_stream_fused(const gort_canopy *canopy_dev, int n_members, const double *L_dev, int nw,
                                 const double *angles_dev, long nA, double *rsurf_dev, double *K_dev, void *stream)
{
    if (nA <= 0 || nw < 0 || n_members <= 0 || (nw == 0 && !K_dev)) return GORT_OK;      // nw = 0: the proportions K alone
    if (n_members > 65535) return fail(GORT_EINVAL, "geometry: %d members in one launch (max 65535)", n_members);
#ifndef GORT_PROBE_GEOM_LDS_PAD
#define GORT_PROBE_GEOM_LDS_PAD 0      // probe builds: bytes of unused LDS per workgroup = fewer resident waves (tools/probes/geometry_occupancy.sh)
#endif
#if GORT_PROBE_GEOM_LDS_PAD > 0
    static const hipError_t pad_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&geometry_stream_kernel<true>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, GORT_PROBE_GEOM_LDS_PAD);
    if (pad_ok != hipSuccess) return fail(GORT_ENODEVICE, "probe build: %d bytes of LDS padding refused", GORT_PROBE_GEOM_LDS_PAD);
#endif
    hipLaunchKernelGGL(geometry_stream_kernel<true>, dim3((unsigned)((nA + 255) / 256), 1, (unsigned)n_members), dim3(256), GORT_PROBE_GEOM_LDS_PAD,
                       (hipStream_t)stream, canopy_dev, angles_dev, nA, (double *)nullptr, K_dev, 0, L_dev, nw, rsurf_dev, 0);
    return check_launch("geometry_stream_kernel<fused>");
}

// do the azimuth nodes of this grid run exactly once round the circle from 0 (node nphi-1-l mirrors node l)?
// Only for grids of non-negative zeniths, where the sun azimuth of every node is the 0 of the grid's lines.
static bool grid_mirrors(const gort_grid &g)
{
    static const bool on = !(ab_env("GORT_GRID_MIRROR") && atoi(ab_env("GORT_GRID_MIRROR")) == 0);
    return on && g.nphi >= 3 && g.phi0 == 0.0 && g.dphi > 0.0 && g.dphi * (g.nphi - 1) == 360.0 && g.sza0 >= 0.0 && g.dsza >= 0.0 &&
           g.vza0 >= 0.0 && g.dvza >= 0.0;
}

// Rows per workgroup.  A workgroup's life is the serial chain of its rows' azimuth-independent terms (on as many lanes
// as it has rows, the rest waiting) + its share of azimuth nodes; at 168 VGPRs a CU holds six of these two-wave
// workgroups.  What decides is whether the launch fits the machine in ONE round: the hemisphere's 8281 rows are 2071
// workgroups of four rows - 1.35 rounds of 1536 slots, the second a third full - but 1381 of six.  Measured with the
// mirrored nodes (BASELINE config 3, profiles/r03/c3_rows2.log): 4 rows 79.3 us, 5 rows 79.8 (still two rounds),
// 6 rows 68.7, 7 rows 76.3, 8 rows 83.6 (one round each, ever longer workgroups); 64-thread workgroups 84.3.
// So: the smallest of 4 / 6 / 8 rows that makes one round, 4 where nothing does.
// workgroup slots of the device at `waves` waves per SIMD (two-wave workgroups)
static long geom_slots(int waves, int threads = GEOM_ROW_THREADS)
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
        (void)hipGetLastError();
    }
    return (long)cus * (waves * 4 / (threads / 64));
}

static int geom_rows_per_workgroup(long rows)
{
    const long slots = geom_slots(GORT_GEOM_WAVES);
    for (int per : {4, 6, 8})
        if ((rows + per - 1) / per <= slots) return per;
    return 4;
}

static int launch_geometry_grid_any(const gort_canopy *canopy_dev, const gort_grid &g, long row_begin, long rows, double *coef_dev,
                                    int compact, const double *L_dev, int nw, double *rsurf_dev, void *stream,
                                    const double *sun_dev = nullptr, int q_begin = 0)
{
    const int mirror = grid_mirrors(g) ? 1 : 0;
    // the azimuth table: grids of non-negative zeniths (no line's azimuths are turned by pi, gortt.c:244-251) whose rows fit it
    const char *az = ab_env("GORT_GRID_AZ_TABLE");                  // measuring build, read per call: 0 = every node forms its own
    const bool az_on = !(az && atoi(az) == 0);
    const int az_table = az_on && g.sza0 >= 0.0 && g.dsza >= 0.0 && g.vza0 >= 0.0 && g.dvza >= 0.0 &&
                         (mirror ? (g.nphi + 1) / 2 : g.nphi) <= GEOM_AZ_TABLE ? 1 : 0;
    const dim3 block(GEOM_ROW_THREADS);
    hipStream_t s = (hipStream_t)stream;
    const long rows_per_member = (long)g.nsza * g.nvza;
    const char *br = ab_env("GORT_GRID_BY_ROWS");                      // measuring build: the general form for everything (read per call)
    const bool by_rows = br && atoi(br) != 0;
    if (!by_rows && row_begin / rows_per_member == (row_begin + rows - 1) / rows_per_member) {
        // one member: partitioned by nodes.  One round of the machine's slots (four waves per SIMD, two per workgroup) where the
        // launch has two node rounds per workgroup to give; a workgroup's span stays within GEOM_SPAN_ROWS rows (six row
        // lengths of nodes touch at most seven rows), so the biggest grids take more, equally long, workgroups instead.
        const int per_row = mirror ? (g.nphi + 1) / 2 : g.nphi;
        const long total = rows * per_row, slots = geom_slots(4, GEOM_SPAN_THREADS), span_max = (long)(GEOM_SPAN_ROWS - 2) * per_row;
        long G = (total + 2 * GEOM_SPAN_THREADS - 1) / (2 * GEOM_SPAN_THREADS);
        if (G > slots) G = slots;
        if ((total + G - 1) / G > span_max) G = (total + span_max - 1) / span_max;
        hipLaunchKernelGGL((geometry_grid_kernel<GEOM_SPAN_ROWS, true>), dim3((unsigned)G), dim3(GEOM_SPAN_THREADS), 0, s, canopy_dev, g, row_begin, rows,
                           coef_dev, compact, L_dev, nw, rsurf_dev, mirror, az_table, sun_dev, q_begin);
        return check_launch("geometry_grid_kernel");
    }
    const int per = geom_rows_per_workgroup(rows);
    const dim3 grid((unsigned)((rows + per - 1) / per));
#define GORT_GRID_LAUNCH(ROWS) hipLaunchKernelGGL((geometry_grid_kernel<ROWS, false>), grid, block, 0, s, canopy_dev, g, row_begin, rows, coef_dev, compact, L_dev, nw, rsurf_dev, mirror, az_table, sun_dev, q_begin)
    if (per == 4) GORT_GRID_LAUNCH(4);
    else if (per == 6) GORT_GRID_LAUNCH(6);
    else GORT_GRID_LAUNCH(8);
#undef GORT_GRID_LAUNCH
    return check_launch("geometry_grid_kernel");
}

int launch_geometry_grid(const gort_canopy *canopy_dev, const gort_grid &g, long row_begin, long row_end,
                         double *coef_dev, bool compact, void *stream)
{
    const long rows = row_end - row_begin;
    if (rows <= 0) return GORT_OK;
    return launch_geometry_grid_any(canopy_dev, g, row_begin, rows, coef_dev, compact ? 1 : 0, nullptr, 0, nullptr, stream);
}

int launch_geometry_grid_fused(const gort_canopy *canopy_dev, const double *L_dev, int nw, const gort_grid &g, long row_begin,
                               long row_end, double *rsurf_dev, void *stream, const double *sun_dev, int q_begin)
{
    const long rows = row_end - row_begin;
    if (rows <= 0 || nw <= 0) return GORT_OK;
    if (nw > GEOM_FUSED_MAX_BANDS && !sun_dev) return fail(GORT_EINVAL, "fused grid of %d bands: no sun table", nw);
    if (nw > 128) return fail(GORT_EINVAL, "fused grid of %d bands: a lane holds two bands at most (128)", nw);
    return launch_geometry_grid_any(canopy_dev, g, row_begin, rows, nullptr, 2, L_dev, nw, rsurf_dev, stream, sun_dev, q_begin);
}

}  // namespace gort
