// gort_stream_suns.hip -- wide streams whose lines share few sun zeniths, expanded with the LUT family's five-term sample
// (gortt.c:484-557 regrouped as in gort_internal.h: rsurf = aC C0 + aB B + aZ Z + aG G + aT T).
//
// The stream family forms a sample from twelve line scalars and twelve band constants in 28 issue slots because every
// line may have its own sun (gort_device.h); its wide kernel is co-limited by fp64 issue at ~0.7 of the HBM rate.  The
// LUT kernel spends 5 on a sample - the five (sun zenith, band) terms sit in the lane's registers - and reaches 0.92, but
// it can keep them there only because a grid's nodes come sorted by sun zenith and a lane keeps its bands for life.  An
// arbitrary stream with FEW DISTINCT sun zeniths (a scene, an orbit, a field campaign: one sun per acquisition) gets the
// same inner loop like this:
//   1. the sun-zenith table of the stream (gort_energy.hip's table pass keyed by the normalised zenith alone): idx[line]
//      = the place of the line's sun zenith among the distinct ones, the list of their first lines;
//   2. lines are bucketed by (sun zenith, alignment class): the class is where the line's row begins inside a 128-B
//      line of the output, (out/8 + line nw) mod 16 - within one bucket every row has the same bands in the same lanes
//      when it is cut into 1-KiB pieces on 128-B boundaries.  rank = an atomic counter per bucket, bucket starts = a scan
//      (buckets are padded to whole segments of SEG lines);
//   3. the (sun zenith, band) terms of the distinct zeniths: sun[q][5][nw] (gort_tables.hip, the LUT's table);
//   4. the geometry of every line, its five coefficients + its line number written at its SORTED position (64-B records);
//   5. the expansion: a wave = (segment of <= SEG sorted lines of one bucket, panel of 128 absolute columns).  It loads its
//      two bands' five sun terms once, then per line: one scalar record, ten FMAs, one 16-B store per lane = 1 KiB of
//      the line's row on a 128-B boundary.  The first and last piece of a row are partial lines (masked lanes): a row of
//      2101 bands shares two of its ~132 lines with its neighbours.
// Same functions on the same numbers as the LUT path: a stream made of a grid's nodes in any order equals the LUT bit
// for bit (tests/test_stream_forms.py); against the stream family the rounding differs (a few 1e-16 relative).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gort_flat.h"
#include "gort_geometry.h"

namespace gort {
namespace {

constexpr int SUNS_SEG = 64;             // sorted lines per segment = steps of a wave = one record per lane
constexpr int SUNS_CLASSES = 16;         // doubles per 128-B line: where a row can begin
constexpr int SUNS_REC = 8;              // doubles per sorted record: A_C..A_T, the line number, two pads
constexpr int SUNS_THREADS = 256;

__device__ __forceinline__ unsigned row_class(unsigned shift0, long line, int nw)
{
    return (unsigned)((shift0 + (unsigned long long)line * (unsigned)nw) & (SUNS_CLASSES - 1));
}

// one thread per line: its bucket's next rank
__global__ __launch_bounds__(SUNS_THREADS) void suns_rank_kernel(long nA, int nw, unsigned shift0, const unsigned *__restrict__ idx,
                                                                  unsigned *__restrict__ count, unsigned *__restrict__ rank)
{
    const long line = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= nA) return;
    const unsigned bucket = idx[line] * SUNS_CLASSES + row_class(shift0, line, nw);
    rank[line] = atomicAdd(&count[bucket], 1u);
}

// ONE workgroup: start[b] = sum over the buckets in front of b of their lines rounded up to whole segments (start has
// n_buckets + 1 entries, the last = the total); in place over count
constexpr int SUNS_SCAN_THREADS = 1024;
__global__ __launch_bounds__(SUNS_SCAN_THREADS) void suns_scan_kernel(unsigned *__restrict__ count, int n_buckets,
                                                                       unsigned *__restrict__ start, unsigned seg_len)
{
    __shared__ unsigned s_wave[SUNS_SCAN_THREADS / 64];
    __shared__ unsigned s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n_buckets; base += SUNS_SCAN_THREADS) {
        const int i = base + tid;
        const unsigned v = i < n_buckets ? (count[i] + seg_len - 1) / seg_len * seg_len : 0u;
        unsigned x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        unsigned before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (i < n_buckets) start[i] = before + x - v;
        __syncthreads();
        if (tid == SUNS_SCAN_THREADS - 1) s_carry = before + x;
        __syncthreads();
    }
    if (tid == 0) start[n_buckets] = s_carry;
}

// one thread per bucket: the segments of the bucket, seg[s] = (bucket, lines in the segment)
__global__ __launch_bounds__(SUNS_THREADS) void suns_segments_kernel(const unsigned *__restrict__ count, const unsigned *__restrict__ start,
                                                                      int n_buckets, uint2 *__restrict__ seg, unsigned seg_len)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    const unsigned n = count[b], s0 = start[b] / seg_len;
    for (unsigned k = 0; k * seg_len < n; ++k) {
        const unsigned left = n - k * seg_len;
        seg[s0 + k] = make_uint2((unsigned)b, left < seg_len ? left : seg_len);
    }
}

// one thread per line: the geometry (the stream path's: geometry_core on the normalised line), its five coefficients and
// its line number at the line's sorted position
__global__ __launch_bounds__(SUNS_THREADS) void suns_geometry_kernel(const gort_canopy *__restrict__ canopy,
                                                                      const double *__restrict__ angles, long nA, int nw,
                                                                      unsigned shift0, const unsigned *__restrict__ idx,
                                                                      const unsigned *__restrict__ rank, const unsigned *__restrict__ start,
                                                                      double *__restrict__ rec, double *__restrict__ K)
{
    const long a = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nA) return;
    const gort_canopy &c = canopy[0];
    double vza, sza, saa, raa;
    normalise_angles(angles[4 * a], angles[4 * a + 1], angles[4 * a + 2], angles[4 * a + 3], vza, sza, saa, raa);
    GeomOut g;
    geometry_core(c, vza, sza, raa, g, K == nullptr);
    double r[GORT_COEF_STRIDE];
    store_coef(r, c, g);
    const unsigned bucket = idx[a] * SUNS_CLASSES + row_class(shift0, a, nw);
    double *o = rec + (size_t)(start[bucket] + rank[a]) * SUNS_REC;
    dbl2 p;
    p.x = r[A_C];  p.y = r[A_B];  *reinterpret_cast<dbl2 *>(o) = p;
    p.x = r[A_Z];  p.y = r[A_G];  *reinterpret_cast<dbl2 *>(o + 2) = p;
    p.x = r[A_T];  p.y = __longlong_as_double((long long)a);  *reinterpret_cast<dbl2 *>(o + 4) = p;
    if (K) {
        double *k = K + 4 * a;
        k[0] = g.Kc;  k[1] = g.Kg;  k[2] = g.Kt;  k[3] = g.Kz;
    }
}

// lane l's v in every lane (l wave-uniform): two v_readlane_b32
__device__ __forceinline__ double lane_value(double v, int l)
{
    const long long bits = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)bits, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The expansion.  wave = (segment, panel); everything per step is wave-uniform and scalar (the record, the row's address),
// a lane carries its two bands' five sun terms.
template <bool NT, bool LANES>
__global__ __launch_bounds__(256) void suns_expand_kernel(const double *__restrict__ sun, const double *__restrict__ rec,
                                                           const uint2 *__restrict__ seg, const unsigned *__restrict__ total,
                                                           int nw, int panels, FastDiv div_panels, double *__restrict__ out,
                                                           unsigned seg_len, unsigned n_seg, int seg_fastest)
{
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    unsigned s;
    int panel;
    if (seg_fastest) {
        panel = (int)fast_div(wave, div_panels);          // div_panels divides by n_seg here
        s = wave - (unsigned)panel * n_seg;
        if (panel >= panels) return;
    } else {
        s = fast_div(wave, div_panels);
        panel = (int)(wave - s * (unsigned)panels);
    }
    if (s * seg_len >= *total) return;
    const uint2 info = seg[s];
    const unsigned q = info.x / SUNS_CLASSES, t = info.x % SUNS_CLASSES;
    const int n = (int)info.y;
    const int j0 = panel * CHUNK - (int)t;               // the band in lane 0's first element; the piece starts on a 128-B boundary
    if (j0 >= nw) return;
    const int lane = threadIdx.x & 63;
    const int band0 = j0 + EPL * lane;
    double b[EPL][5];
    bool live[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int band = band0 + j;
        live[j] = band >= 0 && band < nw;
        const double *bp = sun + (size_t)q * 5 * nw + (live[j] ? band : 0);
#pragma unroll
        for (int k = 0; k < 5; ++k) b[j][k] = bp[(size_t)k * nw];
    }
    const bool whole = j0 >= 0 && j0 + CHUNK <= nw;      // no masked lane in this piece
    // lane i holds the record of the segment's line i (seg_len <= 64): the whole segment's records arrive in one round trip
    // together with the sun terms, and a step reads its record out of that lane - no memory wait inside the loop (with the
    // record of step i + 1 requested through the scalar cache during step i every step cost a round trip of ~1 us, and the
    // launch went as fast as the resident waves could hide that: half the waves per CU, 34 % slower)
    const dbl2 *__restrict__ rp = reinterpret_cast<const dbl2 *>(rec + ((size_t)s * seg_len + (LANES && lane < n ? lane : 0)) * SUNS_REC);
    dbl2 r01, r23, r45;
    if (LANES) { r01 = rp[0];  r23 = rp[1];  r45 = rp[2]; }
    const double *__restrict__ r = rec + (size_t)s * seg_len * SUNS_REC;       // !LANES: the record of step i + 1 through the scalar cache
    double nxt[6];
    if (!LANES) {
#pragma unroll
        for (int k = 0; k < 6; ++k) nxt[k] = r[k];
    }
    for (int i = 0; i < n; ++i) {
        double aC, aB, aZ, aG, aT, dest;
        if (LANES) {
            aC = lane_value(r01.x, i);  aB = lane_value(r01.y, i);  aZ = lane_value(r23.x, i);  aG = lane_value(r23.y, i);
            aT = lane_value(r45.x, i);  dest = lane_value(r45.y, i);
        } else {
            aC = nxt[A_C];  aB = nxt[A_B];  aZ = nxt[A_Z];  aG = nxt[A_G];  aT = nxt[A_T];  dest = nxt[5];
            r += SUNS_REC;
#pragma unroll
            for (int k = 0; k < 6; ++k) nxt[k] = r[k];        // (the buffer carries a pad record behind the last)
        }
        const long line = __double_as_longlong(dest);
        double *o = out + line * nw + band0;
        double v[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) v[j] = dot5(aC, aB, aZ, aG, aT, b[j][0], b[j][1], b[j][2], b[j][3], b[j][4]);
        if (whole) {
            dbl2 vv;
            vv.x = v[0];
            vv.y = v[1];
            if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<dbl2 *>(o));
            else *reinterpret_cast<dbl2 *>(o) = vv;
        } else {
#pragma unroll
            for (int j = 0; j < EPL; ++j)
                if (live[j]) o[j] = v[j];
        }
    }
}

// ---- workspace: count / start [n_buckets + 1] u32 each, rank [nA] u32, segments [n_seg] uint2, records [(lines_cap + 1)][8] ----
struct SunsWorkspace {
    unsigned *count, *start, *rank;
    uint2 *seg;
    double *rec;
    long n_seg, lines_cap;
    size_t bytes;
};
size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

unsigned suns_seg_len()
{
    if (const char *v = ab_env("GORT_SUNS_SEG")) return (unsigned)atoi(v) >= 4 && atoi(v) <= 64 ? (unsigned)atoi(v) : 64u;
    return SUNS_SEG;
}
SunsWorkspace carve_suns(void *ws, long nA, long n_suns)
{
    const long SUNS_SEG = suns_seg_len();
    SunsWorkspace w;
    const size_t n_buckets = (size_t)n_suns * SUNS_CLASSES;
    w.n_seg = nA / SUNS_SEG + (long)n_buckets;                        // every bucket ends in at most one partial segment
    w.lines_cap = w.n_seg * SUNS_SEG;
    char *base = static_cast<char *>(ws);
    size_t off = 0;
    w.count = reinterpret_cast<unsigned *>(base + off);  off += align_up((n_buckets + 1) * sizeof(unsigned), 256);
    w.start = reinterpret_cast<unsigned *>(base + off);  off += align_up((n_buckets + 1) * sizeof(unsigned), 256);
    w.rank = reinterpret_cast<unsigned *>(base + off);   off += align_up((size_t)nA * sizeof(unsigned), 256);
    w.seg = reinterpret_cast<uint2 *>(base + off);       off += align_up((size_t)w.n_seg * sizeof(uint2), 256);
    w.rec = reinterpret_cast<double *>(base + off);      off += sizeof(double) * SUNS_REC * (size_t)(w.lines_cap + 1);
    w.bytes = off;
    return w;
}

}  // namespace

size_t stream_suns_workspace(long nA, long n_suns)
{
    if (nA < 1 || n_suns < 1) return 0;
    static char anchor[1];
    return carve_suns(anchor, nA, n_suns).bytes;
}

// idx_dev[line] = the place of the line's sun zenith among the n_suns distinct ones (launch_energy_table, zenith_only);
// sun_dev[n_suns][5][nw] their (sun zenith, band) terms (launch_sun_list_table); ws_dev: stream_suns_workspace(nA, n_suns) bytes.
// ev_begin / ev_end (may be null): recorded around the expansion kernel.
int launch_stream_suns(const gort_canopy *canopy_dev, const double *sun_dev, int nw, const double *angles_dev, long nA,
                       const unsigned *idx_dev, long n_suns, void *ws_dev, double *rsurf_dev, double *K_dev, void *stream,
                       void *ev_begin, void *ev_end)
{
    if (nA <= 0 || nw <= 0) return GORT_OK;
    if (n_suns < 1 || n_suns > STREAM_SUNS_MAX || nA >= (1L << 31) - 1 || nw < CHUNK)
        return fail(GORT_EINVAL, "stream (shared sun zeniths): %ld lines, %ld sun zeniths, %d bands", nA, n_suns, nw);
    hipStream_t s = (hipStream_t)stream;
    const SunsWorkspace w = carve_suns(ws_dev, nA, n_suns);
    const int n_buckets = (int)(n_suns * SUNS_CLASSES);
    const unsigned shift0 = (unsigned)((reinterpret_cast<uintptr_t>(rsurf_dev) / sizeof(double)) % SUNS_CLASSES);
    const int panels = (nw + SUNS_CLASSES - 1 + CHUNK - 1) / CHUNK;           // of the class that begins latest in its line
    const long waves = w.n_seg * panels;
    if (waves >= (1L << 31)) return fail(GORT_EINVAL, "stream (shared sun zeniths): %ld waves in one launch", waves);
    if (hipMemsetAsync(w.count, 0, (size_t)(n_buckets + 1) * sizeof(unsigned), s) != hipSuccess)
        return fail(GORT_ENODEVICE, "stream (shared sun zeniths): cannot clear the bucket counters");
    const dim3 by_lines((unsigned)((nA + SUNS_THREADS - 1) / SUNS_THREADS)), threads(SUNS_THREADS);
    hipLaunchKernelGGL(suns_rank_kernel, by_lines, threads, 0, s, nA, nw, shift0, idx_dev, w.count, w.rank);
    int rc = check_launch("suns_rank_kernel");
    if (rc) return rc;
    const unsigned seg_len = suns_seg_len();
    const char *ord = ab_env("GORT_SUNS_ORDER");
    const int seg_fastest = ord && atoi(ord) == 1;
    hipLaunchKernelGGL(suns_scan_kernel, dim3(1), dim3(SUNS_SCAN_THREADS), 0, s, w.count, n_buckets, w.start, seg_len);
    if ((rc = check_launch("suns_scan_kernel"))) return rc;
    hipLaunchKernelGGL(suns_segments_kernel, dim3((unsigned)((n_buckets + SUNS_THREADS - 1) / SUNS_THREADS)), threads, 0, s,
                       (const unsigned *)w.count, (const unsigned *)w.start, n_buckets, w.seg, seg_len);
    if ((rc = check_launch("suns_segments_kernel"))) return rc;
    hipLaunchKernelGGL(suns_geometry_kernel, by_lines, threads, 0, s, canopy_dev, angles_dev, nA, nw, shift0, idx_dev,
                       (const unsigned *)w.rank, (const unsigned *)w.start, w.rec, K_dev);
    if ((rc = check_launch("suns_geometry_kernel"))) return rc;
    if (ev_begin && hipEventRecord((hipEvent_t)ev_begin, s) != hipSuccess) return fail(GORT_ENODEVICE, "stream: cannot record an event");
    const dim3 grid((unsigned)((waves + 3) / 4));
    const unsigned lds_pad = ab_env("GORT_SUNS_LDS") ? (unsigned)atoi(ab_env("GORT_SUNS_LDS")) : 0u;      // occupancy experiment
    const bool lanes = !(ab_env("GORT_SUNS_FETCH") && ab_env("GORT_SUNS_FETCH")[0] == 's');
#define GORT_SUNS(NT, LN)                                                                                                          \
    hipLaunchKernelGGL((suns_expand_kernel<NT, LN>), grid, dim3(256), lds_pad, s, sun_dev, (const double *)w.rec, (const uint2 *)w.seg,        \
                       (const unsigned *)(w.start + n_buckets), nw, panels, make_fast_div(seg_fastest ? (unsigned)w.n_seg : (unsigned)panels), \
                       rsurf_dev, seg_len, (unsigned)w.n_seg, seg_fastest)
#ifdef GORT_AB
    if (!tuning().nt) GORT_SUNS(false, true); else if (lanes) GORT_SUNS(true, true); else GORT_SUNS(true, false);
#else
    (void)lanes;
    GORT_SUNS(true, true);
#endif
#undef GORT_SUNS
    if ((rc = check_launch("suns_expand_kernel"))) return rc;
    if (ev_end && hipEventRecord((hipEvent_t)ev_end, s) != hipSuccess) return fail(GORT_ENODEVICE, "stream: cannot record an event");
    return GORT_OK;
}

}  // namespace gort
