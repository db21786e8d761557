// gort_lut_expand.hip -- the dominant kernel of the metric grid: the LUT expansion, bound by HBM writes
// (8 B per sample).  The kernel keeps the five (sun zenith, band) numbers of its bands in registers, reads the
// five angle coefficients through the scalar cache and does nothing else but FMA + store.
// No MFMA: K=5 is not a matrix-core shape and the kernel is store-bound anyway.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gort_flat.h"

namespace gort {
namespace {

constexpr int GRID_COEF_STRIDE = 8;     // compact record: A_C..A_T + 3 pad = 64 B

// The hot kernel.  The LUT slab is one contiguous array of
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

// ---- LUTs of 33 ... 127 bands in the same absolutely aligned 1-KiB chunks (gortt.c:484-557 per node and band) ----
// Below 128 bands a chunk of 128 doubles spans up to ceil(127 / nw) + 1 nodes, so the step's record is not wave-uniform any
// more: element e of the chunk belongs to node a_w + (band_w + e) / nw.  What stays true (the stride argument at the top of this
// file holds for any nw): a lane keeps its two elements' BANDS and their node OFFSETS off_j = (band_w + 2 lane + j) / nw for
// life.  The records of all K steps of the wave - K x n_rec of them, 64 B each, the nodes a_w + k da + r - are staged in LDS by
// the prologue (two or three vector loads, waited for once, before the wave's first store), and a step reads its elements' records
// from there: LDS operations count on lgkmcnt, so no wait in the loop ever stands in front of a store (per-lane vector loads of
// the records would put a vmcnt wait - which the stores count on too - into every step).  The sun zenith of a lane's node is
// tracked per lane (its rows change once per nvza x nphi nodes, at different steps for different lanes).
// Round 5 wrote such LUTs from the geometry kernel itself, whole rows of nw doubles per store instruction - on 8-byte boundaries:
// the first and last cache line of every row shared with the neighbours, 0.60-0.71 of HBM where this pattern writes 0.78 at 128
// bands (profiles/r05/few_band_lut_pmc.log).
constexpr int FEW_MIN_BANDS = 33;            // what the kernel can do
constexpr int FEW_DEFAULT_MIN_BANDS = 65;    // what it is used for
constexpr int FEW_MAX_RECS = 5;             // ceil(127 / 33) + 1
constexpr int FEW_MAX_STEPS = 16;
constexpr int FEW_FRONT_PAD_RECORDS = 8;    // readable records in front of node 0 (the chunk that holds element 0 starts up to 127
                                            // elements = 4 nodes in front of it)

template <bool NT>
__global__ __launch_bounds__(256) void expand_flat_few_kernel(const double *__restrict__ sun, int isza_base, int n_sun,
                                                               const double *__restrict__ coef, int nw,
                                                               int angles_per_sza, long angle0, long n_total, int shift,
                                                               long stride_chunks, int da, int steps_per_wave, int n_rec,
                                                               FastDiv div_stride, FastDiv div_nw, FastDiv div_aps,
                                                               double *__restrict__ lut, int xcd_mode,
                                                               XcdDuty duty, long useful_blocks,
                                                               int *__restrict__ xcd_slots)
{
    __shared__ dbl2 s_rec[4][FEW_MAX_STEPS * FEW_MAX_RECS * 4];
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long block = xcd_logical_block(xcd_mode, duty, useful_blocks, xcd_slots);
    if (block < 0) return;
    // the wave's chunks: as in expand_flat_kernel (wave-uniform, 32-bit, divisions by multiply-shift)
    const unsigned wave = (unsigned)(block * 4 + wave_in_block);
    const unsigned stride = (unsigned)stride_chunks;
    const unsigned panel = fast_div(wave, div_stride);
    const unsigned w_in_panel = wave - panel * stride;
    const int lane = threadIdx.x & 63;
    const long step = stride_chunks * CHUNK;       // elements per step = da * nw
    const long c0 = (long)panel * steps_per_wave * stride_chunks + w_in_panel;
    const long last = n_total - 1 + shift;
    const long last_chunk = last / CHUNK;
    const int last_off = (int)(last % CHUNK);
    if (c0 > last_chunk) return;
    const long e0 = c0 * CHUNK - shift;
    const unsigned local = w_in_panel * CHUNK + (unsigned)step - (unsigned)shift;
    const unsigned a_loc = fast_div(local, div_nw);
    const int band_w = (int)(local - a_loc * (unsigned)nw);
    const long a_w = (long)panel * steps_per_wave * da + a_loc - da;          // node of the chunk's first element at step 0, >= -4
    const long rel = last_chunk - c0;
    int k_wave = steps_per_wave, last_step = -1;
    if (rel < (long)steps_per_wave * stride_chunks) {
        const unsigned k_last = fast_div((unsigned)rel, div_stride);
        k_wave = (int)k_last + 1;
        if ((unsigned)rel == k_last * stride) last_step = (int)k_last;
    }
    const int first_off = c0 == 0 ? shift : 0;

    // ---- the records of the wave's steps into LDS: slot (k n_rec + r) 4 + q = quarter q of the record of node a_w + k da + r
    dbl2 *const rec_w = s_rec[wave_in_block];
    {
        const int slots = k_wave * n_rec * 4;
        for (int s = lane; s < slots; s += 64) {
            const int kr = s >> 2, k = kr / n_rec, r = kr - k * n_rec;
            const long node = a_w + (long)k * da + r;
            rec_w[s] = reinterpret_cast<const dbl2 *>(coef + node * GRID_COEF_STRIDE)[s & 3];
        }
    }
    // ---- what a lane keeps for life: bands, node offsets, the sun zenith of each element's node and its five terms there
    double b[EPL][5];
    int band[EPL], off[EPL], isza[EPL], rem[EPL];
    auto load_sun_terms = [&](int j, int src) {          // (src: the element whose tracker holds; rows outside the table: clamped,
        int row = isza[src] - isza_base;                  //  they belong to elements in front of or behind the slab, never stored)
        row = row < 0 ? 0 : (row >= n_sun ? n_sun - 1 : row);
        const double *bp = sun + (long)row * 5 * nw + band[j];
#pragma unroll
        for (int q = 0; q < 5; ++q) b[j][q] = bp[(long)q * nw];
    };
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const unsigned pos = (unsigned)(band_w + EPL * lane + j);
        off[j] = (int)fast_div(pos, div_nw);
        band[j] = (int)(pos - (unsigned)off[j] * (unsigned)nw);
        const long A = angle0 + a_w + off[j];                                 // >= -4: only the first chunk of a slab that starts at angle 0
        long q = A >= 0 ? (A < (1L << 31) ? (long)fast_div((unsigned)A, div_aps) : A / angles_per_sza) : -((-A + angles_per_sza - 1) / angles_per_sza);
        isza[j] = (int)q;
        rem[j] = (int)(A - q * angles_per_sza);
        load_sun_terms(j, j);
    }
    // does any lane of this wave have its two elements in two nodes?  (fixed for the wave's life)
    const bool split = __any(off[1] != off[0]) != 0;
    const int at0 = off[0] * 4, at1 = off[1] * 4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                     // the records are in LDS before other lanes read them (one wave: no s_barrier)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    double *o = lut + e0 + EPL * lane;
    const int rec_stride = n_rec * 4;
    for (int k = 0; k < k_wave; ++k) {
        const dbl2 *rk = rec_w + k * rec_stride;
        const dbl2 a01 = rk[at0], a23 = rk[at0 + 1];
        const double a4 = rk[at0 + 2].x;
        double v[EPL];
        v[0] = dot5(a01.x, a01.y, a23.x, a23.y, a4, b[0][0], b[0][1], b[0][2], b[0][3], b[0][4]);
        if (split) {
            const dbl2 c01 = rk[at1], c23 = rk[at1 + 1];
            const double c4 = rk[at1 + 2].x;
            v[1] = dot5(c01.x, c01.y, c23.x, c23.y, c4, b[1][0], b[1][1], b[1][2], b[1][3], b[1][4]);
        } else {
            v[1] = dot5(a01.x, a01.y, a23.x, a23.y, a4, b[1][0], b[1][1], b[1][2], b[1][3], b[1][4]);
        }
        const bool front = k == 0 && first_off > 0, back = k == last_step;
        if (!front && !back) {
            dbl2 vv;
            vv.x = v[0];
            vv.y = v[1];
            if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<dbl2 *>(o));
            else *reinterpret_cast<dbl2 *>(o) = vv;
        } else {
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = EPL * lane + j;
                if (!(front && e < first_off) && !(back && e > last_off)) o[j] = v[j];
            }
        }
        o += step;
        // the next step's nodes: da further on; a lane whose node crosses into the next sun zenith fetches that row's terms
        if (k + 1 < k_wave) {
            bool ch[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                ch[j] = false;
                if (j == 0 || split) {
                    rem[j] += da;
                    while (rem[j] >= angles_per_sza) { rem[j] -= angles_per_sza;  ++isza[j];  ch[j] = true; }
                }
            }
            if (__any(ch[0] || ch[1])) {
                if (ch[0]) load_sun_terms(0, 0);
                if (split ? ch[1] : ch[0]) load_sun_terms(1, split ? 1 : 0);
            }
        }
    }
}

}  // namespace

bool grid_takes_few_flat_kernel(int nw, long n_total)
{
    if (nw < FEW_MIN_BANDS || nw >= CHUNK) return false;
    if (const char *v = ab_env("GORT_GRID_FEW_FLAT")) return atoi(v) != 0;        // measuring build: either form for any grid
    // from 65 bands (a hemisphere x 33 / 48 / 64 / 65 / 100 / 127 bands, fused | this: 204 | 202, 247 | 272-287, 285-302 | 327,
    // 309-356 | 317-326, 492 | 427, 668 | 508 us: profiles/r06/few_band_flat_ab.log) and for grids worth three launches
    return nw >= FEW_DEFAULT_MIN_BANDS && n_total >= (1L << 23);
}

static int few_steps(int xcd_mode)
{
    int steps = tuning().steps;
    if (steps <= 0) steps = xcd_mode == 2 ? 16 : 6;
    return steps > FEW_MAX_STEPS ? FEW_MAX_STEPS : steps;
}

// readable records in front of node 0 and behind the last node for expand_flat_few_kernel
void expand_grid_few_pad_records(int nw, long n_total, long *front, long *tail)
{
    (void)nw;  (void)n_total;
    *front = FEW_FRONT_PAD_RECORDS;
    *tail = 2 * FEW_MAX_RECS + 6;          // a wave stages the records of the steps it has inside the slab: n_rec - 1 nodes past its last one at most
}

int launch_expand_grid_few(const double *sun_dev, int isza_base, int n_sun, const double *coef_dev, int nw, int nvza, int nphi,
                           long row_begin, long row_end, double *lut_dev, int *xcd_slots_dev, const int *xcd_weights, void *stream)
{
    const long rows = row_end - row_begin;
    if (rows <= 0) return GORT_OK;
    if (nw < FEW_MIN_BANDS || nw >= CHUNK) return fail(GORT_EINVAL, "expand_grid_few: %d bands", nw);
    const ExpandTuning &tune = tuning();
    const long n_total = rows * nphi * (long)nw;
    const int shift = (int)((reinterpret_cast<uintptr_t>(lut_dev) / sizeof(double)) % CHUNK);
    const long chunks = (n_total + shift + CHUNK - 1) / CHUNK;
    const long stride = flat_stride(nw, chunks, tune.waves);
    const int xcd_mode = resolve_xcd_mode(xcd_slots_dev);
    const int steps = few_steps(xcd_mode);
    const long panels = (chunks + (long)steps * stride - 1) / ((long)steps * stride);
    if (panels * stride >= (1L << 31) || chunks >= (1L << 31) || stride * CHUNK >= (1L << 30))
        return fail(GORT_EINVAL, "expand_grid_few: slab of %ld chunks in %ld waves is beyond the kernel's 32-bit indices", chunks, panels * stride);
    const int da = (int)(stride * CHUNK / nw);
    const int n_rec = (CHUNK - 1 + nw - 1) / nw + 1;               // nodes a chunk can touch: ceil(127 / nw) + 1
    if (n_rec > FEW_MAX_RECS) return fail(GORT_EINVAL, "expand_grid_few: %d records per chunk", n_rec);
    const long useful = (panels * stride + 3) / 4;
    XcdDuty duty;
    const long nblocks = plan_xcd_duty(xcd_mode, useful, xcd_weights, duty);
    if (nblocks >= (1L << 31)) return fail(GORT_EINVAL, "expand_grid_few: %ld workgroups in one launch", nblocks);
    const int angles_per_sza = nvza * nphi;
    const long angle0 = row_begin * nphi;
#define GORT_FEW(N)                                                                                                              \
    hipLaunchKernelGGL((expand_flat_few_kernel<N>), dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, sun_dev, isza_base, \
                       n_sun, coef_dev, nw, angles_per_sza, angle0, n_total, shift, stride, da, steps, n_rec,                   \
                       make_fast_div((unsigned)stride), make_fast_div((unsigned)nw), make_fast_div((unsigned)angles_per_sza),   \
                       lut_dev, xcd_mode, duty, useful, xcd_slots_dev)
#ifdef GORT_AB
    if (!tune.nt) GORT_FEW(false); else
#endif
    GORT_FEW(true);
#undef GORT_FEW
    return check_launch("expand_flat_few_kernel");
}

// records the LUT kernel may read past the last angle (prefetch depth x angles per step, + wrap, + slack)
long expand_grid_tail_pad_records(int nw, long n_total)
{
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
#ifdef GORT_AB
    if (tune.nt) {
        if (tune.depth == 1) GORT_FLAT(1, true); else if (tune.depth == 2) GORT_FLAT(2, true); else GORT_FLAT(4, true);
    } else {
        if (tune.depth == 1) GORT_FLAT(1, false); else if (tune.depth == 2) GORT_FLAT(2, false); else GORT_FLAT(4, false);
    }
#else
    GORT_FLAT(2, true);                 // the product's one form: two records in flight per lane, non-temporal stores
#endif
#undef GORT_FLAT
    return check_launch("expand_flat_kernel");
}

}  // namespace gort
