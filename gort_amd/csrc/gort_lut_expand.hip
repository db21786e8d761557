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

}  // namespace

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
