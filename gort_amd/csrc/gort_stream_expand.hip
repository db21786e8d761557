// gort_stream_expand.hip -- expansion of an ARBITRARY-ANGLE stream (the reference's real interface: one line per
// sun/view geometry in any order, gortt.c:232-329) from the per-line records of the geometry kernel into
// rsurf[line][band].  Every kernel here evaluates the stream family's sample (gort_device.h:
// stream_sample), so all of them write the same bits; narrow spectra take the per-sample / band-major kernels,
// wide ones the aligned flat-panel kernel (and 17 ... 255 bands the fused kernel of gort_stream_lines.hip).  (Round 3 built a second wide form - one persistent 1024-thread workgroup per
// CU with the band constants of all 2101 bands resident in the 160 KB of LDS, short row-major tasks, records staged
// through an LDS ring - bitwise equal and 8 % SLOWER at a million lines; and a third, the flat kernel's waves made persistent
// with the band constants of ONE segment of the spectrum in LDS, seven workgroups per CU - bitwise equal, 12-18 % slower:
// profiles/r03/experiments/lds_resident_stream_kernel.md.)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "gort_flat.h"

namespace gort {
namespace {

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

// The aligned flat form for ARBITRARY angle lines (every line has its own sun zenith): the chunking, the
// band-preserving stride, the PANELS (K steps x W waves, each XCD one contiguous run of panels) and the slab-edge
// handling of expand_flat_kernel; but a lane keeps the 12 band constants of its two bands in registers and forms
// the sample per step from the line's 13 LineTerms (records in layout 1, scalar loads, one step ahead): 22 instructions
// + one reciprocal per sample (gort_device.h, stream family).
// coef: stream records (GORT_COEF_STRIDE doubles), one pad record in front, tail pad behind.
// one step of a wave: the samples of the chunk from the record(s) `rec`, stored with the slab-edge handling
// a lane's two samples of a chunk: one 16-B store, but for the two edges of the output (front: the elements of chunk 0
// below first_off lie in front of it; back: those of the last chunk above last_off behind it)
template <bool NT>
__device__ __forceinline__ void store_chunk_pair(const double (&v)[EPL], bool front, bool back, int first_off, int last_off,
                                                 double *__restrict__ o, int lane)
{
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
            vv[w] = stream_sample(rec[w][0], rec[w][1], rec[w][2], rec[w][3], rec[w][4], rec[w][5], rec[w][6], rec[w][7],
                                  rec[w][8], rec[w][9], rec[w][10], rec[w][11], rec[w][12], t[j]);
        }
        v[j] = (WRAP && second[j]) ? vv[1] : vv[0];
    }
    store_chunk_pair<NT>(v, front, back, first_off, last_off, o, lane);
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

// The steps of a wave.  One record set, the first record already requested by the caller (in front of its wait for the band
// constants); the compiler places the scalar loads of step k+1 behind the arithmetic of step k and the wave waits for
// them at the top of the next step; the other waves of the SIMD fill that gap.  (One step is read beyond the last:
// expand_stream_tail_pad_records().)  A hand-made double buffer (loads of step k+1 in front of the arithmetic of step
// k, two SGPR sets) cost a wave of occupancy and ran 30 % SLOWER in round 2: this kernel lives on thread-level parallelism.
// [r6] Tried once more as the round-5 review spelled it out - request and wait written in asm as in stream_lines_kernel, two named
// SGPR sets, the wait behind the store's issue, non-wrapping waves only: the loop itself comes out clean (52 vector instructions,
// the next record in flight under them), but with 2 x 24 SGPRs pinned beside the wrapping path's 2 x 24 the allocator either drops
// to four waves per SIMD (124 VGPRs) or, held to seven, spills the wrapping path to scratch: 1 048 576 x 2101 in 4.70 ms against
// 3.02, 65 536 lines 338 us against 194-201 (profiles/r06/flat_stream_record_ahead.log, the patch beside it).  And there is little
// to win: seven waves cover the record's round trip already (7 x 240 issue cycles per round against ~800 ns of latency + 100 of
// arithmetic), the kernel sits at 0.71-0.73 beside 0.75 for its bare store pattern.  Stopped.
template <bool NT, bool WRAP>
__device__ __forceinline__ void flat_stream_loop(const StreamBand (&t)[EPL], const bool (&second)[EPL], int first_off,
                                                 int last_step, int last_off, int da, long step, int k_wave,
                                                 double (&r)[WRAP ? 2 : 1][LINE_NTERMS], const double *__restrict__ rec_w,
                                                 double *__restrict__ out_w, int lane)
{
    const long rec_step = (long)da * GORT_COEF_STRIDE;
    double *o = out_w + EPL * lane;
    for (int kk = 0; kk < k_wave; ++kk) {
        flat_stream_step<NT, WRAP>(t, second, r, kk == 0 && first_off > 0, kk == last_step, first_off, last_off, o, lane);
        rec_w += rec_step;
        load_line_terms<WRAP>(r, rec_w);
        o += step;
    }
}

// the prologue of a wave behind its index arithmetic: the two bands of every lane (twelve 16-B loads off the band table),
// the first record (scalar loads) requested while those are in flight
template <bool NT, bool WRAP>
__device__ __forceinline__ void flat_stream_wave(const StreamBand *__restrict__ bands, int nw, int band_w, int first_off,
                                                 int last_step, int last_off, int da, long step, int k_wave,
                                                 const double *__restrict__ rec_w, double *__restrict__ out_w, int lane)
{
    StreamBand t[EPL];
    bool second[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        int band = band_w + EPL * lane + j;
        second[j] = WRAP && band >= nw;                      // nw >= CHUNK on this path: one wrap at most
        if (second[j]) band -= nw;
        t[j] = bands[band];
    }
    double r[WRAP ? 2 : 1][LINE_NTERMS];
    load_line_terms<WRAP>(r, rec_w);
    flat_stream_loop<NT, WRAP>(t, second, first_off, last_step, last_off, da, step, k_wave, r, rec_w, out_w, lane);
}

// 72 VGPRs, 7 waves/SIMD.  Forcing 8 (amdgpu_waves_per_eu) spills 20 VGPRs to scratch (round 2: halved the rate).
template <bool NT>
__global__ __launch_bounds__(256) void expand_flat_stream_kernel(const StreamBand *__restrict__ bands, int nw,
                                                                  const double *__restrict__ coef, long n_total,
                                                                  int shift, long stride_chunks, int da,
                                                                  int steps_per_wave, FastDiv div_stride, FastDiv div_nw,
                                                                  double *__restrict__ out, int xcd_mode,
                                                                  XcdDuty duty, long useful_blocks,
                                                                  int *__restrict__ xcd_slots)
{
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
    const double *rec_w = coef + a_w * GORT_COEF_STRIDE;     // may point at the front pad record
    double *out_w = out + e0;
    if (band_w + CHUNK - 1 >= nw) {
        flat_stream_wave<NT, true>(bands, nw, band_w, first_off, last_step, last_off, da, step, k_wave, rec_w, out_w, lane);
    } else {
        flat_stream_wave<NT, false>(bands, nw, band_w, first_off, last_step, last_off, da, step, k_wave, rec_w, out_w, lane);
    }
}

}  // namespace

// ------------------------------------------------------------------- launchers

// nA angle lines for each of n_members members (member-major records and outputs); one thread per sample
int launch_members_stream(const gort_canopy *canopies_dev, int n_members, const double *L_dev, int nw,
                          const double *angles_dev, long nA, double *coef_dev, double *rsurf_dev, void *stream)
{
    const long n = nA * nw;
    if (n <= 0 || n_members <= 0) return GORT_OK;
    if (n_members > 65535) return fail(GORT_EINVAL, "members stream: %d members in one launch (max 65535)", n_members);
    if (stream_fuses(nw, false))
        return launch_geometry_stream_fused(canopies_dev, n_members, L_dev, nw, angles_dev, nA, rsurf_dev, nullptr, stream);
    int rc = launch_geometry_stream(canopies_dev, n_members, angles_dev, nA, coef_dev, nullptr, 0, stream, false);
    if (rc) return rc;
    hipLaunchKernelGGL(expand_stream_kernel<false>, dim3((unsigned)((n + 255) / 256), 1, (unsigned)n_members), dim3(256), 0,
                       (hipStream_t)stream, canopies_dev, L_dev, nw, coef_dev, n, rsurf_dev, (double *)nullptr, 0);
    return check_launch("expand_stream_kernel");
}

int launch_expand_grid_members(const gort_canopy *canopies_dev, const double *L_dev, int nw, const double *coef_dev,
                               long lines_per_member, int n_members, double *lut_dev, void *stream)
{
    const long n = lines_per_member * nw;
    if (n <= 0 || n_members <= 0) return GORT_OK;
    if (n_members > 65535) return fail(GORT_EINVAL, "grid expansion: %d members in one launch (max 65535)", n_members);
    if ((n + 255) / 256 >= (1L << 31)) return fail(GORT_EINVAL, "grid expansion: %ld samples per member in one launch", n);
    hipLaunchKernelGGL(expand_stream_kernel<false>, dim3((unsigned)((n + 255) / 256), 1, (unsigned)n_members), dim3(256), 0,
                       (hipStream_t)stream, canopies_dev, L_dev, nw, coef_dev, n, lut_dev, (double *)nullptr, 1);
    return check_launch("expand_stream_kernel<grid>");
}

// ---- the aligned flat form: large, wide streams without component spectra (their records are in layout 1) ----
// (>= 128 bands and >= 4M samples.  Streams of 17 ... 255 bands take gort_stream_lines.hip before they get here - the
// hand-over is GORT_LINES_MAX_BANDS - and up to 16 bands the stream is fused with the geometry unless GORT_STREAM_FUSE=0.
// Round 3's tile kernel for 17 ... 127 bands, whose rows left as whole cache lines only for multiples of 16 bands, is gone.)
bool stream_is_large(int nw, long nA, bool want_scomp)
{
    if (want_scomp || nw < CHUNK) return false;
    return nA * (long)nw >= (1L << 22);
}

// wave slots of the machine for expand_flat_stream_kernel (CUs x resident waves per CU; 256 x 28 on an MI355X)
static long stream_wave_slots()
{
    static long slots = 0;
    if (slots == 0) {
        int dev = 0, cus = 0, wgs = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgs, expand_flat_stream_kernel<true>, 256, 0) != hipSuccess || wgs < 1) wgs = 7;
        (void)hipGetLastError();
        slots = (long)cus * wgs * 4;
    }
    return slots;
}

// The panels of the flat stream kernel: W waves x K steps.  W = one band-preserving stride of about 2048 chunks (2101 for
// the 2101-band spectrum: 128 lines per step).  K from the size of the stream.  With M = chunks / wave slots steps per
// slot, three things pull: the stores of short-lived waves go faster (bare stores: 6 / 16 / 64 steps 6.3 / 6.1 / 5.6
// TB/s), the waves of a launch drift apart so its tail costs about K/2 steps whatever M, and every
// wave pays a prologue (twelve 16-B loads of band constants, ~2 us of its slot).  Measured optimum with the 28-slot
// sample and the band-table prologue (profiles/r03/stream_panel_sweep.log, us for K = 8 / 12 / 16 / 20 / 24, W = 2101):
//   8192 lines (M = 19)       31 / 34 / 33 / - / 36            131 072 lines (M = 300)    432 / 399 / 388 / - / 403
//   32 768 lines (M = 75)     105 / 103 / 110 / - / 115        262 144 lines (M = 600)    - / 817 / 775 / 780 / 791
//   65 536 lines (M = 150)    - / 200 / 209 / 209 / 213        1 048 576 lines (M = 2400) - / 3178 / 3085 / 3052 / 3060
// i.e. K = 4 log2(M) - 16 between 8 and 22 (round 2 ran 64 steps x 16 808 waves at every size: 233 us / 3.30 ms for
// 65 536 / 1 048 576 lines), and the rows cut into EQUAL panels: a ragged last panel leaves the XCD that owns it idle.
static void stream_panel_shape(int nw, long chunks, long *stride, int *steps)
{
    const ExpandTuning &tune = tuning();
    *stride = flat_stride(nw, chunks, tune.stream_waves > 0 ? tune.stream_waves : 2048);
    const long rows = (chunks + *stride - 1) / *stride;          // steps of a wave that runs through the whole stream
    long K;
    if (tune.stream_steps > 0) {
        K = tune.stream_steps;
    } else {
        const double M = (double)chunks / (double)stream_wave_slots();
        K = (long)(4.0 * log2(M > 1.0 ? M : 1.0) - 16.0 + 0.5);
        K = K < 8 ? 8 : (K > 22 ? 22 : K);
    }
    if (K > rows) K = rows;
    if (tune.stream_steps <= 0) {
        const long panels = (rows + K - 1) / K;
        K = (rows + panels - 1) / panels;
    }
    *steps = (int)K;
}

// readable records the wide expansions may touch behind the last line (the caller also keeps ONE in front)
long expand_stream_tail_pad_records(int nw, long nA)
{
    if (!stream_is_large(nw, nA, false)) return 0;
    long stride;
    int steps;
    stream_panel_shape(nw, (nA * (long)nw + 2 * CHUNK - 2) / CHUNK, &stride, &steps);
    return 2 * (stride * CHUNK / nw) + 4;       // one step of prefetch (da lines) + wrap record + slack
}

static int launch_expand_stream_flat(const double *band_table_dev, int nw, const double *coef_dev, long nA, double *rsurf_dev,
                                     int *xcd_slots_dev, hipStream_t s)
{
    const ExpandTuning &tune = tuning();
    (void)tune;                                 // (read by the measuring build only)
    if (!band_table_dev) return fail(GORT_EINVAL, "stream expansion: wide stream without the band table");
    const StreamBand *bands = reinterpret_cast<const StreamBand *>(band_table_dev);
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
    // equal XCD shares: the duty weights of the LUT kernel (28:32) change nothing here (tried in rounds 2 and 3, the
    // second time with the short panels: 26:32, 28:32, 30:32, 32:30, 32:28 against equal, 65 536 and 1 048 576 lines)
    const long nblocks = plan_xcd_duty(xcd_mode, useful, nullptr, duty);
    if (nblocks >= (1L << 31)) return fail(GORT_EINVAL, "stream expansion: %ld workgroups in one launch", nblocks);
    const dim3 grid((unsigned)nblocks);
#ifdef GORT_AB
    if (!tune.nt)
        hipLaunchKernelGGL(expand_flat_stream_kernel<false>, grid, dim3(256), 0, s, bands, nw, coef_dev, n_total, shift, stride,
                           da, steps, make_fast_div((unsigned)stride), make_fast_div((unsigned)nw), rsurf_dev, xcd_mode, duty,
                           useful, xcd_slots_dev);
    else
#endif
        hipLaunchKernelGGL(expand_flat_stream_kernel<true>, grid, dim3(256), 0, s, bands, nw, coef_dev, n_total, shift, stride,
                           da, steps, make_fast_div((unsigned)stride), make_fast_div((unsigned)nw), rsurf_dev, xcd_mode, duty,
                           useful, xcd_slots_dev);
    return check_launch("expand_flat_stream_kernel");
}

// coef_dev: stream records with ONE readable pad record in front and expand_stream_tail_pad_records() behind the last
// line; large streams (stream_is_large): records in layout 1, the flat-panel kernel.
// grid_form: the "lines" are the nodes of a few-band LUT (classic records): narrow kernels, LUT family's sample.
int launch_expand_stream(const gort_canopy *canopy_dev, const double *L_dev, const double *band_table_dev, int nw,
                         const double *coef_dev, long nA, double *rsurf_dev, double *scomp_dev, int *xcd_slots_dev, void *stream,
                         bool grid_form)
{
    const long n = nA * nw;
    if (n <= 0) return GORT_OK;
    hipStream_t s = (hipStream_t)stream;
    if (!grid_form && stream_is_large(nw, nA, scomp_dev != nullptr))
        return launch_expand_stream_flat(band_table_dev, nw, coef_dev, nA, rsurf_dev, xcd_slots_dev, s);
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

}  // namespace gort
