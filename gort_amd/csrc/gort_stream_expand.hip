// gort_stream_expand.hip -- expansion of an ARBITRARY-ANGLE stream (the reference's real interface: one line per
// sun/view geometry in any order, gortt.c:232-329) from the per-line records of the geometry kernel into
// rsurf[line][band].  Every kernel here evaluates the stream family's sample (gort_device.h: sun_pair +
// stream_sample), so all of them write the same bits; narrow spectra take the per-sample / band-major kernels,
// wide ones the aligned flat kernels.
#include <hip/hip_runtime.h>

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
// p_df, t'_df and the sample per step from the line's 13 LineTerms (records in layout 1, scalar loads, one step
// ahead): ~24 instructions + one fp64 division per sample (gort_device.h, stream family).
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
            double pdf, tpdf;
            sun_pair(t[j], rec[w][9], rec[w][10], rec[w][11], rec[w][12], pdf, tpdf);
            vv[w] = stream_sample(rec[w][0], rec[w][1], rec[w][2], rec[w][3], rec[w][4], rec[w][5], rec[w][6], rec[w][7],
                                  rec[w][8], t[j], pdf, tpdf);
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

// ---- the LDS-resident form ------------------------------------------------------------------------------------
// What limits expand_flat_stream_kernel is the shape of its waves, not its arithmetic: a wave must live for ~64 steps
// to pay for deriving 24 band constants from global memory (a ~3 us dependent chain during which its slot stores
// nothing), and 64-step panels write 13 % slower than the 6-step panels of the LUT kernel even as bare stores
// (DESIGN.md 5.5).  gfx950 has 160 KB of LDS per CU - enough for the band constants of ALL 2101 bands - so here
//   * ONE 1024-thread workgroup per CU stays resident for the whole launch and keeps the band table
//     tab[9][nw] (gam, omega, Rff, Tff, tff, pff, rs, mgk, B: 151 KB for nw = 2101) in LDS;
//   * work is cut into TASKS of 16 adjacent columns (the 16 waves of the workgroup: 16 KiB contiguous per step)
//     x K steps (K = 16), handed out in row-major order of the output, so that the machine sweeps a compact window
//     as the LUT kernel's short panels do; a task switch costs 18 LDS reads + ~10 instructions per lane instead of
//     a round trip to L2/HBM;
//   * the line records of a task (2 lines per step for nw = 2101: 16 columns span 2048 elements) are staged by the
//     whole workgroup - one double per thread, fetched one task ahead into a register, parked in an LDS ring - and
//     reach the lanes as LDS broadcasts: no scalar loads in the loop, and the line terms arrive in VGPRs, which
//     lifts the one-SGPR-per-VOP3 limit off the sample's FMAs;
//   * one workgroup barrier per task keeps the 16 waves - the 16 KiB they write per step - together.
// The arithmetic is flat_stream_step() of the flat kernel on the same records and the same band constants (Zf, Tf
// re-derived from tff and mgk exactly as lambda_table_kernel forms them): the same bits.
constexpr int LDS_THREADS = 1024;
constexpr int LDS_WAVES = LDS_THREADS / 64;
constexpr int LDS_TAB_SLOTS = 9;
constexpr int LDS_REC = 14;              // doubles per staged record: 13 LineTerms + 1 pad (rows stay 16-B aligned)
constexpr size_t LDS_MAX_BYTES = 160u * 1024u;

// One step of a wave's task.  The line terms come out of the LDS ring: in waves whose chunk lies inside one line a
// broadcast of that line's record; in the ~6 % of waves whose band index wraps inside the chunk each element reads the
// record of ITS line (a per-lane LDS address: two distinct rows, no bank conflict) - no second sample, no select.
// MODE (experiments only, GORT_STREAM_LDS_MODE): 0 = the kernel; 1 = arithmetic without stores; 2 = stores of a trivial value;
// 3 = no workgroup barrier between tasks (races on the record ring: wrong results, timing only)
template <bool NT, bool WRAP, int MODE>
__device__ __forceinline__ void lds_step(const StreamBand (&t)[EPL], const bool (&second)[EPL], const double *ring, bool front,
                                         bool back, int first_off, int last_off, double *__restrict__ o, int lane)
{
    double v[EPL];
    if (MODE == 2) {
        v[0] = ring[0] + t[0].B;
        v[1] = ring[0] + t[1].B;
    } else if (!WRAP) {
        double r[LINE_NTERMS];
#pragma unroll
        for (int q = 0; q < LINE_NTERMS; ++q) r[q] = ring[q];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            double pdf, tpdf;
            sun_pair(t[j], r[9], r[10], r[11], r[12], pdf, tpdf);
            v[j] = stream_sample(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], t[j], pdf, tpdf);
        }
    } else {
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const double *rj = ring + (second[j] ? LDS_REC : 0);
            double r[LINE_NTERMS];
#pragma unroll
            for (int q = 0; q < LINE_NTERMS; ++q) r[q] = rj[q];
            double pdf, tpdf;
            sun_pair(t[j], r[9], r[10], r[11], r[12], pdf, tpdf);
            v[j] = stream_sample(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], t[j], pdf, tpdf);
        }
    }
    if (MODE == 1) {
        if (v[0] == -12345.678 && v[1] == 9.87e300) o[0] = v[0];      // never true: keeps the arithmetic alive
    } else {
        store_chunk_pair<NT>(v, front, back, first_off, last_off, o, lane);
    }
}

// the steps of a task at the edges of the output, or cut short by its end: any count, edge handling per step
template <bool NT, bool WRAP, int MODE>
__device__ __forceinline__ void lds_task_steps(const StreamBand (&t)[EPL], const bool (&second)[EPL], int first_off,
                                               int last_step, int last_off, long step, int k_wave,
                                               const double *ring, int ring_step, double *__restrict__ out_w, int lane)
{
    double *o = out_w + EPL * lane;
#pragma unroll 1
    for (int kk = 0; kk < k_wave; ++kk) {
        lds_step<NT, WRAP, MODE>(t, second, ring, kk == 0 && first_off > 0, kk == last_step, first_off, last_off, o, lane);
        ring += ring_step;
        o += step;
    }
}

// the steps of an interior task: exactly KFULL steps of exactly one vector store each, unrolled - straight-line code in
// which the compiler can COUNT the stores.  That matters for more than loop overhead: the staged records of the next
// task are a vector load issued in front of these stores, loads and stores share one in-order counter on gfx9, and
// behind a loop of unknown length the compiler can only wait for `vmcnt(0)` before the value is used - i.e. for
// every store of the task to be acknowledged by memory (~2 us, once per task and wave); here it waits for vmcnt(KFULL).
template <bool NT, bool WRAP, int MODE, int KFULL>
__device__ __forceinline__ void lds_task_full(const StreamBand (&t)[EPL], const bool (&second)[EPL], long step, const double *ring,
                                              int ring_step, double *__restrict__ out_w, int lane)
{
    double *o = out_w + EPL * lane;
#pragma unroll
    for (int kk = 0; kk < KFULL; ++kk) {
        lds_step<NT, WRAP, MODE>(t, second, ring, false, false, 0, 0, o, lane);
        ring += ring_step;
        o += step;
    }
}

template <bool NT, int MODE, int KFULL>
__global__ __launch_bounds__(LDS_THREADS) void expand_stream_lds_kernel(
    const gort_canopy *__restrict__ canopy, const double *__restrict__ L, int nw, const double *__restrict__ coef, long nA,
    long n_total, int shift, long stride_chunks, int da, int K, int npl, int groups, long n_tasks,
    FastDiv div_nw, FastDiv div_groups, double *__restrict__ out, int xcd_split)
{
    extern __shared__ double s_mem[];
    // the tasks of this workgroup: task = (row block jb, column group cg), jb-major
    long task, task_end, task_step;
    if (xcd_split) {                 // every XCD one contiguous eighth of the tasks, dealt round-robin to its workgroups
        const long x = blockIdx.x & 7, q = blockIdx.x >> 3, nq = gridDim.x >> 3;
        task = n_tasks * x / 8 + q;
        task_end = n_tasks * (x + 1) / 8;
        task_step = nq;
    } else {
        task = blockIdx.x;
        task_end = n_tasks;
        task_step = gridDim.x;
    }
    if (task >= task_end) return;                       // the whole workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tab_doubles = (LDS_TAB_SLOTS * nw + 1) & ~1;
    double *s_tab = s_mem;                              // [9][nw]
    double *s_ring = s_mem + tab_doubles;               // [2][K][npl][LDS_REC] + a dump row of 64
    const int ring_doubles = K * npl * LDS_REC;
    for (int i = tid; i < 8 * nw; i += LDS_THREADS) s_tab[i] = L[i];                       // L slots 0..7 = gam .. mgk
    for (int i = tid; i < nw; i += LDS_THREADS) s_tab[8 * nw + i] = L[L_B * nw + i];
    const double kep = canopy->k_openep, kopen = canopy->k_open + canopy->k_openep;       // as lambda_table_kernel

    const unsigned stride = (unsigned)stride_chunks;
    const long step = stride_chunks * CHUNK;            // elements per step = da * nw
    const long last = n_total - 1 + shift;
    const long last_chunk = last / CHUNK;
    const int last_off = (int)(last % CHUNK);

    // this thread's share of a task's records: double q of the record of line a_min + k da + p.  EVERY thread loads
    // and parks a value for every task (threads beyond the K npl 13 needed ones re-read element 0 and park it in a
    // dump row behind the ring; a workgroup without a next task re-reads its last one): unconditional, so that the
    // compiler sees one load and one use per task on every path and never has to guess what is still pending
    const int n_stage = K * npl * LINE_NTERMS;
    const bool stages = tid < n_stage;
    int st_line = 0, st_q = 0, st_ring = 0;
    if (stages) {
        const int k = tid / (npl * LINE_NTERMS), r = tid - k * (npl * LINE_NTERMS), p = r / LINE_NTERMS;
        st_q = r - p * LINE_NTERMS;
        st_line = k * da + p;
        st_ring = (k * npl + p) * LDS_REC + st_q;
    }
    // first line of a task: that of its first column's chunk at step 0 (>= -1: the pad record in front)
    auto task_first_line = [&](long tk, unsigned &jb, unsigned &cg) -> long {
        jb = fast_div((unsigned)tk, div_groups);
        cg = (unsigned)tk - jb * (unsigned)groups;
        const unsigned local0 = cg * LDS_WAVES * CHUNK + (unsigned)step - (unsigned)shift;
        return (long)jb * K * da + fast_div(local0, div_nw) - da;
    };
    auto fetch = [&](long tk) -> double {
        unsigned jb, cg;
        long line = task_first_line(tk, jb, cg) + st_line;
        if (line > nA) line = nA;                       // beyond the stream: a pad record, never used by a stored element
        return coef[line * GORT_COEF_STRIDE + st_q];
    };
    // ring offset of this thread's value in buffer b: staging threads into the ring, the others into the dump row
    auto park_at = [&](int b) -> int { return stages ? b * ring_doubles + st_ring : 2 * ring_doubles + lane; };
    s_ring[park_at(0)] = fetch(task);
    __syncthreads();

    int buf = 0;
    for (; task < task_end; task += task_step, buf ^= 1) {
        const bool more = task + task_step < task_end;
        const double nxt = fetch(more ? task + task_step : task);         // in flight while this task is worked on
        const int park = park_at(buf ^ 1);
        bool parked = false;
        unsigned jb, cg;
        const long a_min = task_first_line(task, jb, cg);
        const unsigned w = cg * LDS_WAVES + (unsigned)wave;               // this wave's column
        const long c0 = (long)jb * K * stride_chunks + w;
        if (__builtin_amdgcn_readfirstlane((int)(w < stride && c0 <= last_chunk))) {
            // index arithmetic of expand_flat_stream_kernel with panel = jb, steps_per_wave = K
            const long e0 = c0 * CHUNK - shift;
            const unsigned local = w * CHUNK + (unsigned)step - (unsigned)shift;
            const unsigned a_loc = fast_div(local, div_nw);
            // (everything here is wave-uniform; the readfirstlanes say so to the compiler, which otherwise predicates the
            // branches below instead of jumping - and then cannot count the stores between a load and its use)
            const int band_w = __builtin_amdgcn_readfirstlane((int)(local - a_loc * (unsigned)nw));
            const long a_w = (long)jb * K * da + a_loc - da;
            const long rel = last_chunk - c0;
            int k_wave = K, last_step = -1;
            if (rel < (long)K * stride_chunks) {             // only the last row block's waves run out of stream
                const unsigned k_last = (unsigned)(rel / stride_chunks);
                k_wave = (int)k_last + 1;
                if ((unsigned long)rel == (unsigned long)k_last * stride) last_step = (int)k_last;
            }
            k_wave = __builtin_amdgcn_readfirstlane(k_wave);
            last_step = __builtin_amdgcn_readfirstlane(last_step);
            const int first_off = __builtin_amdgcn_readfirstlane(c0 == 0 ? shift : 0);
            StreamBand t[EPL];
            bool second[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
#pragma clang fp contract(off)
                int band = band_w + EPL * lane + j;
                second[j] = band >= nw;
                if (second[j]) band -= nw;
                BandTerms bt;
                bt.gam = s_tab[band];            bt.omega = s_tab[nw + band];      bt.Rff = s_tab[2 * nw + band];
                bt.Tff = s_tab[3 * nw + band];   bt.tff = s_tab[4 * nw + band];    bt.pff = s_tab[5 * nw + band];
                bt.rs = s_tab[6 * nw + band];    bt.mgk = s_tab[7 * nw + band];    bt.B = s_tab[8 * nw + band];
                const double tpff = tpff_of(bt.tff, kopen);
                bt.Zf = (tpff - kep) * bt.rs;
                bt.Tf = tpff * bt.mgk;
                t[j] = stream_band(bt);
            }
            const double *ring = s_ring + buf * ring_doubles + __builtin_amdgcn_readfirstlane((int)(a_w - a_min)) * LDS_REC;
            double *out_w = out + e0;
            const bool wraps = band_w + CHUNK - 1 >= nw;
            if (K == KFULL && k_wave == KFULL && first_off == 0 && last_step < 0) {
                // the next task's records are parked in the ring right behind the stores, in the same block of
                // straight-line code: a counted wait (vmcnt(KFULL)), not a wait for the stores themselves
                // (the asm comments keep the three identical ring writes from being merged into one block, which would be
                // entered from the counted and the uncounted paths alike)
                if (wraps) {
                    lds_task_full<NT, true, MODE, KFULL>(t, second, step, ring, npl * LDS_REC, out_w, lane);
                    asm volatile("; records of the next task parked behind KFULL counted stores (wrap)" ::: "memory");
                    s_ring[park] = nxt;
                    asm volatile("; parked (wrap)" ::: "memory");
                } else {
                    lds_task_full<NT, false, MODE, KFULL>(t, second, step, ring, npl * LDS_REC, out_w, lane);
                    asm volatile("; records of the next task parked behind KFULL counted stores" ::: "memory");
                    s_ring[park] = nxt;
                    asm volatile("; parked" ::: "memory");
                }
                parked = true;
            } else if (wraps) {
                lds_task_steps<NT, true, MODE>(t, second, first_off, last_step, last_off, step, k_wave, ring, npl * LDS_REC, out_w, lane);
            } else {
                lds_task_steps<NT, false, MODE>(t, second, first_off, last_step, last_off, step, k_wave, ring, npl * LDS_REC, out_w, lane);
            }
        }
        if (!parked) s_ring[park] = nxt;
        if (MODE != 3) __syncthreads();                 // MODE 3 (experiment, WRONG results): how much the lockstep costs
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
    int rc = launch_geometry_stream(canopies_dev, n_members, angles_dev, nA, coef_dev, nullptr, 0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(expand_stream_kernel<false>, dim3((unsigned)((n + 255) / 256), 1, (unsigned)n_members), dim3(256), 0,
                       (hipStream_t)stream, canopies_dev, L_dev, nw, coef_dev, n, rsurf_dev, (double *)nullptr, 0);
    return check_launch("expand_stream_kernel");
}

// ---- aligned flat forms: large, wide streams without component spectra (their records are in layout 1) ----
bool stream_is_wide(int nw, long nA, bool want_scomp)
{
    return !want_scomp && nw >= CHUNK && nA * (long)nw >= (1L << 22);
}

// panel shape of the per-line flat kernel: stride W (chunks) and steps K per wave
static void stream_panel_shape(int nw, long chunks, long *stride, int *steps)
{
    const ExpandTuning &tune = tuning();
    // small streams: fewer waves, so that a wave still has ~6 steps to spread its prologue (24 band constants per lane)
    // over - 3000 lines: 20 us with 8404 waves, 36 us with 33616; 8192 lines: 37 against 43
    long target = tune.stream_waves;
    if (chunks / 6 < target) target = chunks / 6 < 4202 ? 4202 : chunks / 6;
    *stride = flat_stride(nw, chunks, target);
    *steps = tune.stream_steps;
}

// shape of the LDS-resident form: 16-column groups of a ~2048-chunk stride, K steps per task; false = not applicable
// (band table + record ring beyond the 160 KB of LDS: nw > ~2130)
struct LdsShape {
    long stride, n_tasks;
    int K, npl, groups, da;
    size_t lds_bytes;
};
static bool stream_lds_shape(int nw, long chunks, LdsShape *s)
{
    static const int env_k = getenv("GORT_STREAM_LDS_STEPS") ? atoi(getenv("GORT_STREAM_LDS_STEPS")) : 0;
    s->stride = flat_stride(nw, chunks, 2048);
    s->da = (int)(s->stride * CHUNK / nw);
    s->npl = 2 + (LDS_WAVES * CHUNK - 2) / nw;           // lines a 16-column step can touch
    const long rows = (chunks + s->stride - 1) / s->stride;      // steps per column over the whole stream
    int K = env_k > 0 ? env_k : 16;
    const int k_max = K;                                 // what the LDS budget is checked with: the verdict must not depend on the stream's length
    // short streams: smaller tasks, so that every workgroup still gets a few dozen of them
    while (!env_k && K > 4 && (rows / K) * ((s->stride + LDS_WAVES - 1) / LDS_WAVES) < 24 * 256) K /= 2;
    while (K > 1 && K * s->npl * LINE_NTERMS > LDS_THREADS) --K;
    if (K * s->npl * LINE_NTERMS > LDS_THREADS) return false;
    s->K = K;
    s->groups = (int)((s->stride + LDS_WAVES - 1) / LDS_WAVES);
    s->n_tasks = ((rows + K - 1) / K) * s->groups;
    s->lds_bytes = sizeof(double) * (size_t)(((LDS_TAB_SLOTS * nw + 1) & ~1) + 2 * K * s->npl * LDS_REC + 64);
    const size_t worst = sizeof(double) * (size_t)(((LDS_TAB_SLOTS * nw + 1) & ~1) + 2 * k_max * s->npl * LDS_REC + 64);
    return worst <= LDS_MAX_BYTES && s->n_tasks < (1L << 31) && (long)s->groups * LDS_WAVES * CHUNK < (1L << 30);
}

bool stream_lds_applies(int nw, long nA)
{
    LdsShape s;
    return stream_lds_shape(nw, (nA * (long)nw + 2 * CHUNK - 2) / CHUNK, &s);
}

// readable records the wide expansions may touch behind the last line (the caller also keeps ONE in front)
long expand_stream_tail_pad_records(int nw, long nA)
{
    if (!stream_is_wide(nw, nA, false)) return 0;
    long stride;
    int steps;
    stream_panel_shape(nw, (nA * (long)nw + 2 * CHUNK - 2) / CHUNK, &stride, &steps);
    return 2 * (stride * CHUNK / nw) + 4;       // one step of prefetch (da lines) + wrap record + slack
}

static int launch_expand_stream_flat(const double *L_dev, int nw, const double *coef_dev, long nA, double *rsurf_dev,
                                     int *xcd_slots_dev, hipStream_t s)
{
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
                           useful, xcd_slots_dev);
    else
        hipLaunchKernelGGL(expand_flat_stream_kernel<false>, grid, dim3(256), 0, s, L_dev, nw, coef_dev, n_total, shift, stride,
                           da, steps, make_fast_div((unsigned)stride), make_fast_div((unsigned)nw), rsurf_dev, xcd_mode, duty,
                           useful, xcd_slots_dev);
    return check_launch("expand_flat_stream_kernel");
}

// workgroups of the persistent form: one per CU (a workgroup owns a CU's LDS), a multiple of 8 for the XCD split
static int lds_workgroups()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus < 8)
            cus = 8;
        if (const char *v = getenv("GORT_STREAM_LDS_WGS")) cus = atoi(v) >= 8 ? atoi(v) : cus;
        n = cus / 8 * 8;
    }
    return n;
}

static int launch_expand_stream_lds(const gort_canopy *canopy_dev, const double *L_dev, int nw, const double *coef_dev,
                                    long nA, double *rsurf_dev, hipStream_t s)
{
    static const bool xcd_split = !(getenv("GORT_STREAM_LDS_SPLIT") && atoi(getenv("GORT_STREAM_LDS_SPLIT")) == 0);
    const long n_total = nA * (long)nw;
    const int shift = (int)((reinterpret_cast<uintptr_t>(rsurf_dev) / sizeof(double)) % CHUNK);
    const long chunks = (n_total + shift + CHUNK - 1) / CHUNK;
    LdsShape sh;
    if (!stream_lds_shape(nw, chunks, &sh)) return fail(GORT_EINVAL, "stream expansion: %d bands do not fit the LDS-resident form", nw);
    if (chunks >= (1L << 31) || sh.stride * CHUNK >= (1L << 30))
        return fail(GORT_EINVAL, "stream expansion: %ld chunks are beyond the kernel's 32-bit indices", chunks);
    static const int mode = getenv("GORT_STREAM_LDS_MODE") ? atoi(getenv("GORT_STREAM_LDS_MODE")) : 0;     // experiments
    const bool nt = tuning().nt;
    typedef void (*kern_t)(const gort_canopy *, const double *, int, const double *, long, long, int, long, int, int, int, int, long,
                           FastDiv, FastDiv, double *, int);
    const bool k8 = sh.K == 8;                                // interior tasks unrolled for K = 16 (default) or 8 (short streams)
    const kern_t fn = mode == 1 ? (kern_t)expand_stream_lds_kernel<true, 1, 16>
                      : mode == 2 ? (kern_t)expand_stream_lds_kernel<true, 2, 16>
                      : mode == 3 ? (kern_t)expand_stream_lds_kernel<true, 3, 16>
                      : nt ? (k8 ? (kern_t)expand_stream_lds_kernel<true, 0, 8> : (kern_t)expand_stream_lds_kernel<true, 0, 16>)
                           : (k8 ? (kern_t)expand_stream_lds_kernel<false, 0, 8> : (kern_t)expand_stream_lds_kernel<false, 0, 16>);
    // the full 160 KB of a CU for one workgroup (harmless where the default limit already allows it)
    (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX_BYTES);
    (void)hipGetLastError();
    long wgs = lds_workgroups();
    const bool split = xcd_split && sh.n_tasks >= 8 * wgs;
    if (!split && sh.n_tasks < wgs) wgs = sh.n_tasks;
    hipLaunchKernelGGL(fn, dim3((unsigned)wgs), dim3(LDS_THREADS), sh.lds_bytes, s, canopy_dev, L_dev, nw, coef_dev, nA, n_total,
                       shift, sh.stride, sh.da, sh.K, sh.npl, sh.groups, sh.n_tasks, make_fast_div((unsigned)nw),
                       make_fast_div((unsigned)sh.groups), rsurf_dev, split ? 1 : 0);
    return check_launch("expand_stream_lds_kernel");
}

// coef_dev: stream records with ONE readable pad record in front and expand_stream_tail_pad_records() behind the last
// line; wide streams (stream_is_wide): records in layout 1, wide_form 1 = flat panels, 2 = LDS-resident.
// grid_form: the "lines" are the nodes of a few-band LUT (classic records): narrow kernels, LUT family's sample.
int launch_expand_stream(const gort_canopy *canopy_dev, const double *L_dev, int nw, const double *coef_dev, long nA,
                         double *rsurf_dev, double *scomp_dev, int *xcd_slots_dev, int wide_form, void *stream,
                         bool grid_form)
{
    const long n = nA * nw;
    if (n <= 0) return GORT_OK;
    hipStream_t s = (hipStream_t)stream;
    if (!grid_form && stream_is_wide(nw, nA, scomp_dev != nullptr)) {
        if (wide_form == 2) return launch_expand_stream_lds(canopy_dev, L_dev, nw, coef_dev, nA, rsurf_dev, s);
        return launch_expand_stream_flat(L_dev, nw, coef_dev, nA, rsurf_dev, xcd_slots_dev, s);
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

}  // namespace gort
