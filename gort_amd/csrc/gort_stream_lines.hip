// gort_stream_lines.hip -- streams of 17 ... LINES_MAX_BANDS bands, the range the reference's own command line can
// ingest (its header line holds ~190 wavelengths, gortt.c:153-184): ONE kernel from the angle line to its row of
// reflectances (the per-line loop of main(), gortt.c:232-329, with gortt_rsurf, gortt.c:385-578).
//
// A wave takes 64 consecutive LINES, lane = line.  The lane evaluates its line's geometry (gort_geometry.h) and keeps
// the thirteen LineTerms of the stream family's sample in registers - no 128-B record per line written and read back -
// then walks the bands: their twelve constants are wave-uniform (scalar loads off the StreamBand table), so a sample
// costs its 28 fp64 issue slots and nothing else.
//
// The output of the wave's 64 lines is ONE contiguous span of 64 nw doubles.  A band's 64 samples belong to 64
// different rows of it, nw doubles apart, and rows start on 8-byte boundaries only: round 3's tile kernel stored 16-band
// pieces row by row, which are whole 128-B lines only where nw is a multiple of 16 (everywhere else every cache line
// was written in two halves a tile apart: half the rate, 1.3 x the traffic).  Here every lane has a RING of 32 doubles
// in LDS indexed by the ABSOLUTE position of the sample in the output (mod 32): after band block k the cache line
// number k-1 of every row is complete in its ring whatever the row's alignment, and leaves as one 128-B line, 16 B per
// lane, eight rows per store instruction.  The cache line a row shares with the next one (its tail + the next row's
// head) is completed at the end: every lane keeps the 16 samples of its first band block in registers and drops its
// head into the ring of the row in front.  So only the two ends of a wave's span can be partial lines, and with an
// output that starts on a 128-B boundary only those of the whole stream (64 nw doubles are a multiple of 16).
//
// Same functions on the same numbers as every other stream kernel (geometry_core, store_coef, line_terms_of_record,
// stream_sample): the same bits (tests/test_stream_forms.py).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "gort_geometry.h"

#include "gort_stamps.h"

GORT_STAMPS_DEFINE(lines)

namespace gort {
namespace {

constexpr int RING = 32;                        // doubles per line in flight: the band block being written + the one before
constexpr int BLOCK_BANDS = 16;                 // one cache line per row and block

// The twelve constants of a band, wave-uniform, through the scalar cache.  Left to the compiler a band's s_load sits
// directly in front of the s_waitcnt for it (it sinks the load of a value used in the next iteration to the end of the
// loop body whatever the source says), which at the two waves per SIMD this kernel's LDS rings
// allow is ~150 exposed cycles per 112 of arithmetic.  So the request and the wait are written out: two register sets,
// the band after next requested while this one is evaluated.  The wait takes the registers as operands, so nothing that
// reads them can move in front of it; a set is always waited for before it dies (the loads land whenever they land).
typedef double double8v __attribute__((ext_vector_type(8)));
typedef double double4v __attribute__((ext_vector_type(4)));
struct BandRegs { double8v lo; double4v hi; };
__device__ __forceinline__ void band_request(BandRegs &r, const StreamBand *p)
{
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40" : "=&s"(r.lo), "=&s"(r.hi) : "s"(p));
    __builtin_amdgcn_sched_barrier(0);          // or the scheduler lets the arithmetic that follows in the source go first
}
__device__ __forceinline__ void band_wait(BandRegs &r)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.lo), "+s"(r.hi) : : "memory");
}
__device__ __forceinline__ StreamBand band_of(const BandRegs &r)
{
    StreamBand b;
    b.g2 = r.lo[0];  b.c1 = r.lo[1];  b.c2 = r.lo[2];  b.Rff = r.lo[3];  b.tff = r.lo[5];  b.pff = r.lo[6];
    b.rs = r.lo[7];  b.mgk = r.hi[0];  b.Zf = r.hi[1];  b.Tf = r.hi[2];  b.B = r.hi[3];
    // cT - c2 t0 (stream_sample's n2) reads TWO band constants, and a vector instruction of gfx9 reads one scalar operand: the
    // compiler copies cT into vector registers with two 32-bit moves, each an issue slot of four cycles like a whole fp64 FMA;
    // the 64-bit move is one
    const double cT = r.lo[4];
    asm("v_mov_b64 %0, %1" : "=v"(b.cT) : "s"(cT));
    return b;
}

// the cache line at position X (relative to the wave's 256-B aligned origin, a multiple of 16) of ring row `row`:
// lane q of eight moves 16 bytes; positions outside [lo, hi) are not the wave's to write
template <bool NT>
__device__ __forceinline__ void emit_cache_line(const double *__restrict__ ring, int row, int pitch, int X, int lo, int hi, int q,
                                                double *__restrict__ origin)
{
    const dbl2 v = reinterpret_cast<const dbl2 *>(ring)[((row * pitch + (X & (RING - 1))) >> 1) + q];      // all even: 16-B aligned
    const int p = X + 2 * q;
    double *o = origin + p;
    if (hi - lo == 16) {
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(o));
        else *reinterpret_cast<dbl2 *>(o) = v;
    } else {
        if (p >= lo && p < hi) o[0] = v.x;
        if (p + 1 >= lo && p + 1 < hi) o[1] = v.y;
    }
}

// What a member's waves read of its StreamBand table (96 B per band: 200 KB at 2101 bands, another for every member) a little
// ahead of the scalar loads, into the XCD's L2: 32 consecutive 128-B lines per instruction, a dword each, straight into LDS
// (global_load_lds: no destination register; nothing ever waits for them or reads what they bring).  One canopy's table is hot in
// every L2 after the first waves; with a thousand members every wave found its band constants in HBM (or the Infinity Cache) and
// the band loop - one request ahead, ~150 ns of cover - waited: 69 % of a wave's life at 2101 bands, and every table was fetched
// eight times, once per XCD (FETCH_SIZE 1.6 GB per launch; profiles/r06/members_stream_counters.log).
// WHERE in LDS: ring row 0 - the cache line in front of the wave's first row - holds that row's head in ONE of its two
// 16-double halves and is read there only; the other half is dead for the whole life of the wave: 128 bytes, 32 lanes.  (256 bytes
// of their own behind the rings cost a wave per CU wherever the pitch is 34: eight instead of nine.)
struct TouchRange {
    const char *base;       // the first byte's 128-B line
    int n_lines;            // lines up to the one of the last byte
    int lead;               // bytes between `base` and the first byte
};
__device__ __forceinline__ TouchRange touch_range(const void *first, long bytes)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(first);
    TouchRange r;
    r.lead = (int)(a & 127);
    r.base = reinterpret_cast<const char *>(a - r.lead);
    r.n_lines = (int)((r.lead + bytes + 127) >> 7);
    return r;
}
constexpr int TOUCH_LANES = 32;
// lines first_line + lane of the range (those behind its end: the last one again), a dword of each; lanes 0 ... 31
__device__ __forceinline__ void touch_lines(const TouchRange &r, int first_line, int lane, double *dead_half)
{
    typedef __attribute__((address_space(1))) const void global_cv;
    typedef __attribute__((address_space(3))) void lds_v;
    if (lane < TOUCH_LANES) {
        int line = first_line + lane;
        line = line < r.n_lines ? line : r.n_lines - 1;
        __builtin_amdgcn_global_load_lds((global_cv *)(r.base + 128L * line), (lds_v *)dead_half, 4, 0, 0);
    }
}
constexpr int TOUCH_AHEAD_LINES = 192;          // = 256 bands: six instructions at the start of a wave, under its geometry
constexpr int TOUCH_EVERY_BLOCKS = 2;           // then 32 lines (of which 24 are new) every two band blocks
constexpr int TOUCH_MIN_BANDS = 160;            // below, the XCD-local mapping alone does it: the member's first wave has fetched the
                                                // 10 KB of a 100-band table before the others get to their bands (1000 members x 1000
                                                // lines x 100 / 200 / 640 / 2101 bands, with | without: 233 | 229-235, 383 | 395-403,
                                                // 1180-1244 | 1362-1377, 3490-3520 | 4267-4306 us; profiles/r06/members_stream_ab.log)

// ring rows: 0 = the cache line in front of the wave's first row (its head only), 1 + l = line l of the wave
// MEMBERS (gort_rsurf_members_stream: the same angle lines for every member): a one-dimensional grid of members x waves_per_member
// waves (+ up to 7), XCD x - which takes the workgroups x, x + 8, ... of a launch - works through the contiguous range x of them,
// so the waves of a member share one L2: its band table is fetched from memory once (not once per XCD), the first wave to touch a
// line fetches it for the others, and every XCD writes one window of the output, as in the LUT kernel.
template <bool NT, bool MEMBERS>
__global__ __launch_bounds__(64) void stream_lines_kernel(const gort_canopy *__restrict__ canopy,
                                                          const double *__restrict__ angles, long nA,
                                                          const StreamBand *__restrict__ bands, int nw, int pitch,
                                                          double *__restrict__ out, double *__restrict__ K,
                                                          unsigned waves_per_member, unsigned total_waves)
{
    extern __shared__ __attribute__((aligned(16))) double s_ring[];
    const int lane = threadIdx.x;
    unsigned wave_of_member = blockIdx.x;
    if (MEMBERS) {
        const unsigned per_xcd = (total_waves + 7) >> 3;
        const unsigned j = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (j >= total_waves) return;
        const unsigned member = j / waves_per_member;
        wave_of_member = j - member * waves_per_member;
        canopy += member;
        bands += (long)member * nw;
        out += (long)member * nA * nw;
    }
    GORT_STAMPS_BEGIN();
    GORT_STAMP(0);
    const long a0 = (long)wave_of_member * 64;
    const int lines_here = nA - a0 < 64 ? (int)(nA - a0) : 64;
    const bool live = lane < lines_here;
    const long a = a0 + (live ? lane : lines_here - 1);       // lanes behind the stream compute the last line again, store nothing

    // ---- where the wave's span lies: positions are relative to `origin`, a 256-B aligned address at or below its
    // first element (never dereferenced below `out`)
    const int base_off = (int)((reinterpret_cast<uintptr_t>(out) >> 3) & (RING - 1));
    const long G0 = base_off + a0 * nw;
    const int g0 = (int)(G0 & (RING - 1));
    // ring row 0 is read at the seams only, 16 doubles from (((g0 + 15) & ~15) - 16) & 31: the other half takes the touches
    double *const dead_half = s_ring + (((((g0 + 15) & ~15) - 16) & (RING - 1)) ^ 16);

    // ---- the line: geometry -> record -> line terms, all in registers
    const gort_canopy &c = *canopy;
    LineTerms l;
    {
        const double *ap = angles + 4 * a;
        double in_vza, in_vaa, in_sza, in_saa;
        if (MEMBERS && nw >= TOUCH_MIN_BANDS) {
            // the line's angles, then the first touches - of the canopy record and of the band table's first TOUCH_AHEAD_LINES lines -
            // and a wait for the angles ALONE (the counter retires in order; the compiler would wait for all nine, a trip to HBM at
            // the head of every wave: written out, M0 - the LDS address of global_load_lds - put back as it was)
            static_assert(TOUCH_AHEAD_LINES == 6 * TOUCH_LANES, "six touches of the band table below");
            const TouchRange cr = touch_range(canopy, sizeof(gort_canopy)), tr = touch_range(bands, (long)nw * (long)sizeof(StreamBand));
            auto line_of = [&](const TouchRange &r, int first) {
                int line = first + (lane & (TOUCH_LANES - 1));
                line = line < r.n_lines ? line : r.n_lines - 1;
                return r.base + 128L * line;
            };
            const char *tc = line_of(cr, 0), *t0 = line_of(tr, 0), *t1 = line_of(tr, 32), *t2 = line_of(tr, 64), *t3 = line_of(tr, 96),
                       *t4 = line_of(tr, 128), *t5 = line_of(tr, 160);
            typedef __attribute__((address_space(3))) void lds_v;
            const unsigned scratch = (unsigned)(uintptr_t)(lds_v *)dead_half;
            dbl2 lo, hi;
            unsigned m0_was;
            unsigned long long exec_was;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                         "s_mov_b32 %2, m0\n\ts_mov_b32 m0, %12\n\t"
                         "s_mov_b64 %3, exec\n\ts_mov_b32 exec_hi, 0\n\t"
                         "global_load_lds_dword %5, off\n\tglobal_load_lds_dword %6, off\n\tglobal_load_lds_dword %7, off\n\t"
                         "global_load_lds_dword %8, off\n\tglobal_load_lds_dword %9, off\n\tglobal_load_lds_dword %10, off\n\t"
                         "global_load_lds_dword %11, off\n\t"
                         "s_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_waitcnt vmcnt(7)"
                         : "=&v"(lo), "=&v"(hi), "=&s"(m0_was), "=&s"(exec_was)
                         : "v"(ap), "v"(tc), "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "s"(scratch)
                         : "memory");
            in_vza = lo.x;  in_vaa = lo.y;  in_sza = hi.x;  in_saa = hi.y;
        } else {
            in_vza = ap[0];  in_vaa = ap[1];  in_sza = ap[2];  in_saa = ap[3];
        }
        double vza, sza, saa, raa;
        normalise_angles(in_vza, in_vaa, in_sza, in_saa, vza, sza, saa, raa);
        GeomOut g;
        geometry_core(c, vza, sza, raa, g, K == nullptr);              // reflectances only: gort_geometry.h, row_terms
        double rec[GORT_COEF_STRIDE];
        store_coef(rec, c, g);
        l = line_terms_of_record(rec, c.k_openep, c.k_open);
        if (K && live) {
            double *k = K + 4 * a;
            k[0] = g.Kc;  k[1] = g.Kg;  k[2] = g.Kt;  k[3] = g.Kz;
        }
    }

    GORT_STAMP_ANCHOR(l.alpha);
    GORT_STAMP(1);                                           // the line's geometry and terms
    double *const origin = out - base_off + (G0 - g0);
    const int pos = g0 + lane * nw;                           // my row starts here
    double *const my_ring = s_ring + (lane + 1) * pitch;
    const int sub = lane >> 3, q = lane & 7;                  // as a store lane: row 8 i + sub, 16 bytes number q of its cache line
    const int m = (nw + BLOCK_BANDS - 1) / BLOCK_BANDS;

    // full cache line j of every row (rows whose line j is not complete inside the row skip it): eight reads, then the
    // stores.  What does not depend on j is formed once per wave.
    // Lanes exchange data through LDS here (a lane's ring is read by the eight store lanes of its row; heads go into the
    // neighbour's ring): one wave, so no s_barrier - but the compiler must not move a lane's ring writes behind another
    // lane's reads of them.  Release fence + wave barrier + acquire fence emit no instruction; they only forbid that.
    auto wave_exchange = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // Per store slot: where its 16 bytes of cache line j lie in the ring - the ring holds two lines, line j sits in half
    // (first + j) & 1: rd_even[i] for even j, rd_even[i] + rd_step[i] (+-8 sixteen-byte units) for odd j, one multiply-add per
    // read - and at[i] = the byte offset of its 16 bytes of line 0 from `origin`: 32 bits beside a wave-uniform base that moves
    // 128 bytes per line (the store's scalar-base form: no 64-bit vector address arithmetic).
    int n_full[8], tail_len[8], rd_even[8], rd_step[8];
    unsigned at[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = 8 * i + sub;
        const int pr = g0 + r * nw;
        const int F = (pr + 15) & ~15;
        n_full[i] = r < lines_here ? (pr + nw - F) >> 4 : 0;
        tail_len[i] = r < lines_here ? (pr + nw - F) & 15 : 0;     // what the row leaves in the cache line behind its last full one
        const int first_half = (F >> 4) & 1;
        const int ring_at = (((r + 1) * pitch) >> 1) + q;          // in 16-byte units: pitch is even
        rd_even[i] = ring_at + (first_half << 3);
        rd_step[i] = first_half ? -8 : 8;
        at[i] = (unsigned)(F + 2 * q) * 8u;
    }
    // every row of a full wave has at least this many full lines (pr + nw - F >= nw - 15): lines below it leave unpredicated
    const int n_full_everywhere = lines_here == 64 ? (nw - 15) >> 4 : 0;
    char *const origin_bytes = reinterpret_cast<char *>(origin);
    auto emit_full = [&](int j) {
        wave_exchange();
        dbl2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = reinterpret_cast<const dbl2 *>(s_ring)[rd_even[i] + rd_step[i] * (j & 1)];
        // the reads are waited for HERE, outside the branches of the stores: else the compiler's wait-count pass carries
        // them as pending round the loop and drains everything - the band request included - at the top of the next block
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y));
        char *const line = origin_bytes + 128L * j;               // wave-uniform
        if (j < n_full_everywhere) {
            // the eight stores in their scalar-base form, spelled out: the compiler forms one 64-bit vector address per store
            // (it shares the address arithmetic with the predicated path below).  gfx94x / gfx950 want TWO wait states between a
            // store of more than 8 bytes and a VALU write to its data registers (the compiler emits s_nop 1 for the pattern; its
            // hazard recognizer does not look inside asm, and what it schedules behind the block may write v[i]).
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(at[i]), "v"(v[i]), "s"(line) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(at[i]), "v"(v[i]), "s"(line) : "memory");
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (j < n_full[i]) {
                dbl2 *o = reinterpret_cast<dbl2 *>(line + at[i]);
                if (NT) __builtin_nontemporal_store(v[i], o);
                else *o = v[i];
            }
        }
    };

    // ---- band block 0: its 16 samples stay in registers as well (the row's head is among them).  Its band constants are
    // requested a band ahead like those of every other block (left to the compiler each of the sixteen scalar loads sat directly
    // in front of its wait: ~100 exposed cycles per band of the block)
    double head[BLOCK_BANDS];
    BandRegs A, B;
    band_request(A, bands);
#pragma unroll
    for (int i = 0; i < BLOCK_BANDS; i += 2) {
        band_wait(A);
        band_request(B, bands + i + 1);                        // nw >= 17
        if (i > 0) my_ring[(pos + i - 1) & (RING - 1)] = head[i - 1];
        head[i] = stream_sample(l, band_of(A));
        band_wait(B);
        band_request(A, bands + i + 2);
        my_ring[(pos + i) & (RING - 1)] = head[i];
        head[i + 1] = stream_sample(l, band_of(B));
    }
    // the other blocks: the sample of a band is written to the ring one band later (its ds_write would otherwise sit
    // directly in front of the next wait, which counts LDS operations too)
    {
        int t = BLOCK_BANDS;
        double vprev = head[BLOCK_BANDS - 1];                  // (block 0 left it to this loop's first write)
        // A holds the request for band 16 already
        while (t < nw) {
            const int block_end = t + BLOCK_BANDS < nw ? t + BLOCK_BANDS : nw;
            if (MEMBERS && nw >= TOUCH_MIN_BANDS && (t & (BLOCK_BANDS * TOUCH_EVERY_BLOCKS - 1)) == 0) {        // wave-uniform
                const TouchRange table = touch_range(bands, (long)nw * (long)sizeof(StreamBand));
                const int ahead = ((table.lead + t * (int)sizeof(StreamBand)) >> 7) + TOUCH_AHEAD_LINES;
                if (ahead < table.n_lines) touch_lines(table, ahead, lane, dead_half);
            }
            while (t + 1 < block_end) {
                band_wait(A);
                band_request(B, bands + t + 1);
                my_ring[(pos + t - 1) & (RING - 1)] = vprev;
                vprev = stream_sample(l, band_of(A));
                band_wait(B);
                band_request(A, bands + (t + 2 < nw ? t + 2 : nw - 1));
                my_ring[(pos + t) & (RING - 1)] = vprev;
                vprev = stream_sample(l, band_of(B));
                t += 2;
            }
            if (t < block_end) {                               // an odd band at the end of the last block
                band_wait(A);
                my_ring[(pos + t - 1) & (RING - 1)] = vprev;
                vprev = stream_sample(l, band_of(A));
                t += 1;
            }
            my_ring[(pos + t - 1) & (RING - 1)] = vprev;
            emit_full((t - 1) / BLOCK_BANDS - 1);
        }
        band_wait(A);                                          // the last request may still be on its way
    }
    emit_full(m - 1);
    GORT_STAMP(2);                                           // band blocks and their cache lines

    // ---- the cache lines between rows: my head completes the line my predecessor's tail began
    {
        const int s = (-pos) & 15;
        double *const prev_ring = s_ring + lane * pitch;
        if (live) {
#pragma unroll
            for (int i = 0; i < BLOCK_BANDS - 1; ++i)
                if (i < s) prev_ring[(pos + i) & (RING - 1)] = head[i];
        }
        wave_exchange();                                   // the heads are in their neighbours' rings
        // the seam behind row r = 8 i + sub, from the numbers of its store slot: complete where the next row is the wave's too
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (tail_len[i] > 0) {
                const int r = 8 * i + sub;
                const dbl2 v = reinterpret_cast<const dbl2 *>(s_ring)[rd_even[i] + rd_step[i] * (n_full[i] & 1)];
                double *o = reinterpret_cast<double *>(origin_bytes + 128L * n_full[i] + at[i]);
                if (r + 1 < lines_here) {
                    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(o));
                    else *reinterpret_cast<dbl2 *>(o) = v;
                } else {                                           // the wave's last row: its tail only
                    if (2 * q < tail_len[i]) o[0] = v.x;
                    if (2 * q + 1 < tail_len[i]) o[1] = v.y;
                }
            }
        }
        // the line in front of the wave's first row holds that row's head and nothing else of this wave
        if (sub == 0 && (g0 & 15)) {
            const int hi = (g0 + 15) & ~15;
            emit_cache_line<NT>(s_ring, 0, pitch, hi - 16, g0, hi, q, origin);
        }
    }
    GORT_STAMP(3);                                           // seams
    GORT_STAMPS_END(lines, blockIdx.x, lane == 0);
}

}  // namespace

// Which streams take this kernel: no component spectra, 17 ... LINES_MAX_BANDS bands, and enough lines to fill the
// machine (below that the narrow kernels, whose threads are samples, have more parallelism)
// [r5] ... and beyond 255 bands where the flat-panel kernel's chunks do not fit the rows: a million lines x 257 / 300 / 400 / 500
// bands 446 / 524 / 678 / 837 us here against 539 / 589 / 724 / 862 through records + flat panels; at multiples of 128 bands
// (256: 430 against 447, 384: 625 against 655, 512: 782 against 949) the flat kernel's perfectly aligned chunks win, from ~600
// bands it wins everywhere (2101: 3.04 against 3.43 ms) - profiles/r05/lines_vs_flat_wide.log
constexpr int LINES_MIN_BANDS = 17;
constexpr int LINES_MAX_BANDS = 255;                 // every band count up to here
constexpr int LINES_MAX_BANDS_OFF_GRID = 600;        // and up to here unless the band count is a multiple of 128
bool stream_takes_lines_kernel(int nw, long nA, bool want_scomp)
{
    if (want_scomp || nw < LINES_MIN_BANDS || nA * (long)nw < (1L << 18)) return false;
    if (const char *v = ab_env("GORT_LINES_MAX_BANDS"))       // measuring build, read per call: tests and tools/shape_scan.py move
        return nw <= atoi(v);                                  // the hand-over to the flat-panel kernel inside one process
    return nw <= LINES_MAX_BANDS || (nw <= LINES_MAX_BANDS_OFF_GRID && nw % 128 != 0);
}

// the same lines for several members: the flat-panel kernel has no member dimension, so the line kernel keeps every band count
// from 17 (a thousand members x 1000 lines x 2101 bands: 0.60 of HBM against 0.11 through records + one thread per sample).  A wave
// is 64 lines of ONE member (its band constants come through the scalar unit), so with few lines per member its time is one wave's
// walk over the bands whatever the number of live lanes - 1000 members x 8 ... 64 lines x 100 / 640 bands: 34 / 200 us - where the
// narrow path's grows with the samples: x 100 bands 42 ... 80 us (slower from 8 lines on), x 640 bands 79 / 110 / 151 / 199 / 287 us
// at 8 / 16 / 24 / 32 / 48 lines (faster up to 24).  ADVICE r5; profiles/r06/members_small_lines.log
constexpr long MEMBERS_MIN_LINES_WIDE = 32;          // lines per member from which spectra of >= MEMBERS_WIDE_BANDS take this kernel
constexpr int MEMBERS_WIDE_BANDS = 256;
bool members_stream_takes_lines_kernel(int nw, long lines_per_member, int n_members)
{
    long min_lines = nw >= MEMBERS_WIDE_BANDS ? MEMBERS_MIN_LINES_WIDE : 1;
    if (const char *v = ab_env("GORT_MEMBERS_MIN_LINES")) min_lines = atol(v);      // measuring build: tools/probes/members_stream.py
    return nw >= LINES_MIN_BANDS && lines_per_member >= min_lines && lines_per_member * n_members * (long)nw >= (1L << 18);
}

int launch_stream_lines(const gort_canopy *canopy_dev, int n_members, const double *band_table_dev, int nw, const double *angles_dev,
                        long nA, double *rsurf_dev, double *K_dev, void *stream)
{
    if (nA <= 0 || n_members <= 0) return GORT_OK;
    if (n_members > 1 && K_dev) return fail(GORT_EINVAL, "stream lines kernel: %d members", n_members);
    if (!band_table_dev) return fail(GORT_EINVAL, "stream lines kernel: no band table");
    if (nw < LINES_MIN_BANDS) return fail(GORT_EINVAL, "stream lines kernel: %d bands (needs at least %d)", nw, LINES_MIN_BANDS);
    const long blocks = (nA + 63) / 64;
    if (blocks * n_members + 7 >= (1L << 31) || (long)nw * 64 + RING >= (1L << 30))
        return fail(GORT_EINVAL, "stream lines kernel: %ld lines x %d bands x %d members in one launch", nA, nw, n_members);
    // ring pitch: even (16-B aligned rows); lanes' ring positions differ by nw, so a multiple of four bands wants
    // rows two doubles apart in the banks (two-way conflicts at worst), any other count none
    const int pitch = (nw % 4 == 0) ? RING + 2 : RING;
    const StreamBand *tb = reinterpret_cast<const StreamBand *>(band_table_dev);
    hipStream_t s = (hipStream_t)stream;
    if (n_members > 1) {
        // members: one dimension, ranges of waves per XCD
        const unsigned total = (unsigned)(blocks * n_members);
        const size_t lds = sizeof(double) * 65 * (size_t)pitch;
        hipLaunchKernelGGL((stream_lines_kernel<true, true>), dim3((total + 7) / 8 * 8), dim3(64), lds, s, canopy_dev, angles_dev, nA, tb, nw,
                           pitch, rsurf_dev, K_dev, (unsigned)blocks, total);
        return check_launch("stream_lines_kernel<members>");
    }
    size_t lds = sizeof(double) * 65 * (size_t)pitch;
#ifdef GORT_AB
    if (const char *v = ab_env("GORT_LINES_LDS_BYTES"))        // occupancy probe: a larger allocation = fewer resident waves per CU
        if ((size_t)atol(v) > lds) lds = (size_t)atol(v);
    static const bool nt = !(ab_env("GORT_EXPAND_NT") && atoi(ab_env("GORT_EXPAND_NT")) == 0);
    if (!nt)
        hipLaunchKernelGGL((stream_lines_kernel<false, false>), dim3((unsigned)blocks), dim3(64), lds, s, canopy_dev, angles_dev, nA, tb, nw,
                           pitch, rsurf_dev, K_dev, 0u, 0u);
    else
#endif
        hipLaunchKernelGGL((stream_lines_kernel<true, false>), dim3((unsigned)blocks), dim3(64), lds, s, canopy_dev, angles_dev, nA, tb, nw,
                           pitch, rsurf_dev, K_dev, 0u, 0u);
    return check_launch("stream_lines_kernel");
}

}  // namespace gort
