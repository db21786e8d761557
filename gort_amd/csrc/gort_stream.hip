// gort_stream.hip -- the arbitrary-angle stream (the reference's real interface, gortt.c:232-329) at the
// standard of the LUT kernel: lines that share a sun zenith share their five (sun zenith, band) terms.
//
// Real streams carry few distinct sun zeniths (a scene, a day of overpasses, a principal plane), but in any
// order, and the output rsurf[line][band] has rows of nw doubles (nw = 2101: odd) that start on 8-byte
// boundaries only.  So:
//
//   stream_group_kernel   one workgroup per TILE of consecutive lines: a dictionary of the tile's distinct sun
//                         zeniths in LDS (keys = the normalised zenith's bits), the tile's lines counted and
//                         listed per zenith, the lists cut into items of <= M lines, and every zenith entered
//                         in a global dictionary whose slot is the row of the sun table.  No global atomics on
//                         the lines themselves: a skewed stream (one sun zenith) costs the same as a flat one.
//   stream_sun_kernel     sun[slot][2][nw] = p_df, t'_df for the occupied slots (sun_pair, the stream family's form).
//   expand_stream_grouped_kernel
//                         wave = (item, 112-position segment of the row).  p_df, t'_df and six band constants of the
//                         segment's bands stay in registers for all lines of the item; per line 10 FMAs per sample
//                         from the line's nine terms (scalar cache, one line ahead).
//                         A row starts s = (line * nw) mod 16 doubles past a 128-B boundary, different for every
//                         line, so the values are rotated by s through a wave-private LDS strip (the wave also
//                         evaluates the 16 bands in front of its segment) and leave as whole 128-B lines:
//                         one aligned 16-B non-temporal store per lane.  Tiles are dealt to the XCDs in
//                         contiguous runs, so every XCD writes one compact window (TILE x nw x 8 B) at a time.
//
// A tile with more than GRP_UMAX distinct sun zeniths (or more than GRP_GMAX in the whole call) raises the
// `direct` flag (flags[0]: -1 = clear, as one memset of 0xff leaves it together with the empty dictionary): the grouped kernel then returns at once and expand_flat_stream_kernel (per-line sun terms,
// gort_brdf.hip) does the whole stream instead.  The decision is taken on the device; the host never waits.
#include <cstdlib>
#include <cstring>

#include "gort_device.h"

namespace gort {
namespace {

constexpr unsigned long long KEY_EMPTY = ~0ull;
constexpr int GRP_UMAX = 128;        // distinct sun zeniths per tile
constexpr int GRP_HT = 512;          // LDS dictionary entries per tile (4 x GRP_UMAX)
constexpr int GRP_GSLOTS = 1024;     // global dictionary = rows of the sun table
constexpr int GRP_GMAX = 512;        // distinct sun zeniths per call
constexpr int GRP_THREADS = 1024;
constexpr int GRP_LPT = 4;           // lines per thread of stream_group_kernel: tiles of up to 4096 lines
constexpr int SUN_NTERMS = 2;        // (sun zenith, band) numbers of the stream family's sample: p_df, t'_df
constexpr int SEG = 128;             // bands a wave evaluates per line
constexpr int SEGP = 112;            // positions a wave stores per line: seven 128-B lines
constexpr int HALO = 16;             // doubles a row may start past a 128-B boundary (exclusive)

struct StreamItem {
    int slot;        // row of the sun table
    int first;       // index into order[]
    int count;       // lines, 1..lines_per_item
    int pad;
};

__device__ __forceinline__ unsigned hash_key(unsigned long long k)
{
    k ^= k >> 29;
    k *= 0x9E3779B97F4A7C15ull;
    return (unsigned)(k >> 40);
}

__device__ __forceinline__ int wave_excl_sum(int v, int lane, int &total)
{
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    total = __shfl(x, 63, 64);
    return x - v;
}

// One workgroup per tile.  order[], items[] and n_items[] describe the tile's groups; flags[0] is raised when
// the tile (or the call) has too many distinct sun zeniths for the grouped form.
__global__ __launch_bounds__(GRP_THREADS) void stream_group_kernel(const double *__restrict__ angles, long nA,
                                                                    int tile_lines, int max_items, int lines_per_item,
                                                                    unsigned long long *__restrict__ gkeys,
                                                                    int *__restrict__ flags, int *__restrict__ n_items,
                                                                    StreamItem *__restrict__ items,
                                                                    int *__restrict__ order)
{
    __shared__ unsigned long long s_key[GRP_HT];
    __shared__ unsigned long long s_ukey[GRP_UMAX];
    __shared__ int s_uid[GRP_HT];
    __shared__ int s_cnt[GRP_UMAX], s_off[GRP_UMAX + 1], s_ioff[GRP_UMAX + 1], s_gslot[GRP_UMAX];
    __shared__ int s_wsum[GRP_HT / 64];
    __shared__ int s_nins, s_over, s_nuniq;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long tile = blockIdx.x;
    const long base = tile * tile_lines;
    const int n = (int)(nA - base < tile_lines ? nA - base : tile_lines);
    if (tid < GRP_HT) s_key[tid] = KEY_EMPTY;
    if (tid < GRP_UMAX) s_cnt[tid] = 0;
    if (tid == 0) { s_nins = 0; s_over = 0; }
    __syncthreads();

    // 1) the tile's dictionary: every line finds (or claims) the slot of its sun zenith
    int slot[GRP_LPT];
#pragma unroll
    for (int k = 0; k < GRP_LPT; ++k) {
        slot[k] = -1;
        const int l = tid + k * GRP_THREADS;
        if (l >= n) continue;
        unsigned long long key = (unsigned long long)__double_as_longlong(normalised_sza(angles[4 * (base + l) + 2]));
        if (key == KEY_EMPTY) key ^= 1ull;                  // a NaN payload that happens to be all ones
        unsigned h = hash_key(key) & (GRP_HT - 1);
        bool found = false;
        for (int probe = 0; probe < GRP_HT && !found; ++probe) {
            if (*(volatile int *)&s_over) break;            // the tile is lost already: stop filling (and scanning) the table
            unsigned long long cur = *(volatile unsigned long long *)&s_key[h];
            if (cur == KEY_EMPTY) {
                cur = atomicCAS(&s_key[h], KEY_EMPTY, key);
                if (cur == KEY_EMPTY) {
                    if (atomicAdd(&s_nins, 1) + 1 > GRP_UMAX) s_over = 1;
                    cur = key;
                }
            }
            if (cur == key) found = true;
            else h = (h + 1) & (GRP_HT - 1);
        }
        if (found) slot[k] = (int)h;
        else s_over = 1;
    }
    __syncthreads();
    if (s_over) {
        if (tid == 0) { atomicAnd(&flags[0], 0); n_items[tile] = 0; }
        return;
    }

    // 2) number the occupied slots (in slot order: deterministic) and remember their keys
    const bool occ = tid < GRP_HT && s_key[tid < GRP_HT ? tid : 0] != KEY_EMPTY;
    const unsigned long long occ_mask = __ballot(occ);
    if (tid < GRP_HT && lane == 0) s_wsum[wave] = __popcll(occ_mask);
    __syncthreads();
    if (tid < GRP_HT) {
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_wsum[w];
        const int uid = before + __popcll(occ_mask & ((1ull << lane) - 1ull));
        s_uid[tid] = uid;
        if (occ) s_ukey[uid] = s_key[tid];
        if (tid == GRP_HT - 1) s_nuniq = uid + (occ ? 1 : 0);
    }
    __syncthreads();
    const int nuniq = s_nuniq;

    // 3) lines per sun zenith, and each line's place in its list
    int pos[GRP_LPT], uid_of[GRP_LPT];
#pragma unroll
    for (int k = 0; k < GRP_LPT; ++k) {
        pos[k] = 0;
        uid_of[k] = 0;
        if (slot[k] >= 0) {
            uid_of[k] = s_uid[slot[k]];
            pos[k] = atomicAdd(&s_cnt[uid_of[k]], 1);
        }
    }
    __syncthreads();

    // 4) wave 0: offsets of the lists and of their items; waves 2..: the sun table row of every zenith
    if (wave == 0) {
        const int u0 = 2 * lane, u1 = 2 * lane + 1;
        const int c0 = u0 < nuniq ? s_cnt[u0] : 0, c1 = u1 < nuniq ? s_cnt[u1] : 0;
        const int i0 = (c0 + lines_per_item - 1) / lines_per_item, i1 = (c1 + lines_per_item - 1) / lines_per_item;
        int tot, itot;
        const int ex = wave_excl_sum(c0 + c1, lane, tot);
        const int iex = wave_excl_sum(i0 + i1, lane, itot);
        s_off[u0] = ex;            s_ioff[u0] = iex;
        s_off[u1] = ex + c0;       s_ioff[u1] = iex + i0;
        if (lane == 63) { s_off[GRP_UMAX] = tot; s_ioff[GRP_UMAX] = itot; }
    } else if (tid >= 128 && tid < 128 + GRP_UMAX) {
        const int u = tid - 128;
        if (u < nuniq) {
            const unsigned long long key = s_ukey[u];
            unsigned h = hash_key(key) & (GRP_GSLOTS - 1);
            int g = -1;
            for (int probe = 0; probe < GRP_GSLOTS && g < 0; ++probe) {
                unsigned long long cur = __hip_atomic_load(&gkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == KEY_EMPTY) {
                    cur = atomicCAS(&gkeys[h], KEY_EMPTY, key);
                    if (cur == KEY_EMPTY) {
                        if (atomicAdd(&flags[1], 1) + 2 > GRP_GMAX) atomicAnd(&flags[0], 0);       // counts from -1
                        cur = key;
                    }
                }
                if (cur == key) g = (int)h;
                else h = (h + 1) & (GRP_GSLOTS - 1);
            }
            if (g < 0) { atomicAnd(&flags[0], 0); g = 0; }
            s_gslot[u] = g;
        }
    }
    __syncthreads();

    // 5) the lists, and the items cut from them
#pragma unroll
    for (int k = 0; k < GRP_LPT; ++k)
        if (slot[k] >= 0) order[base + s_off[uid_of[k]] + pos[k]] = (int)(base + tid + k * GRP_THREADS);
    const int total_items = s_ioff[GRP_UMAX];
    for (int j = tid; j < total_items; j += GRP_THREADS) {
        int lo = 0, hi = nuniq - 1;                          // last u with s_ioff[u] <= j
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_ioff[mid] <= j) lo = mid; else hi = mid - 1;
        }
        const int part = j - s_ioff[lo];
        const int left = s_cnt[lo] - part * lines_per_item;
        StreamItem it;
        it.slot = s_gslot[lo];
        it.first = (int)base + s_off[lo] + part * lines_per_item;
        it.count = left < lines_per_item ? left : lines_per_item;
        it.pad = 0;
        items[tile * max_items + j] = it;
    }
    if (tid == 0) n_items[tile] = total_items;
}

// sun[slot][2][nw] = p_df, t'_df (sun_pair, gort_device.h) for every occupied slot of the global dictionary
// (blockIdx.y, blockIdx.y + gridDim.y, ...); the scalars mu, t0, 1 - t'0, 1 + 2 mu are formed from the zenith exactly
// as line_terms() forms them from a line's record
__global__ __launch_bounds__(256) void stream_sun_kernel(const gort_canopy *__restrict__ canopy,
                                                          const double *__restrict__ L, int nw,
                                                          const unsigned long long *__restrict__ gkeys,
                                                          const int *__restrict__ flags, double *__restrict__ sun)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (flags[0] != -1 || i >= nw) return;
    const gort_canopy &c = *canopy;
    StreamBand t;
    bool have = false;
    for (int slot = blockIdx.y; slot < GRP_GSLOTS; slot += gridDim.y) {
        const unsigned long long key = gkeys[slot];
        if (key == KEY_EMPTY) continue;
        if (!have) { t = stream_band(load_band(L, nw, i)); have = true; }
        const SunScalars s = sun_from_zenith(c, __longlong_as_double((long long)key));
        const LineTerms l = line_terms(0.0, 0.0, 0.0, 0.0, 0.0, s.fd, s.mu, s.t0, s.tp0, s.eps, c.k_openep, c.k_openep + c.k_open);
        double pdf, tpdf;
        sun_pair(t, l.mu, l.t0, l.omtp0, l.m2, pdf, tpdf);
        double *o = sun + (long)slot * SUN_NTERMS * nw;
        o[i] = pdf;
        o[nw + i] = tpdf;
    }
}

// wave = (tile, item, segment).  See the head of this file.
// coef: the stream records (GORT_COEF_STRIDE doubles per line, A_C..A_T in front).
// shift0 = (address of rsurf / 8) mod 16; a line's row starts s = (shift0 + line * nw) mod 16 doubles past a 128-B
// boundary, and position P of the row's aligned image holds band P - s.
// A wave evaluates SEG = 128 bands (two per lane) and stores SEGP = 112 positions (seven 128-B lines, 16 B per lane
// on 56 lanes): positions [112 seg, 112 seg + 112) need the bands [112 seg - 15, 112 seg + 112) whatever s is, and
// the wave's strip holds band 112 seg - 16 + i at index i.
template <bool NT>
__global__ __launch_bounds__(256) void expand_stream_grouped_kernel(const double *__restrict__ sun,
                                                                     const double *__restrict__ L,
                                                                     const double *__restrict__ coef, int nw,
                                                                     const int *__restrict__ flags,
                                                                     const int *__restrict__ n_items,
                                                                     const StreamItem *__restrict__ items,
                                                                     const int *__restrict__ order, int n_tiles,
                                                                     int tiles_per_xcd, int max_items, int nq,
                                                                     FastDiv div_tile, FastDiv div_nq, int shift0,
                                                                     double *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) double s_strip[4][SEG];
    if (flags[0] != -1) return;                             // the per-line kernel does this stream
    // workgroups b, b+8, ... share an XCD (round-robin dispatch): XCD x works through tiles x*tiles_per_xcd ...
    // one after the other, items in list order, the nq segment quads of an item side by side.
    // tiles_per_xcd = 0: too few tiles to go round, plain order.
    const unsigned b = blockIdx.x;
    const unsigned per_tile = (unsigned)(max_items * nq);
    unsigned tile, r;
    if (tiles_per_xcd > 0) {
        const unsigned x = b & 7u, i = b >> 3;
        const unsigned t_in = fast_div(i, div_tile);
        r = i - t_in * per_tile;
        tile = x * (unsigned)tiles_per_xcd + t_in;
    } else {
        tile = fast_div(b, div_tile);
        r = b - tile * per_tile;
    }
    if (tile >= (unsigned)n_tiles) return;
    const unsigned item = fast_div(r, div_nq);
    const int quad = (int)(r - item * (unsigned)nq);
    if ((int)item >= n_items[tile]) return;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int seg = quad * 4 + wv;
    if (seg * SEGP >= nw + HALO - 1) return;                // behind the last position of any row
    const StreamItem it = items[(long)tile * max_items + item];
    const int count = __builtin_amdgcn_readfirstlane(it.count);
    const int slot = __builtin_amdgcn_readfirstlane(it.slot);

    // the item's lines, one per lane (read back with v_readlane: no dependent scalar load per line)
    const int my_line = order[it.first + (lane < count ? lane : 0)];
    // the wave's 128 bands, two per lane: p_df, t'_df of the item's sun zenith and the six band constants of the
    // stream family's sample (gort_device.h)
    const double *__restrict__ sp = sun + (long)slot * SUN_NTERMS * nw;
    const int band0 = seg * SEGP - HALO + 2 * lane;
    StreamBand t[2];
    double pdf[2], tpdf[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int bj = band0 + j;
        const bool ok = bj >= 0 && bj < nw;
        t[j] = stream_band(load_band(L, nw, ok ? bj : 0));
        pdf[j] = ok ? sp[bj] : 0.0;
        tpdf[j] = ok ? sp[(long)nw + bj] : 0.0;
    }
    double *strip = s_strip[wv];
    const int nw16 = nw & 15;
    const int p0 = seg * SEGP + 2 * lane;                    // position in a row's aligned image (lanes < 56 store)
    const bool storer = lane < SEGP / 2;
    // the nine line terms come through the scalar cache (records in layout 1), one line ahead
    int a = __builtin_amdgcn_readlane(my_line, 0);
    const double *__restrict__ rec = coef + (long)a * GORT_COEF_STRIDE;
    double c[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) c[q] = rec[q];
    for (int i = 0; i < count; ++i) {
        const int a_next = __builtin_amdgcn_readlane(my_line, i + 1 < count ? i + 1 : i);
        const double *__restrict__ rn = coef + (long)a_next * GORT_COEF_STRIDE;
        double n[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) n[q] = rn[q];
        dbl2 v;
        v.x = stream_sample(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], t[0], pdf[0], tpdf[0]);
        v.y = stream_sample(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], t[1], pdf[1], tpdf[1]);
        *reinterpret_cast<dbl2 *>(strip + 2 * lane) = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int s = (shift0 + (a & 15) * nw16) & 15;      // wave-uniform
        const int idx = (storer ? 2 * lane : 0) + HALO - s;  // strip index of position p0
        dbl2 w;
        w.x = strip[idx];
        w.y = strip[idx + 1];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double *o = out + ((long)a * nw - s) + p0;
        const bool interior = seg * SEGP >= s && seg * SEGP + SEGP - 1 - s < nw;     // wave-uniform
        if (interior) {
            if (storer) {
                if (NT) __builtin_nontemporal_store(w, reinterpret_cast<dbl2 *>(o));
                else *reinterpret_cast<dbl2 *>(o) = w;
            }
        } else if (storer) {
            const int b0 = p0 - s;
            const bool ok0 = b0 >= 0 && b0 < nw, ok1 = b0 + 1 >= 0 && b0 + 1 < nw;
            if (ok0 && ok1) {
                if (NT) __builtin_nontemporal_store(w, reinterpret_cast<dbl2 *>(o));
                else *reinterpret_cast<dbl2 *>(o) = w;
            } else if (ok0) {
                o[0] = w.x;
            } else if (ok1) {
                o[1] = w.y;
            }
        }
        a = a_next;
#pragma unroll
        for (int q = 0; q < 9; ++q) c[q] = n[q];
    }
}

struct GroupTuning {
    bool enabled = true, nt = true;
    int tile = 2048, lines_per_item = 32;
    GroupTuning()
    {
        if (const char *v = getenv("GORT_STREAM_TILE")) tile = atoi(v);
        if (const char *v = getenv("GORT_STREAM_ITEM")) lines_per_item = atoi(v);
        if (const char *v = getenv("GORT_EXPAND_NT")) nt = atoi(v) != 0;
        if (tile < 256) tile = 256;
        if (tile > GRP_THREADS * GRP_LPT) tile = GRP_THREADS * GRP_LPT;
        if (lines_per_item < 1) lines_per_item = 1;
        if (lines_per_item > 64) lines_per_item = 64;        // one line per lane
    }
};

const GroupTuning &group_tuning()
{
    static const GroupTuning t;
    return t;
}

struct GroupLayout {
    int tile, n_tiles, max_items, lines_per_item;
    size_t off_gkeys, off_nitems, off_items, off_order, bytes;
};

GroupLayout group_layout(long nA)
{
    const GroupTuning &t = group_tuning();
    GroupLayout g;
    g.tile = t.tile;
    g.lines_per_item = t.lines_per_item;
    g.n_tiles = (int)((nA + g.tile - 1) / g.tile);
    g.max_items = (g.tile + g.lines_per_item - 1) / g.lines_per_item + GRP_UMAX;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    g.off_gkeys = 256;                                                     // flags in front
    g.off_nitems = g.off_gkeys + up(sizeof(unsigned long long) * GRP_GSLOTS);
    g.off_items = g.off_nitems + up(sizeof(int) * (size_t)g.n_tiles);
    g.off_order = g.off_items + up(sizeof(StreamItem) * (size_t)g.n_tiles * g.max_items);
    g.bytes = g.off_order + up(sizeof(int) * (size_t)nA);
    return g;
}

FastDiv fast_div_for(unsigned d)
{
    FastDiv f;
    f.sh = 0;
    while ((1ull << f.sh) < d) ++f.sh;
    f.mul = (unsigned)((1ull << (31 + f.sh)) / d + 1);
    return f;
}

int launch_ok(const char *what)
{
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(GORT_ENODEVICE, "%s: %s", what, hipGetErrorString(err));
    return GORT_OK;
}

}  // namespace

bool stream_group_enabled() { return group_tuning().enabled; }

void stream_group_workspace(int nw, long nA, size_t *ws_bytes, size_t *sun_bytes)
{
    *ws_bytes = 0;
    *sun_bytes = 0;
    if (!group_tuning().enabled || nw < SEG || nA <= 0 || nA >= (1L << 31)) return;
    *ws_bytes = group_layout(nA).bytes;
    *sun_bytes = sizeof(double) * SUN_NTERMS * (size_t)nw * GRP_GSLOTS;
}

// Everything of the grouped form on `stream`: dictionary + lists, sun table, expansion.  *direct_flag_dev is the
// device flag the per-line kernel has to honour (it runs only when the flag is raised).
int launch_expand_stream_grouped(const gort_canopy *canopy_dev, const double *L_dev, int nw, const double *angles_dev,
                                 const double *coef_dev, long nA, double *rsurf_dev, void *ws_dev, double *sun_dev,
                                 void *stream, void *coef_ready_event, const int **direct_flag_dev)
{
    hipStream_t s = (hipStream_t)stream;
    const GroupLayout g = group_layout(nA);
    char *ws = static_cast<char *>(ws_dev);
    int *flags = reinterpret_cast<int *>(ws);
    unsigned long long *gkeys = reinterpret_cast<unsigned long long *>(ws + g.off_gkeys);
    int *n_items = reinterpret_cast<int *>(ws + g.off_nitems);
    StreamItem *items = reinterpret_cast<StreamItem *>(ws + g.off_items);
    int *order = reinterpret_cast<int *>(ws + g.off_order);
    *direct_flag_dev = flags;
    // flags (clear = -1) and the empty dictionary in one fill
    hipError_t err = hipMemsetAsync(ws, 0xff, g.off_gkeys + sizeof(unsigned long long) * GRP_GSLOTS, s);
    if (err != hipSuccess) return fail(GORT_ENODEVICE, "stream grouping: %s", hipGetErrorString(err));
    hipLaunchKernelGGL(stream_group_kernel, dim3((unsigned)g.n_tiles), dim3(GRP_THREADS), 0, s, angles_dev, nA, g.tile,
                       g.max_items, g.lines_per_item, gkeys, flags, n_items, items, order);
    int rc = launch_ok("stream_group_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(stream_sun_kernel, dim3((nw + 255) / 256, GRP_GSLOTS / 8), dim3(256), 0, s, canopy_dev, L_dev, nw,
                       gkeys, flags, sun_dev);
    if ((rc = launch_ok("stream_sun_kernel"))) return rc;
    if (coef_ready_event && hipStreamWaitEvent(s, (hipEvent_t)coef_ready_event, 0) != hipSuccess)
        return fail(GORT_ENODEVICE, "stream expansion: cannot wait for the geometry kernel");
    const int nseg = (nw + HALO - 1 + SEGP - 1) / SEGP;
    const int nq = (nseg + 3) / 4;
    const long per_tile = (long)g.max_items * nq;
    const int tiles_per_xcd = g.n_tiles >= 16 ? (g.n_tiles + 7) / 8 : 0;
    const long blocks = tiles_per_xcd ? 8L * tiles_per_xcd * per_tile : (long)g.n_tiles * per_tile;
    if (blocks >= (1L << 31) || per_tile >= (1L << 31))
        return fail(GORT_EINVAL, "stream expansion: %ld workgroups in one launch", blocks);
    const int shift0 = (int)((reinterpret_cast<uintptr_t>(rsurf_dev) / sizeof(double)) % 16);
    const FastDiv dt = fast_div_for((unsigned)per_tile), dq = fast_div_for((unsigned)nq);
    if (group_tuning().nt)
        hipLaunchKernelGGL(expand_stream_grouped_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, sun_dev, L_dev, coef_dev,
                           nw, flags, n_items, items, order, g.n_tiles, tiles_per_xcd, g.max_items, nq, dt, dq, shift0,
                           rsurf_dev);
    else
        hipLaunchKernelGGL(expand_stream_grouped_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, sun_dev, L_dev, coef_dev,
                           nw, flags, n_items, items, order, g.n_tiles, tiles_per_xcd, g.max_items, nq, dt, dq, shift0,
                           rsurf_dev);
    return launch_ok("expand_stream_grouped_kernel");
}

}  // namespace gort
