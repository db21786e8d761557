// gort_energy.hip -- spectral albedo, vegetation and soil absorption per sun direction
// (replaces gortt_energy / gortt_albedo, gortt_albedo.c:7-138).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gort_flat.h"
#include "gort_geometry.h"

#include "gort_stamps.h"

GORT_STAMPS_DEFINE(energy)

namespace gort {
namespace {

// One 512-thread workgroup per angle line = the 32 x 16 Gauss-Legendre nodes of the
// viewing hemisphere (gortt_albedo.c:89-134), one node per thread.  By linearity of
// rsurf in the five angle coefficients the quadrature is applied to the coefficients
// (wavefront shuffle + LDS reduction), then every band costs 5 FMAs:
//   albedo(band) = sum_k [sum_nodes w_node a_k(node)] b_k(sun zenith, band).
constexpr int ENERGY_THREADS = 512;
constexpr long ENERGY_DEDUP_MIN_LINES = 128;      // below: one workgroup per line, no table (BASELINE config 4 has 91 lines, all distinct)

// The sum of v over the wave's 64 lanes, IN LANE 63 (the other lanes hold partial sums).  Data-parallel moves - shifts inside
// the rows of sixteen lanes (an inclusive scan: 1, 2, 4, 8), then the two row broadcasts of the GCN reduction idiom - instead
// of six __shfl_down steps: those are ds_bpermute round trips through the LDS pipe, ~30 dependent ones per line for the five
// sums of the quadrature (1.9 us of a line's 9, by the stamps), where a DPP move costs an issue slot.  One association for
// every form of the albedo kernels (they all call this): the forms still write the same bits.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_moved(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, false);      // lanes without a source: 0
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_moved<0x111, 0xf>(v);          // row_shr:1
    v += dpp_moved<0x112, 0xf>(v);          // row_shr:2
    v += dpp_moved<0x114, 0xf>(v);          // row_shr:4
    v += dpp_moved<0x118, 0xf>(v);          // row_shr:8   -> lane 15 of every row: the row's sum
    v += dpp_moved<0x142, 0xa>(v);          // row_bcast:15 into rows 1 and 3
    v += dpp_moved<0x143, 0xc>(v);          // row_bcast:31 into rows 2 and 3 -> lane 63: the wave's sum
    return v;
}

// ---- lines that share a sun direction share their row ----
// The hemispherical integral depends on the line's SUN direction only: the view angles are the quadrature nodes
// (gortt_albedo.c:62-138 overwrites g->vza / g->vaa; what it keeps of the line is g->sza and g->saa - the sun azimuth
// too, because 32 Gauss-Legendre nodes in azimuth do not integrate the hot-spot cusp exactly and the result moves with
// saa in the 4th digit).  A stream of a million lines with 91 sun zeniths asked for 11 000 x the necessary geometry
// (512 evaluations per line).  So: key = the bits of the normalised (sza, saa) - the very values the kernel uses -,
// a hash table on the device names ONE owner line per key (the lowest line index), energy_kernel evaluates the owners
// only, and a store-bound copy kernel broadcasts their rows.  Bitwise equal to the per-line evaluation by construction.
struct SunKey { unsigned long long z, a; };

__device__ inline SunKey sun_key(const double *__restrict__ angles, long line)
{
    double vza, sza, saa, raa;
    normalise_angles(angles[4 * line], angles[4 * line + 1], angles[4 * line + 2], angles[4 * line + 3], vza, sza, saa, raa);
    SunKey k;
    k.z = (unsigned long long)__double_as_longlong(sza);
    k.a = (unsigned long long)__double_as_longlong(saa);
    return k;
}

__device__ inline unsigned long long sun_hash(SunKey k)
{
    unsigned long long h = k.z * 0x9E3779B97F4A7C15ull ^ (k.a + 0xD1B54A32D192ED03ull + (k.z << 6) + (k.z >> 2));
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h ? h : 1ull;                                 // 0 marks an empty slot
}

// one thread per line: claim / find the slot of the line's key hash, lowest line index becomes the slot's owner.
// A stream has few sun directions far more often than many, and a million compare-and-swaps on ONE address are 13 ms
// (profiles/r05/suns/README.md, energy_table_cost.log).  So the 256 lines of a workgroup first meet in a table in LDS - the
// first line of every hash among them speaks for the others - and a speaker reads before it writes: a slot that holds its
// hash needs no claim, an owner in front of it no minimum (slots are claimed once, owners only ever decrease: a stale read
// costs an atomic, never a result).  The table it leaves is the one every line for itself would have left.
constexpr int KEY_THREADS = 256, KEY_SLOTS = 2 * KEY_THREADS;
__global__ __launch_bounds__(KEY_THREADS) void energy_key_kernel(const double *__restrict__ angles, long nA,
                                                                  unsigned long long *__restrict__ tab, unsigned *__restrict__ owner,
                                                                  unsigned mask, unsigned *__restrict__ slot_of)
{
    __shared__ unsigned long long s_hash[KEY_SLOTS];
    __shared__ unsigned s_first[KEY_SLOTS], s_idx[KEY_SLOTS];
    for (int t = threadIdx.x; t < KEY_SLOTS; t += KEY_THREADS) {
        s_hash[t] = 0ull;
        s_first[t] = 0xffffffffu;
    }
    __syncthreads();
    const long line = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = line < nA;
    unsigned long long h = 0ull;
    unsigned mine = 0;
    if (live) {
        h = sun_hash(sun_key(angles, line));
        mine = (unsigned)(h >> 40) & (KEY_SLOTS - 1);        // other bits than the device table's
        for (;;) {                                           // at most 256 hashes in 512 slots
            const unsigned long long old = atomicCAS(&s_hash[mine], 0ull, h);
            if (old == 0ull || old == h) break;
            mine = (mine + 1) & (KEY_SLOTS - 1);
        }
        atomicMin(&s_first[mine], (unsigned)line);
    }
    __syncthreads();
    if (live && s_first[mine] == (unsigned)line) {
        unsigned idx = (unsigned)h & mask;
        for (unsigned tries = 0; tries <= mask; ++tries) {   // the table has >= 2 nA slots: ends long before
            unsigned long long old = __hip_atomic_load(&tab[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == 0ull) old = atomicCAS(&tab[idx], 0ull, h);
            if (old == 0ull || old == h) break;
            idx = (idx + 1) & mask;
        }
        if (__hip_atomic_load(&owner[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (unsigned)line) atomicMin(&owner[idx], (unsigned)line);
        s_idx[mine] = idx;
    }
    __syncthreads();
    if (live) slot_of[line] = s_idx[mine];
}

// the line whose row `line` shares: the owner of its slot if that one has the same key (two keys with one hash share a
// slot; the one that is not the owner's stands for itself), else the line itself
__device__ inline long energy_rep(const double *__restrict__ angles, long line, const unsigned *__restrict__ owner,
                                  const unsigned *__restrict__ slot_of)
{
    const long o = owner[slot_of[line]];
    if (o == line) return line;
    const SunKey a = sun_key(angles, line), b = sun_key(angles, o);
    return (a.z == b.z && a.a == b.a) ? o : line;
}

// one thread per line, behind energy_key_kernel: slot_of[line] becomes rep[line] (in place: a line reads its own slot
// only), and every block of 256 lines counts its lines that stand for themselves
constexpr int TABLE_THREADS = 256;
static_assert(KEY_THREADS == TABLE_THREADS, "the key kernel is launched on the blocks of the other table kernels");
__global__ __launch_bounds__(TABLE_THREADS) void energy_rep_kernel(const double *__restrict__ angles, long nA,
                                                                    const unsigned *__restrict__ owner, unsigned *__restrict__ slot_of,
                                                                    unsigned *__restrict__ blocks)
{
    const long line = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool own = false;
    if (line < nA) {
        const long rep = energy_rep(angles, line, owner, slot_of);
        slot_of[line] = (unsigned)rep;
        own = rep == line;
    }
    const int n = __syncthreads_count(own ? 1 : 0);
    if (threadIdx.x == 0) blocks[blockIdx.x] = (unsigned)n;
}

// ONE workgroup: blocks[0 .. n_blocks) becomes its exclusive prefix sum, the total goes to uniq[0] (and to *n_rows_out,
// if the caller wants the number of distinct rows on the device).  The rows of the indexed output are numbered by it:
// in the order in which their sun directions first appear in the stream, whatever the order the lines were hashed in.
constexpr int SCAN_THREADS = 1024;
__global__ __launch_bounds__(SCAN_THREADS) void energy_scan_kernel(unsigned *__restrict__ blocks, long n_blocks,
                                                                    unsigned *__restrict__ uniq, unsigned *__restrict__ n_rows_out)
{
    __shared__ unsigned s_wave[SCAN_THREADS / 64];
    __shared__ unsigned s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (long base = 0; base < n_blocks; base += SCAN_THREADS) {
        const long i = base + tid;
        const unsigned v = i < n_blocks ? blocks[i] : 0u;
        unsigned x = v;                                       // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        unsigned before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (i < n_blocks) blocks[i] = before + x - v;
        __syncthreads();
        if (tid == SCAN_THREADS - 1) s_carry = before + x;
        __syncthreads();
    }
    if (tid == 0) {
        uniq[0] = s_carry;
        if (n_rows_out) *n_rows_out = s_carry;
    }
}

// one thread per line, the blocks of energy_rep_kernel: a line that stands for itself takes the next place of the list
// (uniq[1 + place] = line; place = lines of that kind in front of it: the list is in line order) and notes it in idx[line]
__global__ __launch_bounds__(TABLE_THREADS) void energy_place_kernel(long nA, const unsigned *__restrict__ rep,
                                                                      const unsigned *__restrict__ blocks,
                                                                      unsigned *__restrict__ uniq, unsigned *__restrict__ idx)
{
    __shared__ unsigned s_wave[TABLE_THREADS / 64];
    const long line = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool own = line < nA && rep[line] == (unsigned)line;
    const unsigned long long m = __ballot(own ? 1 : 0);
    if (lane == 0) s_wave[wave] = (unsigned)__popcll(m);
    __syncthreads();
    if (!own) return;
    unsigned place = blocks[blockIdx.x] + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) place += s_wave[w];
    uniq[1 + place] = (unsigned)line;
    idx[line] = place;
}

// idx[line] of the other lines: the place of the line they share their row with
__global__ __launch_bounds__(TABLE_THREADS) void energy_index_kernel(long nA, const unsigned *__restrict__ rep, unsigned *__restrict__ idx)
{
    const long line = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= nA) return;
    const unsigned r = rep[line];
    if (r != (unsigned)line) idx[line] = idx[r];
}

// The evaluation of ONE line by one 512-thread workgroup (thread = quadrature node); energy = this member's [nA][nw][3].
// The 512 nodes are 32 azimuths x 16 zeniths (node = 16 i + j, ensure_nodes() in gort_api.hip), and ~1100 of the ~1500
// instructions of a node's geometry do not depend on its azimuth (row_terms, gort_geometry.h): sixteen lanes evaluate them
// once per zenith node into LDS, then every thread finishes its own node - the split the grid kernel makes for its rows.
// Same two functions on the same numbers as geometry_core(): the same bits (SHARE_ROWS = false keeps the round-3 form,
// every thread the whole geometry, for the test that says so: GORT_ENERGY_SHARE_ROWS=0).
constexpr int ENERGY_ZENITH_NODES = 16;
struct EnergyShared {
    double part[5][ENERGY_THREADS / 64];
    double sun[6];
    RowTerms row[ENERGY_ZENITH_NODES];
    RowScratch scr[ENERGY_ZENITH_NODES];
};

// PREFETCH (the per-line kernel, BASELINE config 4): the band constants of the thread's first band pass are requested in front
// of the geometry and those of pass k + 1 in front of the arithmetic of pass k - by the stamps a band pass was a wait for its
// eleven cold constants (the table is 185 KB and nobody has touched it) followed by forty instructions, 4.5 us for two passes
template <bool SHARE_ROWS, bool PREFETCH = false>
__device__ __forceinline__ void energy_line(const gort_canopy &c, const double *__restrict__ L, int nw,
                                            const double *__restrict__ angles, const double *__restrict__ nodes,
                                            double *__restrict__ energy, long a, long out_row, EnergyShared &sh, int band_begin,
                                            int band_end)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    GORT_STAMPS_BEGIN();
    GORT_STAMP(0);
    // everything this thread reads of the node table, in front of the prefetch (the load counter is in order: a wait for a node
    // behind the band constants would be a wait for the band constants); my_row = the row this thread serves in row_terms_split()
    const double node_vaa = nodes[3 * tid], w = nodes[3 * tid + 2];
    const int my_row = tid < 2 * ENERGY_ZENITH_NODES ? tid >> 1 : ((tid - 64) >> 1) & (ENERGY_ZENITH_NODES - 1);
    const double row_vza = nodes[3 * my_row + 1], node_vza = nodes[3 * tid + 1];
    BandTerms ahead;
    if (PREFETCH) ahead = load_band(L, nw, band_begin + tid < band_end ? band_begin + tid : band_begin);
    double vza, sza, saa, raa;
    normalise_angles(angles[4 * a], angles[4 * a + 1], angles[4 * a + 2], angles[4 * a + 3], vza, sza, saa, raa);
    // node geometry (gortt_albedo.c:91-105): vaa = pi + pi x_i in (0, 2pi); vza = acos(x_j)
    {
#pragma clang fp contract(off)
        raa = saa - node_vaa;
        raa = fabs((raa - 2 * PI * (int)(0.5 + raa * INV_PI * 0.5)));
    }
    GeomOut g;
    if (SHARE_ROWS) {
        // reflectances only leave this kernel (gort_geometry.h, row_terms: the 90-degree sun of BASELINE config 4 walked the
        // reference's route for 8.8 us where the other lines' row terms take 4.5, and a launch ends with its longest line)
        // split over lanes (gort_geometry.h, row_terms_split: the same numbers as row_terms())
        row_terms_split(ENERGY_ZENITH_NODES, sh.row, sh.scr, true, [&](int i, const gort_canopy *&ci, double &vz, double &sz) {
            ci = &c;
            vz = i == my_row ? row_vza : node_vza;           // (asked for my_row, or - the first sixteen threads - for row tid:
            sz = sza;                                        //  node tid's zenith is row tid's)
        });
        GORT_STAMP(1);                                       // row terms
        finish_angle(c, sh.row[tid & (ENERGY_ZENITH_NODES - 1)], raa, g);
    } else {
        geometry_core(c, node_vza, sza, raa, g);
    }
    double rec[GORT_COEF_STRIDE];
    store_coef(rec, c, g);
    double part[5] = {w * rec[A_C], w * rec[A_B], w * rec[A_Z], w * rec[A_G], w * rec[A_T]};
    GORT_STAMP_ANCHOR(part[4]);
    GORT_STAMP(2);                                           // this thread's node
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        part[k] = wave_sum(part[k]);
        if (lane == 63) sh.part[k][wave] = part[k];
    }
    if (tid == 0) {
        sh.sun[0] = g.sun.fd;  sh.sun[1] = g.sun.mu;  sh.sun[2] = g.sun.t0;
        sh.sun[3] = g.sun.tp0; sh.sun[4] = g.sun.eps; sh.sun[5] = g.sun.pn0;
    }
    __syncthreads();
    // every thread adds the eight waves' partial sums itself, in wave order (what five threads did for all between two
    // barriers: the same additions in the same order, one barrier less)
    double abar[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < ENERGY_THREADS / 64; ++q) t += sh.part[k][q];
        abar[k] = t;
    }
    GORT_STAMP(3);                                           // the five sums
    SunScalars s;
    s.fd = sh.sun[0];  s.mu = sh.sun[1];  s.t0 = sh.sun[2];  s.tp0 = sh.sun[3];  s.eps = sh.sun[4];  s.pn0 = sh.sun[5];
    const double aC = abar[0], aB = abar[1], aZ = abar[2], aG = abar[3], aT = abar[4];
    for (int i = band_begin + tid; i < band_end; i += ENERGY_THREADS) {
        BandTerms t;
        if (PREFETCH) {
            t = ahead;
            const int next = i + ENERGY_THREADS;
            ahead = load_band(L, nw, next < band_end ? next : i);
        } else {
            t = load_band(L, nw, i);
        }
        const SunTerms b = sun_terms(t, s, c.k_open, c.k_openep);
        const double albedo = dot5(aC, aB, aZ, aG, aT, b.C0, b.B, b.Z, b.G, b.T);
        const double rs = t.rs;
        // energy balance, Lambertian background (gortt_albedo.c:39-52)
        const double Fu2 = b.G * s.pn0 + b.Z * (1. - s.pn0);
        const double Fd2 = s.pn0 + b.Z * (1. - s.pn0) / rs;
        double *o = energy + (out_row * nw + i) * 3;
        o[0] = albedo;
        o[1] = 1. - albedo - Fd2 + Fu2;
        o[2] = Fd2 - Fu2;
    }
    GORT_STAMP(4);                                           // band passes
    GORT_STAMPS_END(energy, a, tid == 0 && band_begin == 0);
}

// blockIdx.x = angle line, blockIdx.y = ensemble member (its canopy and band tables); the nA angle lines are shared by
// the members; energy[member][nA][nw][3].  blockIdx.z = band range: a table of a few lines (BASELINE config 4: 91) leaves
// most CUs idle, so its lines are evaluated by several workgroups, each with the whole quadrature (redundant, on CUs that
// had nothing to do) and one ENERGY_THREADS-wide pass over its share of the bands instead of nw / ENERGY_THREADS passes.
template <bool SHARE_ROWS>
__global__ __launch_bounds__(ENERGY_THREADS, 4) void energy_kernel(const gort_canopy *__restrict__ canopies,
                                                                 const double *__restrict__ Lall, int nw,
                                                                 const double *__restrict__ angles, long nA,
                                                                 const double *__restrict__ nodes,   // [512][3] vaa, vza, weight
                                                                 double *__restrict__ energy_all)
{
    __shared__ EnergyShared sh;
    const long member = blockIdx.y;
    const int per = (nw + (int)gridDim.z - 1) / (int)gridDim.z, band_begin = (int)blockIdx.z * per;
    energy_line<SHARE_ROWS, true>(canopies[member], Lall + member * L_NSLOT * nw, nw, angles, nodes, energy_all + member * nA * nw * 3,
                                  (long)blockIdx.x, (long)blockIdx.x, sh, band_begin, band_begin + per < nw ? band_begin + per : nw);
}

// the lines that stand for themselves (uniq[0] of them, uniq[1..]): workgroups stride over the list.
// Where a list line's row goes (all list kernels): into the dense output at the line's own index, energy_all[member][nA][nw][3]
// (rows_cap < 0) - or, the INDEXED output, to the line's place in the list, rows[member][rows_cap][nw][3], places beyond
// rows_cap not evaluated.
struct RowsOut {
    long member_stride;                 // doubles between the members' outputs
    long rows_cap;                      // < 0: dense
};
template <bool SHARE_ROWS>
__global__ __launch_bounds__(ENERGY_THREADS, 4) void energy_list_kernel(const gort_canopy *__restrict__ canopies,
                                                                      const double *__restrict__ Lall, int nw,
                                                                      const double *__restrict__ angles, long nA,
                                                                      const double *__restrict__ nodes,
                                                                      double *__restrict__ energy_all,
                                                                      const unsigned *__restrict__ uniq, RowsOut ro)
{
    __shared__ EnergyShared sh;
    const long member = blockIdx.y;
    long n_lines = uniq[0];
    if (ro.rows_cap >= 0 && n_lines > ro.rows_cap) n_lines = ro.rows_cap;
    for (long u = blockIdx.x; u < n_lines; u += gridDim.x) {
        __syncthreads();                                     // the shared arrays of the previous line are done with
        // A loop around the inlined body lets the compiler hoist the body's loop-invariant loads (canopy, nodes) in front
        // of the loop: 256 VGPRs + spills in round 3, which then put the body behind a call (162 VGPRs, one workgroup
        // per CU where the per-line kernel has two).  Memory the compiler must assume changed keeps the loads where
        // they are used.
        asm volatile("" ::: "memory");
        const long a = (long)uniq[1 + u];
        energy_line<SHARE_ROWS>(canopies[member], Lall + member * L_NSLOT * nw, nw, angles, nodes,
                                energy_all + member * ro.member_stride, a, ro.rows_cap >= 0 ? u : a, sh, 0, nw);
    }
}

// The same list in BATCHES of ENERGY_BATCH lines per workgroup pass (round 4).  Stamps of the line-after-line form (BASELINE
// config 4, s_memrealtime): 4.5 us of row terms on sixteen lanes of one wave with seven waves waiting at the barrier, 1.9 us
// of node geometry, 1.9 us of reductions, 4.5 us of band passes that each wait for eleven band constants - a chain of
// latencies per line, two lines in flight per CU.  Here one wave evaluates the row terms of the batch's four lines, sixteen
// lanes each (four chains for the issue slots of one), every thread then finishes its node for each of the lines
// (independent work back to back), and the band passes load a band's constants ONCE for the four lines.  Same functions
// on the same numbers, the partial sums added in the same order: the same numbers as energy_line(), bit for bit
// (tests/test_energy_forms.py compares the two).  A million lines, every one its own sun: 36.5 -> 27 ms at 2101 bands,
// 22.7 -> 16 ms at 7 - the latter is the fp64 issue time of the geometry (8300 wave instructions per line: 15 ms).
constexpr int ENERGY_BATCH = 4;                   // x ENERGY_ZENITH_NODES = the lanes of one wave
struct EnergySharedBatch {
    double part[ENERGY_BATCH][5][ENERGY_THREADS / 64];
    double abar[ENERGY_BATCH][5];
    double sun[ENERGY_BATCH][6];
    RowTerms row[ENERGY_BATCH][ENERGY_ZENITH_NODES];
    RowScratch scr[ENERGY_BATCH][ENERGY_ZENITH_NODES];
};

__global__ __launch_bounds__(ENERGY_THREADS, 4) void energy_list_batched_kernel(const gort_canopy *__restrict__ canopies,
                                                                              const double *__restrict__ Lall, int nw,
                                                                              const double *__restrict__ angles, long nA,
                                                                              const double *__restrict__ nodes,
                                                                              double *__restrict__ energy_all,
                                                                              const unsigned *__restrict__ uniq, RowsOut ro)
{
    __shared__ EnergySharedBatch sh;
    const long member = blockIdx.y;
    const gort_canopy &c = canopies[member];
    const double *__restrict__ L = Lall + member * L_NSLOT * nw;
    double *__restrict__ energy = energy_all + member * ro.member_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long n_lines = uniq[0];
    if (ro.rows_cap >= 0 && n_lines > ro.rows_cap) n_lines = ro.rows_cap;
    const double node_vaa = nodes[3 * tid], w = nodes[3 * tid + 2];
    // lines per pass: four where that still leaves two workgroups' worth of batches per CU, fewer for short lists (a
    // million lines of 91 sun directions are a list of 91: one line per workgroup, as many workgroups as lines)
    const int per_pass = n_lines >= 4 * 512 ? ENERGY_BATCH : (n_lines >= 2 * 512 ? 2 : 1);
    GORT_STAMPS_BEGIN();
    for (long base = (long)blockIdx.x * per_pass; base < n_lines; base += (long)gridDim.x * per_pass) {
        asm volatile("" ::: "memory");                       // keeps the body's loads where they are used (see energy_list_kernel)
        GORT_STAMP(0);
        const int lines_here = n_lines - base < per_pass ? (int)(n_lines - base) : per_pass;
        // ---- row terms of the batch's lines, sixteen rows per line, split over lanes (gort_geometry.h: two lanes per row on the
        // first waves, then a lane per row and principal-plane azimuth beside a lane per row on the wave behind them) ----
        row_terms_split(lines_here * ENERGY_ZENITH_NODES, &sh.row[0][0], &sh.scr[0][0], true,
                        [&](int i, const gort_canopy *&ci, double &vz, double &sz) {
                            const long a = (long)uniq[1 + base + (i >> 4)];
                            double vza, saa, raa;
                            normalise_angles(angles[4 * a], angles[4 * a + 1], angles[4 * a + 2], angles[4 * a + 3], vza, sz, saa, raa);
                            ci = &c;
                            vz = nodes[3 * (i & (ENERGY_ZENITH_NODES - 1)) + 1];
                        });
        GORT_STAMP(1);                                       // row terms of the batch
        // ---- every thread its node, line after line ----
        for (int b = 0; b < lines_here; ++b) {
            const long a = (long)uniq[1 + base + b];
            double vza, sza, saa, raa;
            normalise_angles(angles[4 * a], angles[4 * a + 1], angles[4 * a + 2], angles[4 * a + 3], vza, sza, saa, raa);
            {
#pragma clang fp contract(off)
                raa = saa - node_vaa;
                raa = fabs((raa - 2 * PI * (int)(0.5 + raa * INV_PI * 0.5)));
            }
            GeomOut g;
            finish_angle(c, sh.row[b][tid & (ENERGY_ZENITH_NODES - 1)], raa, g);
            double rec[GORT_COEF_STRIDE];
            store_coef(rec, c, g);
            double part[5] = {w * rec[A_C], w * rec[A_B], w * rec[A_Z], w * rec[A_G], w * rec[A_T]};
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                part[k] = wave_sum(part[k]);
                if (lane == 63) sh.part[b][k][wave] = part[k];
            }
            if (tid == 0) {
                sh.sun[b][0] = g.sun.fd;  sh.sun[b][1] = g.sun.mu;  sh.sun[b][2] = g.sun.t0;
                sh.sun[b][3] = g.sun.tp0; sh.sun[b][4] = g.sun.eps; sh.sun[b][5] = g.sun.pn0;
            }
        }
        __syncthreads();
        GORT_STAMP(2);                                       // nodes of the batch
        if (tid < 5 * ENERGY_BATCH) {                        // the order of energy_line()'s adding threads
            const int b = tid / 5, k = tid - 5 * b;
            double x = 0.0;
            for (int q = 0; q < ENERGY_THREADS / 64; ++q) x += sh.part[b][k][q];
            sh.abar[b][k] = x;
        }
        __syncthreads();
        GORT_STAMP(3);                                       // sums
        // ---- band passes: a band's constants once for the lines of the batch ----
        for (int i = tid; i < nw; i += ENERGY_THREADS) {
            const BandTerms t = load_band(L, nw, i);
            for (int b = 0; b < lines_here; ++b) {
                const double *abar = sh.abar[b];
                SunScalars s;
                s.fd = sh.sun[b][0];  s.mu = sh.sun[b][1];  s.t0 = sh.sun[b][2];  s.tp0 = sh.sun[b][3];  s.eps = sh.sun[b][4];  s.pn0 = sh.sun[b][5];
                const SunTerms bt = sun_terms(t, s, c.k_open, c.k_openep);
                const double albedo = dot5(abar[0], abar[1], abar[2], abar[3], abar[4], bt.C0, bt.B, bt.Z, bt.G, bt.T);
                const double rs = t.rs;
                // energy balance, Lambertian background (gortt_albedo.c:39-52)
                const double Fu2 = bt.G * s.pn0 + bt.Z * (1. - s.pn0);
                const double Fd2 = s.pn0 + bt.Z * (1. - s.pn0) / rs;
                double *o = energy + ((ro.rows_cap >= 0 ? base + b : (long)uniq[1 + base + b]) * nw + i) * 3;
                o[0] = albedo;
                o[1] = 1. - albedo - Fd2 + Fu2;
                o[2] = Fd2 - Fu2;
            }
        }
        GORT_STAMP(4);                                       // band passes
        GORT_STAMPS_END(energy, base / per_pass, tid == 0 && blockIdx.y == 0);
        __syncthreads();                                     // rows, sums and sun scalars are done with
    }
}

// rows of the lines that share another line's sun direction: energy[line] = energy[rep[line]].  The output is walked as
// ONE flat array in 1-KiB chunks aligned in absolute address (rows of 3 nw doubles start on 8-byte boundaries only, and
// HBM wants whole lines per wave store: DESIGN.md 5.1 step 2), in PANELS of K steps x W waves like the flat expansion
// kernels - short-lived waves, every XCD a contiguous run of panels where dispatch is round-robin (DESIGN.md 5.1, 5.5 (7):
// a grid-stride loop of long-lived waves wrote 4.85 TB/s here) - and the few source rows stay in L2.  A chunk of 128
// doubles lies in at most two rows when a row has >= 128 doubles (ROWS2: their representatives are two wave-uniform
// loads, the chunk's row and offset advance incrementally); shorter rows (a handful of bands) take the per-element form.
// Which of the two broadcasts works is decided on the device, by the number of owner lines the table found (the host never
// waits for it): both kernels are launched, one returns at once.  1M lines x 2101 bands: the chunk form costs 3 ms when
// nothing is to be copied (it walks every chunk of the output) and 9.3 ms when everything is (0.71 of the roofline); the
// row form 0.1 ms and 10.3 ms - by rows while fewer than ~72 % of the lines are copies.
__device__ __forceinline__ bool broadcast_goes_by_rows(const unsigned *__restrict__ uniq, long nA)
{
    return uniq && 100L * (long)uniq[0] >= 28L * nA;
}

template <bool ROWS2>
__global__ __launch_bounds__(256) void energy_broadcast_kernel(long nA, int row, double *__restrict__ energy_all,
                                                                const unsigned *__restrict__ rep, int shift, long chunks,
                                                                int K, unsigned W, long dq, int dr, int xcd_static,
                                                                XcdDuty duty, long useful_blocks,
                                                                const unsigned *__restrict__ uniq_if_rows_may_take_it)
{
    if (broadcast_goes_by_rows(uniq_if_rows_may_take_it, nA)) return;
    // member y's slab starts y nA row doubles further on: its own offset against the 1-KiB chunk grid (else an odd nA row
    // would leave every other member's 16-byte stores on 8-byte boundaries and its chunks off the grid they are cut for)
    const long n_total = nA * (long)row;
    double *__restrict__ energy = energy_all + (long)blockIdx.y * n_total;
    if (blockIdx.y) {
        shift = (int)((shift + (long)blockIdx.y * (n_total % CHUNK)) % CHUNK);
        chunks = (n_total + shift + CHUNK - 1) / CHUNK;
    }
    const int lane = threadIdx.x & 63;
    const long block = xcd_static ? duty_logical_block((long)blockIdx.x, duty, useful_blocks)
                                  : ((long)blockIdx.x < useful_blocks ? (long)blockIdx.x : -1);
    if (block < 0) return;
    const long wave = block * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long panel = wave / W;
    long ch = panel * K * W + (wave - panel * W);
    // the row of the chunk's first double and its offset in that row (row -1 for the doubles in front of the output)
    long c0 = ch * CHUNK - shift, line_c = -1;
    int off_c = row + (int)c0;
    if (ROWS2 && c0 >= 0) {
        line_c = c0 / row;
        off_c = (int)(c0 - line_c * row);
    }
    int k = 0;
    constexpr int PIPE_STEPS = 16, PIPE_AHEAD = 4, PIPE_MAX_ROW = 1536;
    if (ROWS2 && K == PIPE_STEPS && row <= PIPE_MAX_ROW && ch + (PIPE_STEPS - 1L) * W < chunks && c0 >= 0 &&
        c0 + (PIPE_STEPS - 1L) * W * CHUNK + CHUNK <= n_total) {
        // All sixteen chunks of the wave lie inside the output: ONE straight line of code in which the loads run four steps
        // ahead of the stores - L0 L1 L2 L3, L4 S0, L5 S1, ... - so that a store never waits for more than its own load
        // (the counter is in order: the loads it waits for are older than the stores still in flight).  Step by step, with
        // a branch around every load and store, the compiler waits for everything at the top of each step and the kernel
        // is a chain of load and store latencies.  Every element is copied here, an owner line's onto itself - the same bits.
        // A million lines, all copies (profiles/r04/energy_stream.log): 128 bands 0.80 against 0.88 ms, 400 bands 1.85 against
        // 2.18 - but 2101 bands 9.6 against 9.3 (where the same stores without any load take 7.8 and with loads that never
        // leave one KiB per row 9.1: it is the loads' presence, not their latency or their bytes, that the long rows pay for):
        // rows of up to PIPE_MAX_ROW doubles take this path.
        dbl2 x[PIPE_STEPS];
#pragma unroll
        for (int i = 0; i < PIPE_STEPS + PIPE_AHEAD; ++i) {
            if (i < PIPE_STEPS) {
                const long line0 = line_c;
                const int off0 = off_c;
                off_c += dr;
                line_c += dq;
                if (off_c >= row) { off_c -= row; ++line_c; }
                const long rep0 = (long)rep[line0], rep1 = (long)rep[line0 + 1 < nA ? line0 + 1 : line0];
                const int t = off0 + EPL * lane;
                const bool second0 = t >= row, second1 = t + 1 >= row;
                x[i].x = energy[(second0 ? rep1 : rep0) * row + (second0 ? t - row : t)];
                x[i].y = energy[(second1 ? rep1 : rep0) * row + (second1 ? t + 1 - row : t + 1)];
            }
            if (i >= PIPE_AHEAD)
                __builtin_nontemporal_store(x[i - PIPE_AHEAD], reinterpret_cast<dbl2 *>(energy + c0 + (long)(i - PIPE_AHEAD) * W * CHUNK + EPL * lane));
        }
        return;
    }
    for (; k < K && ch < chunks; ++k, ch += W, c0 += (long)W * CHUNK) {
        const long e0 = c0 + EPL * lane;                              // first of this lane's two elements
        double v[EPL];
        bool put[EPL];
        if (ROWS2) {
            const long line0 = line_c;
            const int off0 = off_c;
            off_c += dr;                                              // the next step's chunk
            line_c += dq;
            if (off_c >= row) { off_c -= row; ++line_c; }
            const long rep0 = line0 >= 0 ? (long)rep[line0] : line0, rep1 = line0 + 1 < nA ? (long)rep[line0 + 1] : line0 + 1;
            if (rep0 == line0 && (off0 + CHUNK <= row || rep1 == line0 + 1)) continue;      // nothing to copy in this chunk
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const long e = e0 + j;
                const int t = off0 + EPL * lane + j;
                const bool second = t >= row;
                const long line = second ? line0 + 1 : line0, r = second ? rep1 : rep0;
                put[j] = e >= 0 && e < n_total && r != line;
                v[j] = put[j] ? energy[r * row + (second ? t - row : t)] : 0.0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const long e = e0 + j;
                put[j] = false;
                v[j] = 0.0;
                if (e >= 0 && e < n_total) {
                    const long line = e / row;
                    const long r = rep[line];
                    if (r != line) {
                        v[j] = energy[r * row + (e - line * row)];
                        put[j] = true;
                    }
                }
            }
        }
        if (put[0] && put[1]) {
            dbl2 x;
            x.x = v[0];
            x.y = v[1];
            __builtin_nontemporal_store(x, reinterpret_cast<dbl2 *>(energy + e0));
        } else {
            if (put[0]) energy[e0] = v[0];
            if (put[1]) energy[e0 + 1] = v[1];
        }
    }
}

// The same broadcast ROW BY ROW (round 4), for lists with few copies: the chunk form above walks every chunk of the output
// (3 ms for a million rows of 2101 bands with nothing to copy).  Here a workgroup takes a whole destination row at a time: every thread requests ALL its 16-B pieces of the source row (a dozen loads in flight, from
// L2: the few source rows are read over and over), waits once, and then has a dozen stores in flight.  Stores are laid on
// the absolute 128-B grid of the destination (the row's head up to the first boundary and a last odd double go singly);
// the source is read at whatever 8-B alignment it has.  Workgroups take blocks of 256 lines off a counter and copy those
// of them that stand for another line.
typedef double dbl2_align8 __attribute__((ext_vector_type(2), aligned(8)));
// ROWCOPY_BATCH pieces per thread and pass: 13 x 256 x 16 B = 53 KB hold a row of 2101 bands in one pass; shorter rows take
// the smallest of 1 / 2 / 4 / 8 / 13 that does
template <int ROWCOPY_BATCH>
__global__ __launch_bounds__(256) void energy_broadcast_rows_kernel(long nA, int row, double *__restrict__ energy_all,
                                                                     const unsigned *__restrict__ rep, unsigned *__restrict__ counters,
                                                                     const unsigned *__restrict__ uniq, int always)
{
    if (!always && !broadcast_goes_by_rows(uniq, nA)) return;
    __shared__ long s_base;
    __shared__ int s_n;
    __shared__ unsigned s_line[256];
    const int tid = threadIdx.x;
    double *__restrict__ energy = energy_all + (long)blockIdx.y * nA * row;
    unsigned *counter = counters + blockIdx.y;
    for (;;) {
        if (tid == 0) {
            s_base = (long)atomicAdd(counter, 256u);
            s_n = 0;
        }
        __syncthreads();
        const long base = s_base;
        if (base >= nA) break;
        const long l = base + tid;
        if (l < nA && rep[l] != (unsigned)l) s_line[atomicAdd(&s_n, 1)] = (unsigned)l;
        __syncthreads();
        const int n = s_n;
        for (int j = 0; j < n; ++j) {
            const long line = s_line[j];
            double *d = energy + line * row;
            const double *src = energy + (long)rep[line] * row;
            int h = (int)((16 - ((reinterpret_cast<uintptr_t>(d) >> 3) & 15)) & 15);     // doubles to the next 128-B boundary
            if (h > row) h = row;
            if (tid < h) d[tid] = src[tid];
            const int pieces = (row - h) >> 1;
            dbl2 *d2 = reinterpret_cast<dbl2 *>(d + h);
            const double *s1 = src + h;
            // straight-line: a piece beyond the row's last is the last one again (the same 16 B written once more) - a branch
            // around a load or a store would make the compiler wait for everything in flight at every join
            for (int p0 = 0; p0 < pieces; p0 += 256 * ROWCOPY_BATCH) {
                dbl2 v[ROWCOPY_BATCH];
                int at[ROWCOPY_BATCH];
#pragma unroll
                for (int b = 0; b < ROWCOPY_BATCH; ++b) {
                    const int p = p0 + b * 256 + tid;
                    at[b] = p < pieces ? p : pieces - 1;
                    v[b] = *reinterpret_cast<const dbl2_align8 *>(s1 + 2 * at[b]);      // 16 B at 8-B alignment: one load
                }
#pragma unroll
                for (int b = 0; b < ROWCOPY_BATCH; ++b) __builtin_nontemporal_store(v[b], d2 + at[b]);
            }
            if (((row - h) & 1) && tid == 0) d[row - 1] = src[row - 1];
        }
        __syncthreads();                                      // the list is done with
    }
}

}  // namespace

// ---- the sun-direction table of nA lines (any nA >= 1) in ws_dev, energy_table_workspace(nA) bytes:
// hash table [cap] u64 + owner [cap] u32 + slot_of / rep [nA] u32 + the list of self-standing lines [1 + nA] u32 + idx [nA] u32 +
// per-block counts [blocks + 1] u32, cap = 2^k >= 2 nA
namespace {
struct EnergyTable {
    unsigned long long *tab;
    unsigned *owner, *rep, *uniq, *idx, *blocks;
    size_t cap;
    long n_blocks;
};
size_t table_cap(long nA)
{
    size_t cap = 1024;
    while (cap < 2 * (size_t)nA) cap <<= 1;
    return cap;
}
EnergyTable carve_table(void *ws_dev, long nA)
{
    EnergyTable t;
    t.cap = table_cap(nA);
    t.n_blocks = (nA + TABLE_THREADS - 1) / TABLE_THREADS;
    t.tab = static_cast<unsigned long long *>(ws_dev);
    t.owner = reinterpret_cast<unsigned *>(t.tab + t.cap);
    t.rep = t.owner + t.cap;
    t.uniq = t.rep + nA;
    t.idx = t.uniq + 1 + nA;
    t.blocks = t.idx + nA;
    return t;
}
}  // namespace

size_t energy_table_workspace(long nA)
{
    if (nA < 1) return 0;
    const size_t cap = table_cap(nA), blocks = (size_t)((nA + TABLE_THREADS - 1) / TABLE_THREADS);
    return cap * (sizeof(unsigned long long) + sizeof(unsigned)) + (3 * (size_t)nA + 1 + blocks + 1) * sizeof(unsigned);
}

// the dense entry points build the table from ENERGY_DEDUP_MIN_LINES lines on
size_t energy_dedup_workspace(long nA) { return nA < ENERGY_DEDUP_MIN_LINES ? 0 : energy_table_workspace(nA); }

const unsigned *energy_table_index(const void *ws_dev, long nA) { return carve_table(const_cast<void *>(ws_dev), nA).idx; }
const unsigned *energy_table_count(const void *ws_dev, long nA) { return carve_table(const_cast<void *>(ws_dev), nA).uniq; }

// rep[line] = the first line of the stream with `line`'s normalised sun direction; the list of those first lines in line
// order (uniq[0] = their number, also written to *n_rows_out_dev if given); idx[line] = the place of rep[line] in the list
int launch_energy_table(const double *angles_dev, long nA, void *ws_dev, unsigned *n_rows_out_dev, void *stream)
{
    if (nA <= 0) return GORT_OK;
    if (nA >= (1L << 31) - 1) return fail(GORT_EINVAL, "energy: %ld lines in one call", nA);
    if (!ws_dev) return fail(GORT_EINVAL, "energy: no workspace for the sun-direction table");
    hipStream_t s = (hipStream_t)stream;
    const EnergyTable t = carve_table(ws_dev, nA);
    if (hipMemsetAsync(t.tab, 0, t.cap * sizeof(unsigned long long), s) != hipSuccess ||
        hipMemsetAsync(t.owner, 0xff, t.cap * sizeof(unsigned), s) != hipSuccess)
        return fail(GORT_ENODEVICE, "energy: cannot clear the sun-direction table");
    const dim3 grid((unsigned)t.n_blocks), block(TABLE_THREADS);
    hipLaunchKernelGGL(energy_key_kernel, grid, block, 0, s, angles_dev, nA, t.tab, t.owner, (unsigned)(t.cap - 1), t.rep);
    int rc = check_launch("energy_key_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(energy_rep_kernel, grid, block, 0, s, angles_dev, nA, (const unsigned *)t.owner, t.rep, t.blocks);
    if ((rc = check_launch("energy_rep_kernel"))) return rc;
    hipLaunchKernelGGL(energy_scan_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, t.blocks, t.n_blocks, t.uniq, n_rows_out_dev);
    if ((rc = check_launch("energy_scan_kernel"))) return rc;
    hipLaunchKernelGGL(energy_place_kernel, grid, block, 0, s, nA, (const unsigned *)t.rep, (const unsigned *)t.blocks, t.uniq, t.idx);
    if ((rc = check_launch("energy_place_kernel"))) return rc;
    hipLaunchKernelGGL(energy_index_kernel, grid, block, 0, s, nA, (const unsigned *)t.rep, t.idx);
    return check_launch("energy_index_kernel");
}

// the list kernel behind a table: rows_cap < 0 dense (out_dev[member][nA][nw][3], the owners' rows only), else the indexed
// form (out_dev[member][rows_cap][nw][3]); n_rows_known: the length of the list if the host has it (else -1: up to nA)
static int launch_energy_list(const gort_canopy *canopies_dev, int n_members, const double *L_dev, int nw, const double *angles_dev,
                              long nA, const double *nodes_dev, double *out_dev, long rows_cap, const EnergyTable &t,
                              long n_rows_known, hipStream_t s)
{
    const char *sr = ab_env("GORT_ENERGY_SHARE_ROWS");       // measuring build, read per call: the test switches it inside one process
    const bool share_rows = !(sr && atoi(sr) == 0);
    long most = n_rows_known >= 0 ? n_rows_known : nA;
    if (rows_cap >= 0 && most > rows_cap) most = rows_cap;
    if (most <= 0) return GORT_OK;
    const unsigned wgs = (unsigned)(most < 8192 ? most : 8192);
    RowsOut ro;
    ro.rows_cap = rows_cap;
    ro.member_stride = (rows_cap >= 0 ? rows_cap : nA) * (long)nw * 3;
#ifdef GORT_AB
    const char *pl = ab_env("GORT_ENERGY_BATCH");            // 0: one line after the other (tests compare the two)
    if (share_rows && pl && atoi(pl) == 0)
        hipLaunchKernelGGL(energy_list_kernel<true>, dim3(wgs, (unsigned)n_members), dim3(ENERGY_THREADS), 0, s, canopies_dev, L_dev, nw,
                           angles_dev, nA, nodes_dev, out_dev, (const unsigned *)t.uniq, ro);
    else if (!share_rows)
        hipLaunchKernelGGL(energy_list_kernel<false>, dim3(wgs, (unsigned)n_members), dim3(ENERGY_THREADS), 0, s, canopies_dev, L_dev, nw,
                           angles_dev, nA, nodes_dev, out_dev, (const unsigned *)t.uniq, ro);
    else
#endif
        hipLaunchKernelGGL(energy_list_batched_kernel, dim3(wgs, (unsigned)n_members), dim3(ENERGY_THREADS), 0, s, canopies_dev, L_dev,
                           nw, angles_dev, nA, nodes_dev, out_dev, (const unsigned *)t.uniq, ro);
    (void)share_rows;
    return check_launch("energy_list_kernel");
}

// INDEXED output: the distinct rows only, rows_dev[member][rows_cap][nw][3] in the order of the table's list (places beyond
// rows_cap are not evaluated); ws_dev holds the table of these very lines (launch_energy_table, any stream ordered before `stream`)
int launch_energy_rows(const gort_canopy *canopies_dev, int n_members, const double *L_dev, int nw, const double *angles_dev,
                       long nA, const double *nodes_dev, double *rows_dev, long rows_cap, const void *ws_dev, long n_rows_known,
                       void *stream)
{
    if (nA <= 0 || nw <= 0 || n_members <= 0 || rows_cap <= 0) return GORT_OK;
    if (n_members > 65535) return fail(GORT_EINVAL, "energy: %d members in one launch (max 65535)", n_members);
    if (!ws_dev) return fail(GORT_EINVAL, "energy: no sun-direction table");
    return launch_energy_list(canopies_dev, n_members, L_dev, nw, angles_dev, nA, nodes_dev, rows_dev, rows_cap,
                              carve_table(const_cast<void *>(ws_dev), nA), n_rows_known, (hipStream_t)stream);
}

// ws_dev: energy_dedup_workspace(nA) bytes, or nullptr = every line evaluated (few lines; tests compare the two)
int launch_energy(const gort_canopy *canopies_dev, int n_members, const double *L_dev, int nw,
                  const double *angles_dev, long nA, const double *nodes_dev, double *energy_dev, void *ws_dev,
                  bool xcd_round_robin, void *stream)
{
    if (nA <= 0 || nw <= 0 || n_members <= 0) return GORT_OK;
    if (n_members > 65535) return fail(GORT_EINVAL, "energy: %d members in one launch (max 65535)", n_members);
    hipStream_t s = (hipStream_t)stream;
    if (!ws_dev || nA < ENERGY_DEDUP_MIN_LINES) {
        const char *sr = ab_env("GORT_ENERGY_SHARE_ROWS");   // measuring build, read per call: the test switches it inside one process
        const bool share_rows = !(sr && atoi(sr) == 0);
        if (nA >= (1L << 31)) return fail(GORT_EINVAL, "energy: %ld lines in one launch", nA);
        // band ranges: as many as give every range one pass, as far as every workgroup still has a CU of its own (two
        // per CU measured slower than none: C4 27.4 us unsplit, 25.8 in two ranges = 182 workgroups, 29.3 in three)
        static int cus = 0;
        if (cus == 0) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
                cus = 256;
            (void)hipGetLastError();
        }
        long splits = (nw + ENERGY_THREADS - 1) / ENERGY_THREADS;
        if (splits * nA * n_members > (long)cus) splits = (long)cus / (nA * n_members);
        if (splits < 1) splits = 1;
        if (splits > 64) splits = 64;
        const dim3 grid((unsigned)nA, (unsigned)n_members, (unsigned)splits);
#ifdef GORT_AB
        if (!share_rows)
            hipLaunchKernelGGL(energy_kernel<false>, grid, dim3(ENERGY_THREADS), 0, s,
                               canopies_dev, L_dev, nw, angles_dev, nA, nodes_dev, energy_dev);
        else
#endif
            hipLaunchKernelGGL(energy_kernel<true>, grid, dim3(ENERGY_THREADS), 0, s,
                               canopies_dev, L_dev, nw, angles_dev, nA, nodes_dev, energy_dev);
        (void)share_rows;
        return check_launch("energy_kernel");
    }
    int rc = launch_energy_table(angles_dev, nA, ws_dev, nullptr, stream);
    if (rc) return rc;
    const EnergyTable t = carve_table(ws_dev, nA);
    if ((rc = launch_energy_list(canopies_dev, n_members, L_dev, nw, angles_dev, nA, nodes_dev, energy_dev, -1, t, -1, s))) return rc;
    const unsigned *rep = t.rep;
    const unsigned *uniq = t.uniq;
    unsigned long long *tab = t.tab;
    const size_t cap = t.cap;
    const int row = 3 * nw;
    // rows of at least a KiB, and counters that fit the table's memory (done with by now): the row-by-row form is launched
    // too, and the number of owner lines decides on the device which of the two works (broadcast_goes_by_rows)
    const char *bf = ab_env("GORT_ENERGY_BROADCAST");        // measuring build: "chunks" / "rows" = one form for everything
    const bool rows_possible = row >= CHUNK && (size_t)n_members * sizeof(unsigned) <= cap * sizeof(unsigned long long) &&
                               !(bf && bf[0] == 'c');
    const bool rows_always = rows_possible && bf && bf[0] == 'r';
    if (rows_possible) {
        unsigned *counters = reinterpret_cast<unsigned *>(tab);
        if (hipMemsetAsync(counters, 0, (size_t)n_members * sizeof(unsigned), s) != hipSuccess)
            return fail(GORT_ENODEVICE, "energy: cannot clear the broadcast counters");
        static int cus = 0;
        if (cus == 0) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
                cus = 256;
            (void)hipGetLastError();
        }
        long wg = (nA + 255) / 256;                           // blocks of lines there are
        if (wg > 7L * cus) wg = 7L * cus;                    // 66 VGPRs: seven of these four-wave workgroups per CU
        const int passes_of_one = (row / 2 + 255) / 256;      // 16-B pieces of a row over the 256 threads
#define GORT_ROWS_LAUNCH(B) hipLaunchKernelGGL(energy_broadcast_rows_kernel<B>, dim3((unsigned)wg, (unsigned)n_members), dim3(256), 0, s, nA, row, energy_dev, rep, counters, (const unsigned *)uniq, rows_always ? 1 : 0)
        if (passes_of_one <= 1) GORT_ROWS_LAUNCH(1);
        else if (passes_of_one <= 2) GORT_ROWS_LAUNCH(2);
        else if (passes_of_one <= 4) GORT_ROWS_LAUNCH(4);
        else if (passes_of_one <= 8) GORT_ROWS_LAUNCH(8);
        else GORT_ROWS_LAUNCH(13);
#undef GORT_ROWS_LAUNCH
        if ((rc = check_launch("energy_broadcast_rows_kernel"))) return rc;
        if (rows_always) return GORT_OK;
    }
    const unsigned *uniq_for_choice = rows_possible ? uniq : nullptr;
    const int shift = (int)((reinterpret_cast<uintptr_t>(energy_dev) / sizeof(double)) % CHUNK);
    const long chunks = (nA * (long)row + shift + CHUNK - 1) / CHUNK;         // member 0; the others derive theirs (one more at most)
    // panels of 16 steps x 2048 waves (32 MB), XCD-contiguous where the dispatch is round-robin.  1M lines x 2101 bands, 91
    // sun directions (profiles/r03/energy_broadcast_sweep.log): the whole call 11.15 ms with the grid-stride loop of round 3's
    // first version, 10.8 / 9.7 / 9.5 / 9.4 / 9.26 / 9.4 ms for 4 / 6 / 8 / 12 / 16 / 32 steps, the same for 1024 ... 8192 waves
    const int K = 16;
    const unsigned W = 2048;
    const long panels = (chunks + (n_members > 1 ? 1 : 0) + (long)K * W - 1) / ((long)K * W);
    const long useful = (panels * W + 3) / 4;
    XcdDuty duty;
    const long nblocks = plan_xcd_duty(xcd_round_robin ? 1 : 0, useful, nullptr, duty);
    if (nblocks >= (1L << 31)) return fail(GORT_EINVAL, "energy: %ld workgroups in one launch", nblocks);
    const dim3 grid((unsigned)nblocks, (unsigned)n_members);
    const long dq = (long)W * CHUNK / row;
    const int dr = (int)((long)W * CHUNK - dq * row);
    if (row >= CHUNK)
        hipLaunchKernelGGL(energy_broadcast_kernel<true>, grid, dim3(256), 0, s, nA, row, energy_dev, rep, shift, chunks, K, W, dq, dr,
                           xcd_round_robin ? 1 : 0, duty, useful, uniq_for_choice);
    else
        hipLaunchKernelGGL(energy_broadcast_kernel<false>, grid, dim3(256), 0, s, nA, row, energy_dev, rep, shift, chunks, K, W, dq, dr,
                           xcd_round_robin ? 1 : 0, duty, useful, uniq_for_choice);
    return check_launch("energy_broadcast_kernel");
}

}  // namespace gort
