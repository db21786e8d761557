// front_probe.hip -- the store pattern of PERSISTENT waves that own a column for life and walk down the output in step:
// does a horizontal write front (all resident waves in the same few row blocks) beat the diagonal one of the flat
// kernels (waves of a panel start one after the other, so at any instant the resident waves stand on K different rows)?
// Output of `lines` x 2101 doubles.  A row block = A lines (A doubles of alignment: 16 -> every 1-KiB wave store starts on
// a 128-B line; 128 -> on a 1-KiB boundary, the flat kernels' alignment); a wave = 128 consecutive elements of a row
// block, the same bands in every row block.  XCD x (blockIdx & 7) owns an eighth of the row blocks; inside it T teams
// of W = ceil(A * 2101 / 128) waves, team j takes the row blocks j, j + T, ... of the eighth (INTERLEAVE 1) or the
// j-th of T contiguous parts (0).  Optional per step: one dependent scalar load (the line record) and `work` dependent
// fp64 FMA pairs (the sample's issue slots).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/front_probe.hip -o /tmp/front_probe;  /tmp/front_probe [lines] [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double dbl2 __attribute__((ext_vector_type(2)));

template <int WORK, bool REC, int THROTTLE = -1>
__global__ __launch_bounds__(256) void front(double *out, long row_blocks, long block_elems, int W, int wgs_per_team, int T,
                                             int interleave, const double *__restrict__ rec, int A, int chipwide)
{
    const int x = blockIdx.x & 7;
    const int i = blockIdx.x >> 3;                            // workgroup of this XCD
    const int team = i / wgs_per_team;
    if (team >= T) return;
    int col = __builtin_amdgcn_readfirstlane((i - team * wgs_per_team) * 4 + (int)(threadIdx.x >> 6));
    // chipwide: a team spans the eight XCDs, XCD x holds columns [x, x + 1) * 4 wgs_per_team of EVERY row block
    if (chipwide) col += x * 4 * wgs_per_team;
    if (col >= W) return;
    const int lane = threadIdx.x & 63;
    const long per_xcd = chipwide ? row_blocks : (row_blocks + 7) / 8;
    long r0 = chipwide ? 0 : x * per_xcd, r1 = r0 + per_xcd < row_blocks ? r0 + per_xcd : row_blocks;
    long r, dr;
    if (interleave) { r = r0 + team; dr = T; }
    else { const long part = (per_xcd + T - 1) / T; r = r0 + team * part; r1 = r + part < r1 ? r + part : r1; dr = 1; }
    const long e = (long)col * 128 + 2 * lane;
    const bool live = e < block_elems;
    double *o = out + e + r * block_elems;
    const long step = dr * block_elems;
    dbl2 v; v.x = 1.0 + lane; v.y = 2.0;
    for (; r < r1; r += dr) {
        dbl2 y = v;
        if (REC) y.x += rec[r * A * 16];                     // wave-uniform: a scalar load the store depends on
#pragma unroll
        for (int q = 0; q < WORK; ++q) { y.x = __builtin_fma(y.x, 1.0000001, 0.5); y.y = __builtin_fma(y.y, 0.9999999, 0.25); }
        if (live) __builtin_nontemporal_store(y, reinterpret_cast<dbl2 *>(o));
        if (THROTTLE >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(THROTTLE) : "memory");     // at most THROTTLE stores of a wave in flight
        o += step;
    }
}

// the same teams (A lines per row block, XCD-owned, T contiguous parts), but column c starts S_of(c) row blocks into its
// part and wraps around: at any instant a team's writes are spread over `stagger` row blocks (a diagonal front, as the
// steady state of short-lived waves makes one) instead of standing on one
template <int WORK, bool REC>
__global__ __launch_bounds__(256) void front_staggered(double *out, long row_blocks, long block_elems, int W, int wgs_per_team, int T,
                                                       const double *__restrict__ rec, int A, int stagger, int spread)
{
    const int x = blockIdx.x & 7;
    const int i = blockIdx.x >> 3;
    const int team = i / wgs_per_team;
    if (team >= T) return;
    const int col = __builtin_amdgcn_readfirstlane((i - team * wgs_per_team) * 4 + (int)(threadIdx.x >> 6));
    if (col >= W) return;
    const int lane = threadIdx.x & 63;
    const long per_xcd = (row_blocks + 7) / 8;
    const long r0 = x * per_xcd, r1 = r0 + per_xcd < row_blocks ? r0 + per_xcd : row_blocks;
    const long part = (per_xcd + T - 1) / T;
    const long p0 = r0 + team * part, p1 = p0 + part < r1 ? p0 + part : r1;
    const long len = p1 - p0;
    if (len <= 0) return;
    // spread 0: neighbouring columns one row block apart (col % stagger); 1: runs of W / stagger columns share a row block
    long k0 = spread ? (long)col * stagger / W : col % stagger;
    k0 %= len;
    const long e = (long)col * 128 + 2 * lane;
    const bool live = e < block_elems;
    dbl2 v; v.x = 1.0 + lane; v.y = 2.0;
    long r = p0 + k0;
    for (long k = 0; k < len; ++k) {
        dbl2 y = v;
        if (REC) y.x += rec[r * A * 16];
#pragma unroll
        for (int q = 0; q < WORK; ++q) { y.x = __builtin_fma(y.x, 1.0000001, 0.5); y.y = __builtin_fma(y.y, 0.9999999, 0.25); }
        if (live) __builtin_nontemporal_store(y, reinterpret_cast<dbl2 *>(out + e + r * block_elems));
        if (++r == p1) r = p0;
    }
}

// the flat kernels' pattern without arithmetic (as panel_shape_probe.hip): panels of K steps x W waves of 1-KiB chunks,
// XCD x owns a contiguous run of workgroups; short-lived waves for small K.  K is a template parameter so that the
// kernels have different NAMES in a counter trace.
template <int K>
__global__ __launch_bounds__(256) void panels_bare(double *out, long chunks, unsigned W, long per_xcd_blocks, long useful)
{
    const long b = blockIdx.x;
    const long block = (b & 7) * per_xcd_blocks + (b >> 3);
    if ((b >> 3) >= per_xcd_blocks || block >= useful) return;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(block * 4 + (threadIdx.x >> 6)));
    const unsigned panel = wave / W, w = wave - panel * W;
    const long c0 = (long)panel * K * W + w;
    const int lane = threadIdx.x & 63;
    dbl2 v; v.x = 1.0 + lane; v.y = 2.0;
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        const long c = c0 + (long)k * W;
        if (c < chunks) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(out + c * 128 + 2 * lane));
    }
}

// persistent waves that work through the SAME tasks in the SAME order as the short-lived waves of panels_bare<K> (task =
// K steps of one column of a panel, adjacent waves take adjacent columns, XCD x owns a contiguous run of tasks): is it the
// lifetime of a wave that slows its stores, or what it does during that life?
template <int K>
__global__ __launch_bounds__(256) void tasks_bare(double *out, long chunks, unsigned W, long tasks_per_xcd, long n_tasks, int waves_per_xcd)
{
    const int x = blockIdx.x & 7;
    const long lw = (long)(blockIdx.x >> 3) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    dbl2 v; v.x = 1.0 + lane; v.y = 2.0;
    const long t1 = (x + 1) * tasks_per_xcd < n_tasks ? (x + 1) * tasks_per_xcd : n_tasks;
    for (long t = x * tasks_per_xcd + lw; t < t1; t += waves_per_xcd) {
        const long panel = t / W, w = t - panel * W;
        const long c0 = panel * K * W + w;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long c = c0 + (long)k * W;
            if (c < chunks) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(out + c * 128 + 2 * lane));
        }
    }
}

// ... and the same persistent waves taking their tasks DYNAMICALLY (one returning atomic per task on their XCD's counter,
// requested one task ahead): a wave on a slower CU then simply takes fewer tasks, as the dispatcher arranges for short-lived
// workgroups
template <int K>
__global__ __launch_bounds__(256) void tasks_dynamic(double *out, long chunks, unsigned W, long tasks_per_xcd, long n_tasks, int *counters)
{
    // one atomic per WORKGROUP and group of four adjacent tasks (a returning atomic on one address completes every ~16 ns:
    // one per wave and task capped 6-step tasks at 3 TB/s), requested one group ahead, handed to the waves through LDS
    __shared__ int s_group[2];
    const int x = blockIdx.x & 7;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    dbl2 v; v.x = 1.0 + lane; v.y = 2.0;
    const long t0 = x * tasks_per_xcd, t1 = t0 + tasks_per_xcd < n_tasks ? t0 + tasks_per_xcd : n_tasks;
    int *ctr = counters + 64 * x;
    if (threadIdx.x == 0) s_group[0] = atomicAdd(ctr, 1);
    __syncthreads();
    int buf = 0;
    long t = t0 + 4L * s_group[0] + wave;
    while (t - wave < t1) {
        if (threadIdx.x == 0) s_group[buf ^ 1] = atomicAdd(ctr, 1);          // the next group, requested before this one's stores
        if (t < t1) {
            const long panel = t / W, w = t - panel * W;
            const long c0 = panel * K * W + w;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const long c = c0 + (long)k * W;
                if (c < chunks) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(out + c * 128 + 2 * lane));
            }
        }
        __syncthreads();
        buf ^= 1;
        t = t0 + 4L * s_group[buf] + wave;
    }
}

template <int K>
static float run_dynamic(double *out, long n, unsigned W, int wgs_per_cu, int reps, hipEvent_t e0, hipEvent_t e1, int *counters)
{
    const long chunks = (n + 127) / 128;
    const long panels = (chunks + (long)K * W - 1) / ((long)K * W);
    const long n_tasks = panels * W, tasks_per_xcd = (n_tasks + 7) / 8;
    const int wgs = 256 * wgs_per_cu;
    float total = 0.f;
    for (int i = 0; i < reps + 2; ++i) {
        CK(hipMemsetAsync(counters, 0, 8 * 64 * sizeof(int), 0));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(tasks_dynamic<K>, dim3(wgs), dim3(256), 0, 0, out, chunks, W, tasks_per_xcd, n_tasks, counters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 2) total += ms;
    }
    return total / reps;
}

template <int K>
static float run_tasks(double *out, long n, unsigned W, int wgs_per_cu, int reps, hipEvent_t e0, hipEvent_t e1)
{
    const long chunks = (n + 127) / 128;
    const long panels = (chunks + (long)K * W - 1) / ((long)K * W);
    const long n_tasks = panels * W, tasks_per_xcd = (n_tasks + 7) / 8;
    const int wgs = 256 * wgs_per_cu, waves_per_xcd = wgs / 8 * 4;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(tasks_bare<K>, dim3(wgs), dim3(256), 0, 0, out, chunks, W, tasks_per_xcd, n_tasks, waves_per_xcd);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(tasks_bare<K>, dim3(wgs), dim3(256), 0, 0, out, chunks, W, tasks_per_xcd, n_tasks, waves_per_xcd);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int K>
static float run_panels(double *out, long n, unsigned W, int reps, hipEvent_t e0, hipEvent_t e1)
{
    const long chunks = (n + 127) / 128;
    const long panels = (chunks + (long)K * W - 1) / ((long)K * W);
    const long useful = (panels * W + 3) / 4, per_xcd = (useful + 7) / 8;
    const dim3 grid((unsigned)(8 * per_xcd));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(panels_bare<K>, grid, dim3(256), 0, 0, out, chunks, W, per_xcd, useful);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(panels_bare<K>, grid, dim3(256), 0, 0, out, chunks, W, per_xcd, useful);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const long nlines = argc > 1 ? atol(argv[1]) : 1048576;
    const int reps = argc > 2 ? atoi(argv[2]) : 8;
    const int nw = 2101;
    const long n = nlines * nw;
    double *out, *rec;
    CK(hipMalloc(&out, (n + 4096) * 8));
    CK(hipMalloc(&rec, (nlines + 4096) * 16 * 8));
    CK(hipMemset(rec, 0, (nlines + 4096) * 16 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto f) {
        f(); f(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps;
    };
    if (argc > 3 && argv[3][0] == 't') {
        // tasks: short-lived waves against persistent waves on the same tasks
        printf("panels of K steps x 2101 waves, short-lived waves:  K=6 %6.0f  K=8 %6.0f  K=12 %6.0f  K=16 %6.0f GB/s\n",
               n * 8 / run_panels<6>(out, n, 2101, reps, e0, e1) / 1e6, n * 8 / run_panels<8>(out, n, 2101, reps, e0, e1) / 1e6,
               n * 8 / run_panels<12>(out, n, 2101, reps, e0, e1) / 1e6, n * 8 / run_panels<16>(out, n, 2101, reps, e0, e1) / 1e6);
        for (int wgs_per_cu : {2, 4, 7})
            printf("the same tasks, persistent waves (%d workgroups per CU):    K=6 %6.0f  K=8 %6.0f  K=12 %6.0f  K=16 %6.0f GB/s\n", wgs_per_cu,
                   n * 8 / run_tasks<6>(out, n, 2101, wgs_per_cu, reps, e0, e1) / 1e6, n * 8 / run_tasks<8>(out, n, 2101, wgs_per_cu, reps, e0, e1) / 1e6,
                   n * 8 / run_tasks<12>(out, n, 2101, wgs_per_cu, reps, e0, e1) / 1e6, n * 8 / run_tasks<16>(out, n, 2101, wgs_per_cu, reps, e0, e1) / 1e6);
        int *counters;
        CK(hipMalloc(&counters, 8 * 64 * sizeof(int)));
        for (int wgs_per_cu : {2, 4, 7})
            printf("the same tasks, persistent waves, tasks taken dynamically (%d workgroups per CU): K=6 %6.0f  K=8 %6.0f  K=12 %6.0f  K=16 %6.0f GB/s\n", wgs_per_cu,
                   n * 8 / run_dynamic<6>(out, n, 2101, wgs_per_cu, reps, e0, e1, counters) / 1e6, n * 8 / run_dynamic<8>(out, n, 2101, wgs_per_cu, reps, e0, e1, counters) / 1e6,
                   n * 8 / run_dynamic<12>(out, n, 2101, wgs_per_cu, reps, e0, e1, counters) / 1e6, n * 8 / run_dynamic<16>(out, n, 2101, wgs_per_cu, reps, e0, e1, counters) / 1e6);
        return 0;
    }
    if (argc > 3 && argv[3][0] == 'p') {
        // pmc: three kernels with different names for a counter trace - short-lived waves (K = 6), long ones (K = 64),
        // persistent ones (A = 16, three XCD-owned teams)
        const float k6 = run_panels<6>(out, n, 2101, reps, e0, e1);
        const float k64 = run_panels<64>(out, n, 2101, reps, e0, e1);
        const int A = 16, T = 3;
        const long block_elems = (long)A * nw, row_blocks = nlines / A;
        const int W = (int)((block_elems + 127) / 128), wgs_per_team = (W + 3) / 4;
        const dim3 grid(8 * wgs_per_team * T);
        const float pers = timeit([&] { hipLaunchKernelGGL((front<0, false>), grid, dim3(256), 0, 0, out, row_blocks, block_elems, W, wgs_per_team, T, 0, rec, A, 0); });
        printf("panels K=6 W=2101: %7.1f us %5.0f GB/s | K=64: %7.1f us %5.0f GB/s | persistent: %7.1f us %5.0f GB/s\n", k6 * 1e3, n * 8 / k6 / 1e6,
               k64 * 1e3, n * 8 / k64 / 1e6, pers * 1e3, n * 8 / pers / 1e6);
        return 0;
    }
    if (argc > 3) {
        // scan: the same pattern (A = 16, three XCD-owned teams) at offsets of 2 GiB through an allocation 48 GiB larger
        // than the output - the placement comb of gort_lut_alloc for THIS pattern
        CK(hipFree(out));
        const long slack = 48L << 30;
        CK(hipMalloc(&out, (n + 4096) * 8 + slack));
        const int A = 16, T = 3;
        const long block_elems = (long)A * nw, row_blocks = nlines / A;
        const int W = (int)((block_elems + 127) / 128), wgs_per_team = (W + 3) / 4;
        const dim3 grid(8 * wgs_per_team * T);
        for (long off = 0; off <= slack; off += 2L << 30) {
            double *o = out + off / 8;
            for (int inter = 1; inter >= 0; --inter) {
                const float bare = timeit([&] { hipLaunchKernelGGL((front<0, false>), grid, dim3(256), 0, 0, o, row_blocks, block_elems, W, wgs_per_team, T, inter, rec, A, 0); });
                const float work = timeit([&] { hipLaunchKernelGGL((front<28, true>), grid, dim3(256), 0, 0, o, row_blocks, block_elems, W, wgs_per_team, T, inter, rec, A, 0); });
                printf("offset %2ld GiB, %s: bare %7.1f us %5.0f GB/s | + record + 56 FMA %7.1f us %5.0f GB/s\n", off >> 30,
                       inter ? "interleaved" : "contiguous ", bare * 1e3, n * 8 / bare / 1e6, work * 1e3, n * 8 / work / 1e6);
                fflush(stdout);
            }
        }
        return 0;
    }
    {
        // does bounding the stores a wave has in flight help?  (A = 16, three XCD-owned teams, contiguous parts)
        const int A = 16, T = 3;
        const long block_elems = (long)A * nw, row_blocks = nlines / A;
        const int W = (int)((block_elems + 127) / 128), wgs_per_team = (W + 3) / 4;
        const dim3 grid(8 * wgs_per_team * T);
#define THR(N) { const float b = timeit([&] { hipLaunchKernelGGL((front<0, false, N>), grid, dim3(256), 0, 0, out, row_blocks, block_elems, W, wgs_per_team, T, 0, rec, A, 0); }); \
                 const float w = timeit([&] { hipLaunchKernelGGL((front<28, true, N>), grid, dim3(256), 0, 0, out, row_blocks, block_elems, W, wgs_per_team, T, 0, rec, A, 0); }); \
                 printf("at most %2d stores of a wave in flight: bare %7.1f us %5.0f GB/s | + record + 56 FMA %7.1f us %5.0f GB/s\n", N, b * 1e3, n * 8 / b / 1e6, w * 1e3, n * 8 / w / 1e6); fflush(stdout); }
        THR(0) THR(1) THR(2) THR(4) THR(8) THR(16) THR(32)
        for (int spread = 0; spread < 2; ++spread)
            for (int stagger : {1, 4, 16, 64, 256, 1024}) {
                const float b = timeit([&] { hipLaunchKernelGGL((front_staggered<0, false>), grid, dim3(256), 0, 0, out, row_blocks, block_elems, W, wgs_per_team, T, rec, A, stagger, spread); });
                const float w = timeit([&] { hipLaunchKernelGGL((front_staggered<28, true>), grid, dim3(256), 0, 0, out, row_blocks, block_elems, W, wgs_per_team, T, rec, A, stagger, spread); });
                printf("team spread over %4d row blocks (%s): bare %7.1f us %5.0f GB/s | + record + 56 FMA %7.1f us %5.0f GB/s\n", stagger,
                       spread ? "runs of columns" : "column % stagger", b * 1e3, n * 8 / b / 1e6, w * 1e3, n * 8 / w / 1e6);
                fflush(stdout);
            }
    }
    for (int A : {16, 32, 128}) {
        const long block_elems = (long)A * nw;
        const long row_blocks = nlines / A;
        const int W = (int)((block_elems + 127) / 128);
        for (int chipwide = 0; chipwide < 2; ++chipwide)
        for (int T : {1, 2, 3, 4, 6, 8, 12, 24}) {
            const int wgs_per_team = chipwide ? (W + 31) / 32 : (W + 3) / 4;       // per XCD
            if ((long)wgs_per_team * T > 7 * 32) continue;       // 7 workgroups per CU, 32 CUs per XCD: all resident at once
            for (int inter = 1; inter >= 0; --inter) {
                if (T == 1 && !inter) continue;
                const dim3 grid(8 * wgs_per_team * T);
                const float bare = timeit([&] { hipLaunchKernelGGL((front<0, false>), grid, dim3(256), 0, 0, out, row_blocks, block_elems, W, wgs_per_team, T, inter, rec, A, chipwide); });
                const float recs = timeit([&] { hipLaunchKernelGGL((front<0, true>), grid, dim3(256), 0, 0, out, row_blocks, block_elems, W, wgs_per_team, T, inter, rec, A, chipwide); });
                const float work = timeit([&] { hipLaunchKernelGGL((front<28, true>), grid, dim3(256), 0, 0, out, row_blocks, block_elems, W, wgs_per_team, T, inter, rec, A, chipwide); });
                printf("A=%3d lines/row block, %s teams of W=%4d waves, T=%d (%s): bare %7.1f us %5.0f GB/s | + record %7.1f us %5.0f GB/s | + record + 56 FMA %7.1f us %5.0f GB/s\n",
                       A, chipwide ? "chip-wide" : "XCD-owned", W, T, inter ? "interleaved" : "contiguous ", bare * 1e3, n * 8 / bare / 1e6, recs * 1e3, n * 8 / recs / 1e6,
                       work * 1e3, n * 8 / work / 1e6);
                fflush(stdout);
            }
        }
    }
    CK(hipGetLastError());
    return 0;
}
