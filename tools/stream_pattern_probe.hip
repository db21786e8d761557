// stream_pattern_probe.hip -- bare store patterns on a stream-sized output (65 536 x 2101 doubles = 1.1 GB) (tuning aid)
//   hipcc --offload-arch=gfx950 -O3 tools/stream_pattern_probe.hip -o tools/stream_pattern_probe
// A: the LUT kernel's panels (K steps x W waves of 1-KiB chunks), XCD x owns a contiguous run of panels
// B: row segments: wave = (M consecutive lines, one 896-B segment of the row, 128-B aligned), XCD x owns a run of lines
// C: as B with random lines inside a tile of T lines
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double dbl2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void panels(double *out, long chunks, int K, unsigned W, long per_xcd_blocks, long useful)
{
    const long b = blockIdx.x;
    const long block = (b & 7) * per_xcd_blocks + (b >> 3);
    if ((b >> 3) >= per_xcd_blocks || block >= useful) return;
    const unsigned wave = (unsigned)(block * 4 + (threadIdx.x >> 6));
    const unsigned panel = wave / W, w = wave - panel * W;
    const long c0 = (long)panel * K * W + w;
    const int lane = threadIdx.x & 63;
    dbl2 v; v.x = 1.0; v.y = 2.0;
    for (int k = 0; k < K; ++k) {
        const long c = c0 + (long)k * W;
        if (c < chunks) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(out + c * 128 + 2 * lane));
    }
}

// item = M lines listed in order[]; wave = (item, seg); seg bytes = 8 * segp
__global__ __launch_bounds__(256) void rowsegs(double *out, const int *__restrict__ order, int n_items, int M, int nw,
                                               int nq, int segp, long per_xcd, int nlines)
{
    const long b = blockIdx.x;
    const long i = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd) return;
    const long item = i / nq;
    const int quad = (int)(i - item * nq);
    if (item >= n_items) return;
    const int seg = quad * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (seg * segp >= nw + 15) return;
    const bool storer = 2 * lane < segp;
    dbl2 v; v.x = 1.0; v.y = 2.0;
    const int my = order[item * M + (lane < M ? lane : 0)];
    for (int k = 0; k < M; ++k) {
        const int a = __builtin_amdgcn_readlane(my, k);
        if (a < 0) break;
        const int s = (a * (nw & 15)) & 15;
        const int p0 = seg * segp + 2 * lane;
        double *o = out + ((long)a * nw - s) + p0;
        const int b0 = p0 - s;
        if (storer && b0 >= 0 && b0 + 1 < nw) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(o));
    }
}

int main(int argc, char **argv)
{
    const int nlines = argc > 1 ? atoi(argv[1]) : 65536;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int nw = 2101;
    const long n = (long)nlines * nw;
    double *out;
    CK(hipMalloc(&out, (n + 256) * 8));
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
    const long chunks = n / 128;
    for (int K : {6, 16, 32})
        for (unsigned W : {2101u, 4202u}) {
            const long panels_n = (chunks + (long)K * W - 1) / ((long)K * W);
            const long useful = (panels_n * W + 3) / 4;
            const long per = (useful + 7) / 8;
            const float ms = timeit([&] { hipLaunchKernelGGL(panels, dim3((unsigned)(8 * per)), dim3(256), 0, 0, out, chunks, K, W, per, useful); });
            printf("A panels K=%2d W=%4u                      : %7.1f us %5.0f GB/s\n", K, W, ms * 1e3, n * 8 / ms / 1e6);
        }
    int *order_d;
    CK(hipMalloc(&order_d, (nlines + 64) * 2 * sizeof(int)));
    std::mt19937 rng(1);
    for (int segp : {112, 128})
        for (int tile : {0, 2048})
            for (int M : {8, 22, 32, 64}) {
                std::vector<int> lines(nlines);
                for (int i = 0; i < nlines; ++i) lines[i] = i;
                if (tile) for (int t0 = 0; t0 < nlines; t0 += tile) std::shuffle(lines.begin() + t0, lines.begin() + std::min(nlines, t0 + tile), rng);
                const int n_items = (nlines + M - 1) / M;
                std::vector<int> order((size_t)n_items * M, -1);
                for (int i = 0; i < nlines; ++i) order[i] = lines[i];
                CK(hipMemcpy(order_d, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
                const int nseg = (nw + 15 + segp - 1) / segp, nq = (nseg + 3) / 4;
                const long total = (long)n_items * nq, per = (total + 7) / 8;
                const float ms = timeit([&] { hipLaunchKernelGGL(rowsegs, dim3((unsigned)(8 * per)), dim3(256), 0, 0, out, order_d, n_items, M, nw, nq, segp, per, nlines); });
                printf("%s seg %3d doubles, %2d lines/item, %s : %7.1f us %5.0f GB/s\n", tile ? "C" : "B", segp, M,
                       tile ? "random in tiles of 2048" : "consecutive lines      ", ms * 1e3, n * 8 / ms / 1e6);
            }
    return 0;
}
