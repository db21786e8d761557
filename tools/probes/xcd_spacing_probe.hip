// xcd_spacing_probe.hip -- does the DISTANCE between the eight XCD write windows matter? (tuning aid)
// One big allocation; XCD x streams L GB from byte offset x * D with the store pattern of expand_flat_kernel
// (panels of 6 x 2101 1-KiB chunks, 16-B non-temporal stores).  Sweeps D.
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_spacing_probe.hip -o tools/xcd_spacing_probe
//   tools/xcd_spacing_probe [L GB = 2] [D from GB = 2] [D to GB = 8] [D step GB = 0.25] [allocations = 2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double dbl2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void pattern(double *p, long win_chunks, long dist_chunks, int K, unsigned W)
{
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long x = blockIdx.x & 7, i = blockIdx.x >> 3;
    const unsigned wave = (unsigned)(i * 4 + wib);
    const unsigned panel = wave / W, w = wave - panel * W;
    const long c0 = (long)panel * K * W + w;
    const int lane = threadIdx.x & 63;
    dbl2 v;
    v.x = 1.0 + lane;
    v.y = 2.0 + w;
    for (int k = 0; k < K; ++k) {
        const long c = c0 + (long)k * W;
        if (c < win_chunks) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(p + (x * dist_chunks + c) * 128 + 2 * lane));
    }
}

int main(int argc, char **argv)
{
    const double L = argc > 1 ? atof(argv[1]) : 2.0, d0 = argc > 2 ? atof(argv[2]) : 2.0, d1 = argc > 3 ? atof(argv[3]) : 8.0,
                 dd = argc > 4 ? atof(argv[4]) : 0.25;
    const int nalloc = argc > 5 ? atoi(argv[5]) : 2;
    const int K = 6;
    const unsigned W = 2101;
    const long GiB = 1L << 30, win_chunks = (long)(L * GiB / 1024);
    const size_t bytes = (size_t)((7 * d1 + L) * GiB) + (1 << 20);
    const long panels = (win_chunks + (long)K * W - 1) / ((long)K * W);
    const long blocks = 8 * ((panels * W + 3) / 4);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double *buf[8];
    for (int a = 0; a < nalloc; ++a) CK(hipMalloc(&buf[a], bytes));
    printf("window %.2f GiB per XCD (%.1f GiB per launch), allocations of %.1f GiB\n", L, 8 * L, bytes / (double)GiB);
    for (int a = 0; a < nalloc; ++a)
        for (double D = d0; D <= d1 + 1e-9; D += dd) {
            const long dist_chunks = (long)(D * GiB / 1024);
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(pattern, dim3((unsigned)blocks), dim3(256), 0, 0, buf[a], win_chunks, dist_chunks, K, W);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("alloc %d  D = %6.3f GiB: %.3f ms  %.0f GB/s\n", a, D, best, 8.0 * win_chunks * 1024 / best / 1e6);
        }
    return 0;
}
