// store_policy_probe.hip -- the cache-policy bits of a streaming 16-B store on gfx950 (nt, sc0, sc1 and their
// combinations) on the flat kernels' store pattern: panels of K steps x W waves of 1-KiB chunks, XCD x owns a contiguous
// run of panels.  The kernels use __builtin_nontemporal_store (= "nt"); is any other flavour faster?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_policy_probe.hip -o /tmp/store_policy_probe; /tmp/store_policy_probe [lines] [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double dbl2 __attribute__((ext_vector_type(2)));

template <int POLICY>
__device__ __forceinline__ void store16(dbl2 *p, dbl2 v)
{
    if (POLICY == 0) *p = v;
    else if (POLICY == 1) __builtin_nontemporal_store(v, p);
    else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    else if (POLICY == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int POLICY>
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
        if (c < chunks) store16<POLICY>(reinterpret_cast<dbl2 *>(out + c * 128 + 2 * lane), v);
    }
}

int main(int argc, char **argv)
{
    const long nlines = argc > 1 ? atol(argv[1]) : 1048576;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const int nw = 2101;
    const long n = nlines * nw;
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
    const char *names[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
    for (int K : {6, 64})
        for (unsigned W : {2101u, 16808u}) {
            if ((K == 6) != (W == 2101u)) continue;          // the two shapes in use: 6 x 2101 (LUT kernel), 64 x 16808 (per-line stream kernel)
            const long panels_n = (chunks + (long)K * W - 1) / ((long)K * W);
            const long useful = (panels_n * W + 3) / 4;
            const long per = (useful + 7) / 8;
            const dim3 grid((unsigned)(8 * per));
            for (int pol = 0; pol < 8; ++pol) {
                const float ms = timeit([&] {
                    switch (pol) {
                        case 0: hipLaunchKernelGGL(panels<0>, grid, dim3(256), 0, 0, out, chunks, K, W, per, useful); break;
                        case 1: hipLaunchKernelGGL(panels<1>, grid, dim3(256), 0, 0, out, chunks, K, W, per, useful); break;
                        case 2: hipLaunchKernelGGL(panels<2>, grid, dim3(256), 0, 0, out, chunks, K, W, per, useful); break;
                        case 3: hipLaunchKernelGGL(panels<3>, grid, dim3(256), 0, 0, out, chunks, K, W, per, useful); break;
                        case 4: hipLaunchKernelGGL(panels<4>, grid, dim3(256), 0, 0, out, chunks, K, W, per, useful); break;
                        case 5: hipLaunchKernelGGL(panels<5>, grid, dim3(256), 0, 0, out, chunks, K, W, per, useful); break;
                        case 6: hipLaunchKernelGGL(panels<6>, grid, dim3(256), 0, 0, out, chunks, K, W, per, useful); break;
                        default: hipLaunchKernelGGL(panels<7>, grid, dim3(256), 0, 0, out, chunks, K, W, per, useful); break;
                    }
                });
                printf("K=%2d W=%5u  %-12s : %8.1f us %5.0f GB/s\n", K, W, names[pol], ms * 1e3, n * 8 / ms / 1e6);
            }
        }
    return 0;
}
