// What does the shader clock do under short fp64-dense kernels?  A kernel of `chain` dependent v_fma_f64 per wave,
// `waves` waves per SIMD on every CU, launched back to back for `seconds`; wave 0 of workgroup 0 reads s_memtime
// (shader clock) and the 100 MHz wall clock around its chain.  Prints MHz and cycles per dependent FMA over time.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

__global__ __launch_bounds__(256) void chain_kernel(double *out, long long *ticks, int chain, int ilp)
{
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999, c = 1e-7;
    double a2 = a + 1.0, a3 = a + 2.0, a4 = a + 3.0;
    const long long t0 = clock64(), w0 = wall_clock64();
    if (ilp == 1) {
        for (int i = 0; i < chain; i += 32) {
#pragma unroll
            for (int k = 0; k < 32; ++k) a = __builtin_fma(a, b, c);                 // one dependent chain, loop overhead 1/32
        }
    } else if (ilp == 2) {
        for (int i = 0; i < chain; i += 32) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { a = __builtin_fma(a, b, c);  a2 = __builtin_fma(a2, b, c); }
        }
    } else if (ilp == 0) {                                                           // every eighth instruction a reciprocal
        for (int i = 0; i < chain; i += 32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int j = 0; j < 7; ++j) a = __builtin_fma(a, b, c);
                a = __builtin_amdgcn_rcp(a);
            }
        }
    } else {
        for (int i = 0; i < chain; i += 4) {
            a = __builtin_fma(a, b, c);  a2 = __builtin_fma(a2, b, c);  a3 = __builtin_fma(a3, b, c);  a4 = __builtin_fma(a4, b, c);
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = a + a2 + a3 + a4;
    if (blockIdx.x == 0 && threadIdx.x == 0) { ticks[0] = t1 - t0;  ticks[1] = w1 - w0; }
}

__global__ void empty_kernel(long long *t) { if (t && threadIdx.x == 0) t[2] = wall_clock64(); }

// the floor under every short call: one empty kernel on a non-blocking stream, launch to the return of the synchronisation
static void launch_floor()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);  hipEventCreate(&e1);
    long long *t;
    hipMalloc(&t, 64);
    for (int variant = 0; variant < 6; ++variant) {
        std::vector<double> us;
        for (int i = 0; i < 2200; ++i) {
            const auto l0 = std::chrono::steady_clock::now();
            if (variant == 2) hipEventRecord(e0, s);
            hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, t);
            if (variant == 2) hipEventRecord(e1, s);
            if (variant == 1) hipStreamSynchronize(0);          // a second, idle stream synchronised first
            if (variant == 3) { while (hipStreamQuery(s) == hipErrorNotReady) { } }
            else if (variant == 4) { hipEventRecord(e1, s); while (hipEventQuery(e1) == hipErrorNotReady) { } }
            else if (variant == 5) { hipEventRecord(e1, s); hipEventSynchronize(e1); }
            else hipStreamSynchronize(s);
            const auto l1 = std::chrono::steady_clock::now();
            if (i >= 200) us.push_back(std::chrono::duration<double>(l1 - l0).count() * 1e6);
        }
        std::sort(us.begin(), us.end());
        printf("empty kernel, launch + stream synchronisation%s: min %.1f  median %.1f  p90 %.1f us\n",
               variant == 0 ? "" : (variant == 1 ? " (an idle stream synchronised first)" : (variant == 2 ? " (an event recorded in front and behind)" :
               (variant == 3 ? " (hipStreamQuery polled instead)" : (variant == 4 ? " (an event behind it, hipEventQuery polled)" : " (an event behind it, hipEventSynchronize)")))),
               us[0], us[us.size() / 2], us[us.size() * 9 / 10]);
    }
}

int main(int argc, char **argv)
{
    launch_floor();
    const int chain = argc > 1 ? atoi(argv[1]) : 8192;
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    double *out;  long long *ticks, h[2];
    hipMalloc(&out, sizeof(double) * 256 * cus * 8);
    hipMalloc(&ticks, 16);
    for (int ilp : {1, 2, 4, 0})
        for (int wgs_per_cu : {1, 2, 3, 4}) {                        // 256-thread workgroups: 1 / 2 / 3 waves per SIMD
            const auto start = std::chrono::steady_clock::now();
            double next = 0.0;
            for (;;) {
                const auto l0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(chain_kernel, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, ticks, chain, ilp);
                hipDeviceSynchronize();
                const auto l1 = std::chrono::steady_clock::now();
                const double t = std::chrono::duration<double>(l1 - start).count();
                if (t >= next) {
                    hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
                    printf("ilp %d  %d waves/SIMD  t=%.3f s  call %.1f us  shader clock %.0f MHz  %.2f cycles per FMA of one wave (%lld ticks)\n", ilp, wgs_per_cu, t,
                           std::chrono::duration<double>(l1 - l0).count() * 1e6, h[0] / (h[1] / 100.0), (double)h[0] / chain, h[0]);
                    next += 0.1;
                }
                if (t > 0.15) break;
            }
        }
    return 0;
}
