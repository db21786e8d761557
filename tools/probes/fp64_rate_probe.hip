// What does the fp64 VALU of this device sustain?  (1) dependent-free v_fma_f64 chains, (2) the same with one
// v_rcp_f64 per 30 FMAs (the mix of the per-line stream kernel).  Tells whether a kernel at N instructions per
// sample is at the issue limit: rate = waves x instructions x 64 lanes / time.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/fp64_rate_probe.hip -o /tmp/fp64_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int RCP>
__global__ __launch_bounds__(256) void fma_kernel(double *out, int iters, double a, double b)
{
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], b, a);
        if (RCP) x[it & 7] = __builtin_amdgcn_rcp(x[it & 7]);      // 1 rcp per 32 FMAs
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678) out[0] = s;
}

int main()
{
    double *out;
    hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * 8 * 4;
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode) hipLaunchKernelGGL(fma_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5, 0.999);
            else hipLaunchKernelGGL(fma_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5, 0.999);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)blocks * 256 * iters * 32;          // lane-FMAs
            const double per_simd_cycles = instr / 64 / 1024 * 4;           // at 4 cycles per wave instruction
            printf("%s: %.3f ms  %.1f TFLOP/s fp64  -> %.2f GHz if every SIMD issued one FMA per 4 cycles\n",
                   mode ? "fma + 1 rcp per 32" : "fma only         ", ms, 2 * instr / ms / 1e9, per_simd_cycles / ms / 1e6);
        }
    return 0;
}
