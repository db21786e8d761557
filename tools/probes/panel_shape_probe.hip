// panel_shape_probe.hip -- what does the store pattern of the flat kernels deliver as a function of the panel shape
// (K steps x W waves of 1-KiB chunks, XCD x owns a contiguous run of panels) on a stream-sized output, without any
// arithmetic: (a) bare stores, (b) with ONE dependent scalar load per step (the per-line kernel reads a line's record
// through the scalar cache before it can form the step's samples), (c) the same with a prologue of P dependent-free
// vector loads per lane (the per-line kernel derives its band constants from 22 of them).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/panel_shape_probe.hip -o /tmp/panel_shape_probe;  /tmp/panel_shape_probe [lines] [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double dbl2 __attribute__((ext_vector_type(2)));

// XSPLIT: 0 = XCD x owns one contiguous run of logical blocks of the whole launch (whole panels; what the flat kernels do);
//         1 = every panel is split over the eight XCDs, XCD x takes a contiguous range of its columns: a panel that fits
//             the machine's wave slots is worked on by all waves at once, so few rows are in flight whatever K
template <int MODE, int XSPLIT>
__global__ __launch_bounds__(256) void panels(double *out, long chunks, int K, unsigned W, long per_xcd_blocks, long useful,
                                              const double *__restrict__ rec, int da, const double *__restrict__ table, int nw)
{
    const long b = blockIdx.x;
    long block;
    if (XSPLIT == 0) {
        block = (b & 7) * per_xcd_blocks + (b >> 3);
        if ((b >> 3) >= per_xcd_blocks || block >= useful) return;
    } else {
        // per_xcd_blocks = blocks per XCD per panel (ceil); blocks per panel bp = ceil(W / 4)
        const long bp = (W + 3) / 4, slots = 8 * per_xcd_blocks;
        const long panel = b / slots, i = b - panel * slots;
        const long col = (i & 7) * per_xcd_blocks + (i >> 3);
        if (col >= bp) return;
        block = panel * bp + col;
        if (block >= useful) return;
    }
    unsigned panel, w;
    if (XSPLIT == 0) {
        const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(block * 4 + (threadIdx.x >> 6)));
        panel = wave / W;
        w = wave - panel * W;
    } else {
        const long bp = (W + 3) / 4;
        panel = (unsigned)(block / bp);
        w = (unsigned)__builtin_amdgcn_readfirstlane((int)((block - (long)panel * bp) * 4 + (threadIdx.x >> 6)));
        if (w >= W) return;                                  // the last block of a panel may be partly idle
    }
    const long c0 = (long)panel * K * W + w;
    const int lane = threadIdx.x & 63;
    dbl2 v; v.x = 1.0; v.y = 2.0;
    if (MODE == 2) {                                         // prologue: 22 loads from an L2-resident table
        const int band = (int)((c0 * 128 + 2 * lane) % nw);
#pragma unroll
        for (int q = 0; q < 11; ++q) { v.x += table[(long)q * nw + band]; v.y += table[(long)q * nw + (band + 1 < nw ? band + 1 : 0)]; }
    }
    long a = (long)panel * K * da + (w * 128) / nw;          // the line of the chunk start
    for (int k = 0; k < K; ++k) {
        const long c = c0 + (long)k * W;
        dbl2 x = v;
        if (MODE >= 1) x.x += rec[a * 16];                   // wave-uniform address: scalar load, the store depends on it
        a += da;
        if (c < chunks) __builtin_nontemporal_store(x, reinterpret_cast<dbl2 *>(out + c * 128 + 2 * lane));
    }
}

int main(int argc, char **argv)
{
    const long nlines = argc > 1 ? atol(argv[1]) : 1048576;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const int nw = 2101;
    const long n = nlines * nw;
    double *out, *rec, *table;
    CK(hipMalloc(&out, (n + 256) * 8));
    CK(hipMalloc(&rec, (nlines + 4096 * 130) * 16 * 8));
    CK(hipMemset(rec, 0, (nlines + 4096 * 130) * 16 * 8));
    CK(hipMalloc(&table, 11L * nw * 8));
    CK(hipMemset(table, 0, 11L * nw * 8));
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
    for (int xs = 0; xs < 2; ++xs)
        for (int mode = 2; mode < 3; mode += 2)
            for (int K : {6, 16, 64})
                for (unsigned W : {2101u, 4202u, 6303u, 16808u}) {
                    if (W % 4 != 0 && xs == 1 && false) continue;
                    const long panels_n = (chunks + (long)K * W - 1) / ((long)K * W);
                    const int da = (int)((long)W * 128 / nw);
                    long useful, per;
                    dim3 grid;
                    if (xs == 0) {
                        useful = (panels_n * W + 3) / 4;
                        per = (useful + 7) / 8;
                        grid = dim3((unsigned)(8 * per));
                    } else {
                        const long bp = (W + 3) / 4;
                        useful = panels_n * bp;
                        per = (bp + 7) / 8;
                        grid = dim3((unsigned)(panels_n * 8 * per));
                    }
                    const float ms = timeit([&] {
                        if (xs == 0 && mode == 0) hipLaunchKernelGGL((panels<0, 0>), grid, dim3(256), 0, 0, out, chunks, K, W, per, useful, rec, da, table, nw);
                        else if (xs == 0) hipLaunchKernelGGL((panels<2, 0>), grid, dim3(256), 0, 0, out, chunks, K, W, per, useful, rec, da, table, nw);
                        else if (mode == 0) hipLaunchKernelGGL((panels<0, 1>), grid, dim3(256), 0, 0, out, chunks, K, W, per, useful, rec, da, table, nw);
                        else hipLaunchKernelGGL((panels<2, 1>), grid, dim3(256), 0, 0, out, chunks, K, W, per, useful, rec, da, table, nw);
                    });
                    printf("%-18s %-42s K=%3d W=%5u : %8.1f us %5.0f GB/s\n", xs ? "XCDs split panels" : "XCDs own panels",
                           mode == 0 ? "bare stores" : "+ scalar load per step + 22-load prologue", K, W, ms * 1e3, n * 8 / ms / 1e6);
                }
    return 0;
}
