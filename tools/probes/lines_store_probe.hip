// lines_store_probe.hip -- the bare store pattern of stream_lines_kernel (tuning aid, not part of the product): a wave owns 64
// consecutive rows of nw doubles and writes them in band blocks - per block one 128-B cache line of every row, eight rows per
// store instruction (lane = (row of eight, sixteen bytes of the line)), non-temporal - with a little arithmetic in between or
// none.  How fast does HBM take that, by rows' length and by resident waves per CU (dynamic LDS as in the kernel)?
// Variant 2: 256 B of a row per block pair (four rows per store instruction, two consecutive lines each).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/lines_store_probe.hip -o tools/probes/lines_store_probe && tools/probes/lines_store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double dbl2 __attribute__((ext_vector_type(2)));

// RUN = cache lines of a row per store instruction (1: the kernel's pattern; 2, 4: longer runs per row, fewer rows per instruction)
// XCD_RANGES: workgroups b, b + 8, ... run on one XCD (round-robin dispatch); give XCD x the contiguous range x of the row groups
// (eight write windows, one per L2, as the flat kernels and the members stream have them) instead of every eighth group
template <int RUN, bool XCD_RANGES>
__global__ __launch_bounds__(64) void pattern(double *out, long n_rows, int nw, int spin)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    long group = blockIdx.x;
    if (XCD_RANGES) {
        const long groups = (n_rows + 63) / 64, per_xcd = (groups + 7) >> 3;
        group = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (group >= groups) return;
    }
    const long r0 = group * 64;
    if (r0 >= n_rows) return;
    constexpr int ROWS_PER_STORE = 8 / RUN;            // 64 lanes x 16 B = 8 lines
    const int sub = lane / (8 * RUN), q = lane % (8 * RUN);
    const int n_blocks = nw / (16 * RUN);              // whole runs only (the probe ignores the ragged ends)
    double v = 1.0 + lane;
    if (lane == 0 && spin < 0) lds[0] = v;             // keep the allocation
    for (int j = 0; j < n_blocks; ++j) {
        for (int s = 0; s < spin; ++s) v = __builtin_fma(v, 1.0000001, 1e-9);      // stand-in for the band block's arithmetic
#pragma unroll
        for (int i = 0; i < 64 / ROWS_PER_STORE; ++i) {
            const long row = r0 + ROWS_PER_STORE * i + sub;
            if (row < n_rows) {
                // the row's first full line: rows start on 8-byte boundaries, lines on 128-byte ones
                const long first = (row * nw + 15) & ~15L;
                dbl2 x;  x.x = v;  x.y = v + i;
                __builtin_nontemporal_store(x, reinterpret_cast<dbl2 *>(out + first + 16L * RUN * j + 2 * q));
            }
        }
    }
}

template <int RUN, bool XCD_RANGES = false>
static void run(double *buf, long n_rows, int nw, int lds_bytes, int spin)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned grid = (unsigned)(((n_rows + 63) / 64 + 7) / 8 * 8);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pattern<RUN, XCD_RANGES>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((pattern<RUN, XCD_RANGES>), dim3(grid), dim3(64), lds_bytes, 0, buf, n_rows, nw, spin);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)n_rows * (nw / (16 * RUN)) * (16 * RUN) * 8.0;
    printf("rows %8ld x %4d bands, %s, %d line(s) of a row per store, LDS %5d B per wave (%2d waves per CU), %3d FMAs per block: %8.1f us  %6.0f GB/s\n",
           n_rows, nw, XCD_RANGES ? "a range per XCD  " : "groups interleaved", RUN, lds_bytes, lds_bytes ? 163840 / lds_bytes : 32, spin, best * 1e3, bytes / best / 1e6);
}

int main(int argc, char **argv)
{
    const long n_rows = argc > 1 ? atol(argv[1]) : 1000000;
    double *buf;
    const int bands[] = {96, 128, 256};
    CK(hipMalloc(&buf, sizeof(double) * (size_t)(n_rows + 2) * 512));
    CK(hipMemset(buf, 0, sizeof(double) * (size_t)(n_rows + 2) * 512));
    for (int nw : bands) {
        for (int lds : {17680, 8192, 0}) {
            run<1>(buf, n_rows, nw, lds, 0);
        }
        run<4>(buf, n_rows, nw, 17680, 0);
        run<1, true>(buf, n_rows, nw, 17680, 0);
        run<1, true>(buf, n_rows, nw, 0, 0);
        run<4, true>(buf, n_rows, nw, 17680, 0);
    }
    CK(hipFree(buf));
    return 0;
}
