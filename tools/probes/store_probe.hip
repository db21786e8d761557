// store_probe.hip -- micro-probe of HBM write patterns on MI355X (tuning aid, not product).
//   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o gpurun_out/store_probe && gpurun_out/store_probe [GB]
// Each kernel writes the same number of bytes; only the access shape differs.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// 8 B per lane, grid-stride (wave chunk = 512 B aligned)
template <bool NT> __global__ void fill_x2(double *p, long n, double v)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
// 16 B per lane
template <bool NT> __global__ void fill_x4(double2 *p, long n2, double v)
{
    const double2 vv = make_double2(v, v);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        if (NT) { __builtin_nontemporal_store(v, &p[i].x); __builtin_nontemporal_store(v, &p[i].y); } else p[i] = vv;
    }
}
// 8 B per lane + one 64-B record read per step (all lanes of a wave read the same record): the flat LUT kernel minus math
__global__ void fill_x2_rec(double *p, long n, const double *rec, long nrec, int nw)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long a = (i / nw) % nrec;
        const double2 *r = reinterpret_cast<const double2 *>(rec + a * 8);
        const double2 r0 = r[0], r1 = r[1];
        const double r2 = rec[a * 8 + 4];
        __builtin_nontemporal_store(r0.x + r0.y + r1.x + r1.y + r2, p + i);
    }
}
// row pattern of the first LUT kernel: one block per row of `nphi` lines of `nw` doubles; thread t writes bands t+256p
__global__ void fill_rows(double *p, int nw, int nphi, double v)
{
    double *out = p + (long)blockIdx.x * nphi * nw;
    for (int l = 0; l < nphi; ++l, out += nw)
        for (int i = threadIdx.x; i < nw; i += 256) __builtin_nontemporal_store(v + l, out + i);
}
// same bytes, but each block owns a contiguous region and writes it in aligned 2 KiB steps (256 thr x 8 B)
__global__ void fill_block_contig(double *p, long per_block, double v)
{
    double *out = p + (long)blockIdx.x * per_block;
    for (long i = threadIdx.x; i < per_block; i += 256) __builtin_nontemporal_store(v, out + i);
}

template <class F> float timeit(F f, int reps = 5)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int nphi = 361;
    const long rows = argc > 1 ? atol(argv[1]) : 8281;
    for (int nw : {2101, 2112}) {
        const long n = rows * nphi * nw;
        double *p; CK(hipMalloc(&p, n * 8 + 4096));
        double *rec; const long nrec = 1 << 20; CK(hipMalloc(&rec, nrec * 64)); CK(hipMemset(rec, 0, nrec * 64));
        const double gb = n * 8 / 1e9;
        printf("---- nw=%d  %.2f GB ----\n", nw, gb);
        auto rep = [&](const char *name, float ms) { printf("%-34s %8.3f ms  %7.1f GB/s\n", name, ms, gb / ms * 1e3); };
        for (int blocks : {2048, 8192}) {
            char nm[64];
            snprintf(nm, 64, "fill_x2 nt, %d blocks", blocks);   rep(nm, timeit([&] { fill_x2<true><<<blocks, 256>>>(p, n, 1.0); }));
            snprintf(nm, 64, "fill_x2 plain, %d blocks", blocks); rep(nm, timeit([&] { fill_x2<false><<<blocks, 256>>>(p, n, 1.0); }));
            snprintf(nm, 64, "fill_x4 plain, %d blocks", blocks); rep(nm, timeit([&] { fill_x4<false><<<blocks, 256>>>((double2 *)p, n / 2, 1.0); }));
            snprintf(nm, 64, "fill_x2 + record loads, %d", blocks); rep(nm, timeit([&] { fill_x2_rec<<<blocks, 256>>>(p, n, rec, nrec, nw); }));
        }
        rep("fill_x2 nt, base+8B (misaligned)", timeit([&] { fill_x2<true><<<2048, 256>>>(p + 1, n, 1.0); }));
        rep("fill_x2 nt, base+64B", timeit([&] { fill_x2<true><<<2048, 256>>>(p + 8, n, 1.0); }));
        rep("fill_rows (row kernel pattern)", timeit([&] { fill_rows<<<rows, 256>>>(p, nw, nphi, 1.0); }));
        rep("fill_block_contig 8281 blocks", timeit([&] { fill_block_contig<<<rows, 256>>>(p, (long)nphi * nw, 1.0); }));
        rep("fill_block_contig 2048 blocks", timeit([&] { fill_block_contig<<<2048, 256>>>(p, n / 2048, 1.0); }));
        CK(hipFree(p)); CK(hipFree(rec));
    }
    return 0;
}
