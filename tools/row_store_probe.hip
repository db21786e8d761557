// row_store_probe.hip -- what write rate does a BUCKETED stream expansion get on MI355X?  (tuning aid, not product)
//   hipcc --offload-arch=gfx950 -O3 tools/row_store_probe.hip -o gpurun_out/row_store_probe && gpurun_out/row_store_probe
// The stream output is out[line][band] with nw = 2101 bands: a row is 16 808 B and starts on an 8-B boundary only.
// Lines that share a sun zenith AND the alignment class of their row ((line * nw) mod 16 doubles) can be written by
// one workgroup whose lanes keep the same bands for all of them: every wave store then covers whole 128-B lines.
// This probe writes random rows that way (5 FMAs per element from a per-line record, as the real kernel would) and
// compares with a flat aligned fill of the same bytes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef double dbl2 __attribute__((ext_vector_type(2)));

__global__ void fill_flat(dbl2 *p, long n2, double v)
{
    dbl2 vv; vv.x = v; vv.y = v;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(vv, p + i);
}

struct Item { int first, count, cls, pad; };

// One workgroup per item = up to M lines of one alignment class.  Thread t of pass p owns the doubles
// 2 (p TH + t), +1 counted from the 128-B boundary in front of the row start (the row starts `cls` doubles in).
template <int TH, int NP>
__global__ __launch_bounds__(TH) void rows_kernel(double *__restrict__ out, const int *__restrict__ order,
                                                   const Item *__restrict__ items, int n_items, int nw,
                                                   const double *__restrict__ coef, int xcd_map)
{
    long b = blockIdx.x;
    if (xcd_map) {                       // workgroups b, b+8, .. share an XCD: give each XCD one contiguous range of items
        const long per = (n_items + 7) / 8;
        b = (b & 7) * per + (b >> 3);
    }
    if (b >= n_items) return;
    const Item it = items[b];
    const int s = it.cls;
    double bt[NP][2][5];
    int code[NP];                        // 3 = both elements inside the row, 1 = first only, 2 = second only, 0 = none
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int off = 2 * (p * TH + (int)threadIdx.x);
        const int b0 = off - s, b1 = b0 + 1;
        code[p] = ((b0 >= 0 && b0 < nw) ? 1 : 0) | ((b1 >= 0 && b1 < nw) ? 2 : 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bt[p][j][0] = 1.0;
#pragma unroll
            for (int q = 1; q < 5; ++q) bt[p][j][q] = 1e-3 * (q + (off + j) % 7);
        }
    }
    for (int i = 0; i < it.count; ++i) {
        const int a = __builtin_amdgcn_readfirstlane(order[it.first + i]);
        const double *__restrict__ r = coef + (long)a * 8;
        const double c0 = r[0], c1 = r[1], c2 = r[2], c3 = r[3], c4 = r[4];
        double *row = out + (long)a * nw - s;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            dbl2 v;
            v.x = __builtin_fma(c4, bt[p][0][4], __builtin_fma(c3, bt[p][0][3], __builtin_fma(c2, bt[p][0][2], __builtin_fma(c1, bt[p][0][1], c0 * bt[p][0][0]))));
            v.y = __builtin_fma(c4, bt[p][1][4], __builtin_fma(c3, bt[p][1][3], __builtin_fma(c2, bt[p][1][2], __builtin_fma(c1, bt[p][1][1], c0 * bt[p][1][0]))));
            double *o = row + 2 * (p * TH + (int)threadIdx.x);
            if (code[p] == 3) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(o));
            else if (code[p] == 1) o[0] = v.x;
            else if (code[p] == 2) o[1] = v.y;
        }
    }
}

int main(int argc, char **argv)
{
    const int nlines = argc > 1 ? atoi(argv[1]) : 65536;
    const int nw = argc > 2 ? atoi(argv[2]) : 2101;
    const long n = (long)nlines * nw;
    double *out;
    CK(hipMalloc(&out, (n + 64) * 8));
    const int base_shift = 0;            // hipMalloc is 256-B aligned
    std::vector<double> coef((size_t)nlines * 8, 0.0);
    for (int a = 0; a < nlines; ++a) coef[(size_t)a * 8] = (double)a;      // v = a: lets the host check coverage
    double *coef_d;
    CK(hipMalloc(&coef_d, coef.size() * 8));
    CK(hipMemcpy(coef_d, coef.data(), coef.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto f, int reps) {
        f(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps;
    };
    {
        const float ms = timeit([&] { hipLaunchKernelGGL(fill_flat, dim3(256 * 16), dim3(256), 0, 0, (dbl2 *)out, n / 2, 1.0); }, argc > 3 ? atoi(argv[3]) : 10);
        printf("flat aligned NT fill                         : %7.3f ms  %6.0f GB/s\n", ms, n * 8 / ms / 1e6);
    }
    std::mt19937 rng(1);
    int *order_d; Item *items_d;
    CK(hipMalloc(&order_d, nlines * sizeof(int)));
    CK(hipMalloc(&items_d, (nlines + 4096) * sizeof(Item)));
    std::vector<double> host((size_t)n);
    bool checked = false;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    for (int tile : {16384}) {
        for (int M : {11, 45}) {
            std::vector<int> order;
            std::vector<Item> items;
            for (int t0 = 0; t0 < nlines; t0 += tile) {
                const int t1 = std::min(nlines, t0 + tile);
                std::vector<int> lines[16];
                std::vector<int> perm(t1 - t0);
                for (int i = 0; i < t1 - t0; ++i) perm[i] = t0 + i;
                std::shuffle(perm.begin(), perm.end(), rng);
                for (int a : perm) lines[(int)(((long)a * nw + base_shift) % 16)].push_back(a);
                std::vector<Item> tile_items;
                for (int c = 0; c < 16; ++c)
                    for (size_t f = 0; f < lines[c].size(); f += M) {
                        Item it;
                        it.first = (int)order.size();
                        it.count = (int)std::min((size_t)M, lines[c].size() - f);
                        it.cls = c;
                        it.pad = 0;
                        for (int i = 0; i < it.count; ++i) order.push_back(lines[c][f + i]);
                        tile_items.push_back(it);
                    }
                std::shuffle(tile_items.begin(), tile_items.end(), rng);   // classes interleaved as a real tile's groups would be
                items.insert(items.end(), tile_items.begin(), tile_items.end());
            }
            CK(hipMemcpy(order_d, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
            CK(hipMemcpy(items_d, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice));
            const int ni = (int)items.size();
            for (int xcd = 0; xcd < 2; ++xcd) {
                const int grid = xcd ? 8 * ((ni + 7) / 8) : ni;
                float ms256 = timeit([&] { hipLaunchKernelGGL((rows_kernel<256, 5>), dim3(grid), dim3(256), 0, 0, out, order_d, items_d, ni, nw, coef_d, xcd); }, reps);
                float ms512 = timeit([&] { hipLaunchKernelGGL((rows_kernel<512, 3>), dim3(grid), dim3(512), 0, 0, out, order_d, items_d, ni, nw, coef_d, xcd); }, reps);
                float ms1024 = timeit([&] { hipLaunchKernelGGL((rows_kernel<1024, 2>), dim3(grid), dim3(1024), 0, 0, out, order_d, items_d, ni, nw, coef_d, xcd); }, reps);
                printf("tile %6d lines, %2d lines/item, %5d items, xcd ranges %d : 256x5 %7.3f ms %5.0f GB/s | 512x3 %7.3f ms %5.0f GB/s | 1024x2(+tail lost) %7.3f ms %5.0f GB/s\n",
                       tile, M, ni, xcd, ms256, n * 8 / ms256 / 1e6, ms512, n * 8 / ms512 / 1e6, ms1024, n * 8 / ms1024 / 1e6);
            }
            if (!checked) {
                CK(hipMemset(out, 0xff, n * 8));
                hipLaunchKernelGGL((rows_kernel<512, 3>), dim3(ni), dim3(512), 0, 0, out, order_d, items_d, ni, nw, coef_d, 0);
                CK(hipMemcpy(host.data(), out, n * 8, hipMemcpyDeviceToHost));
                long bad = 0;
                for (long i = 0; i < n; ++i) if (host[(size_t)i] != (double)(i / nw)) ++bad;
                printf("coverage check (512x3): %ld wrong of %ld\n", bad, n);
                checked = true;
            }
        }
    }
    return 0;
}
