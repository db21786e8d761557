// store_probe2.hip -- does one scalar record load per wave-step slow a streaming store kernel? (tuning aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// 8 B per lane; one 40-B record per wave-step through the scalar cache; 5 FMAs
template <bool REC> __global__ void k_x2(double *p, long n, const double *rec, long nrec_mask, long rec_stride)
{
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * 4 + wib, nwaves = (long)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    const double b0 = 1.0 + lane, b1 = 2.0, b2 = 3.0, b3 = 4.0, b4 = 5.0;
    long r = wave;
    for (long c = wave; c * 64 < n; c += nwaves, r += rec_stride) {
        double v = b0;
        if (REC) { const double *q = rec + (r & nrec_mask) * 8; v = q[0] * b0 + q[1] * b1 + q[2] * b2 + q[3] * b3 + q[4] * b4; }
        __builtin_nontemporal_store(v, p + c * 64 + lane);
    }
}
// 32 B per lane (2 x dwordx4); one record per wave-step; 20 FMAs
template <bool REC> __global__ void k_x8(double *p, long n, const double *rec, long nrec_mask, long rec_stride)
{
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * 4 + wib, nwaves = (long)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    double b[4][5];
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 5; ++k) b[j][k] = 1.0 + lane + j + k;
    long r = wave;
    for (long c = wave; c * 256 < n; c += nwaves, r += rec_stride) {
        double v[4];
        if (REC) {
            const double *q = rec + (r & nrec_mask) * 8;
            for (int j = 0; j < 4; ++j) v[j] = q[0] * b[j][0] + q[1] * b[j][1] + q[2] * b[j][2] + q[3] * b[j][3] + q[4] * b[j][4];
        } else for (int j = 0; j < 4; ++j) v[j] = b[j][0];
        double *o = p + c * 256 + lane * 4;
        __builtin_nontemporal_store(v[0], o); __builtin_nontemporal_store(v[1], o + 1);
        __builtin_nontemporal_store(v[2], o + 2); __builtin_nontemporal_store(v[3], o + 3);
    }
}
// 16 B per lane, 2 chunks of 1 KB per wave-step (lane writes elements 2*lane, 2*lane+1 of each 128-double chunk)
template <bool REC> __global__ void k_x4(double *p, long n, const double *rec, long nrec_mask, long rec_stride)
{
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * 4 + wib, nwaves = (long)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    double b[2][5];
    for (int j = 0; j < 2; ++j) for (int k = 0; k < 5; ++k) b[j][k] = 1.0 + lane + j + k;
    long r = wave;
    for (long c = wave; c * 128 < n; c += nwaves, r += rec_stride) {
        double v[2];
        if (REC) {
            const double *q = rec + (r & nrec_mask) * 8;
            for (int j = 0; j < 2; ++j) v[j] = q[0] * b[j][0] + q[1] * b[j][1] + q[2] * b[j][2] + q[3] * b[j][3] + q[4] * b[j][4];
        } else for (int j = 0; j < 2; ++j) v[j] = b[j][0];
        double *o = p + c * 128 + lane * 2;
        __builtin_nontemporal_store(v[0], o); __builtin_nontemporal_store(v[1], o + 1);
    }
}
template <class F> float timeit(F f, int reps = 5)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}
int main()
{
    const long n = 8281L * 361 * 2101;
    double *p; CK(hipMalloc(&p, n * 8 + 4096));
    const long nrec = 1L << 22; double *rec; CK(hipMalloc(&rec, nrec * 64)); CK(hipMemset(rec, 0, nrec * 64));
    const double gb = n * 8 / 1e9;
    auto rep = [&](const char *name, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, gb / ms * 1e3); };
    for (int blocks : {2101, 4202, 8404}) {
        char nm[80];
        snprintf(nm, 80, "x2 no record, %d blocks", blocks);            rep(nm, timeit([&] { k_x2<false><<<blocks, 256>>>(p, n, rec, nrec - 1, 256); }));
        snprintf(nm, 80, "x2 + scalar record/step, %d blocks", blocks); rep(nm, timeit([&] { k_x2<true><<<blocks, 256>>>(p, n, rec, nrec - 1, 256); }));
        snprintf(nm, 80, "x4 no record, %d blocks", blocks);            rep(nm, timeit([&] { k_x4<false><<<blocks, 256>>>(p, n, rec, nrec - 1, 256); }));
        snprintf(nm, 80, "x4 + scalar record/step, %d blocks", blocks); rep(nm, timeit([&] { k_x4<true><<<blocks, 256>>>(p, n, rec, nrec - 1, 256); }));
        snprintf(nm, 80, "x8 no record, %d blocks", blocks);            rep(nm, timeit([&] { k_x8<false><<<blocks, 256>>>(p, n, rec, nrec - 1, 256); }));
        snprintf(nm, 80, "x8 + scalar record/step, %d blocks", blocks); rep(nm, timeit([&] { k_x8<true><<<blocks, 256>>>(p, n, rec, nrec - 1, 256); }));
    }
    return 0;
}
