// xcd_stream_probe.hip -- the store pattern of expand_flat_kernel without its arithmetic (tuning aid):
// panels of K steps x W waves of 1-KiB chunks, one contiguous run of panels per XCD (static round-robin
// mapping), 16-B non-temporal stores.  Reports, per allocation, the kernel time and when each XCD finished:
// is a slow allocation slow on every XCD, or do a few XCDs lag and the in-order dispatcher makes the rest wait?
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_stream_probe.hip -o tools/xcd_stream_probe
//   tools/xcd_stream_probe [GB per slab = 50.25] [slabs = 4] [K = 6] [W = 2101] [rot (unused)] [D = 1] [w_even = D] [w_odd = D]: XCD x uses w of every D workgroups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef double dbl2 __attribute__((ext_vector_type(2)));

struct Duty { int D; int w[8]; long start[8], n[8]; };       // XCD x uses w[x] of every D of its workgroups

__global__ __launch_bounds__(256) void pattern(double *p, long chunks, int K, unsigned W, int rot, Duty duty, unsigned long long *t_first,
                                                unsigned long long *t_last, int *xcc_seen)
{
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xd = blockIdx.x & 7;                             // XCD of this block (round-robin dispatch)
    const long i = blockIdx.x >> 3;
    const long cyc = i / duty.D;
    const int ph = (int)(i - cyc * duty.D);
    const long li = cyc * duty.w[xd] + ph;
    if (ph >= duty.w[xd] || li >= duty.n[xd]) return;          // idle slot of a slow XCD
    (void)rot;
    const long block = duty.start[xd] + li;
    const unsigned wave = (unsigned)(block * 4 + wib);
    const unsigned panel = wave / W, w = wave - panel * W;
    const long c0 = (long)panel * K * W + w;
    const int lane = threadIdx.x & 63;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    const bool sampled = (li & 127) == 0;       // one workgroup in 128 reports (atomics are ~200 ns)
    if (threadIdx.x == 0 && sampled) {
        atomicMin(&t_first[xcc], wall_clock64());
        if (blockIdx.x < 8) xcc_seen[blockIdx.x] = (int)xcc;
    }
    dbl2 v;
    v.x = 1.0 + lane;
    v.y = 2.0 + w;
    for (int k = 0; k < K; ++k) {
        const long c = c0 + (long)k * W;
        if (c < chunks) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(p + c * 128 + 2 * lane));
    }
    __syncthreads();
    if (threadIdx.x == 0 && sampled) atomicMax(&t_last[xcc], wall_clock64());
}

int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 50.2465;
    const int nslab = argc > 2 ? atoi(argv[2]) : 4;
    const int K = argc > 3 ? atoi(argv[3]) : 6;
    const unsigned W = argc > 4 ? (unsigned)atoi(argv[4]) : 2101u;
    const int rot = argc > 5 ? atoi(argv[5]) : 0;
    const long chunks = (long)(gb * 1e9 / 1024);
    const long panels = (chunks + (long)K * W - 1) / ((long)K * W);
    const long useful = (panels * W + 3) / 4;
    Duty duty;
    duty.D = argc > 6 ? atoi(argv[6]) : 1;
    const int w_even = argc > 7 ? atoi(argv[7]) : duty.D, w_odd = argc > 8 ? atoi(argv[8]) : duty.D;
    long sumw = 0, acc = 0, cycles = 0;
    for (int x = 0; x < 8; ++x) { duty.w[x] = (x & 1) ? w_odd : w_even; sumw += duty.w[x]; }
    for (int x = 0; x < 8; ++x) {
        duty.start[x] = acc;
        duty.n[x] = x == 7 ? useful - acc : useful * duty.w[x] / sumw;
        acc += duty.n[x];
        const long c = (duty.n[x] + duty.w[x] - 1) / duty.w[x];
        if (c > cycles) cycles = c;
    }
    const long blocks = 8 * cycles * duty.D;
    printf("duty: %d of %d on even XCDs, %d of %d on odd; %ld useful of %ld workgroups\n", w_even, duty.D, w_odd, duty.D, useful, blocks);
    unsigned long long *t;
    int *seen;
    CK(hipMalloc(&t, 16 * sizeof(*t)));
    CK(hipMalloc(&seen, 8 * sizeof(int)));
    double *slab[16];
    for (int s = 0; s < nslab; ++s) CK(hipMalloc(&slab[s], chunks * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int clock_khz = 0;
    CK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeWallClockRate, 0));
    printf("%.2f GB per slab, %ld chunks, panels of %d x %u, %ld workgroups; wall clock %d kHz\n", gb, chunks, K, W, blocks, clock_khz);
    for (int round = 0; round < 2; ++round)
        for (int s = 0; s < nslab; ++s) {
            float best = 1e9f;
            unsigned long long hf[8], hl[8];
            int hs[8];
            for (int rep = 0; rep < 6; ++rep) {
                unsigned long long init[16];
                for (int i = 0; i < 8; ++i) { init[i] = ~0ull; init[8 + i] = 0; }
                CK(hipMemcpy(t, init, sizeof(init), hipMemcpyHostToDevice));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(pattern, dim3((unsigned)blocks), dim3(256), 0, 0, slab[s], chunks, K, W, rot, duty, t, t + 8, seen);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) {
                    best = ms;
                    CK(hipMemcpy(hf, t, sizeof(hf), hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hl, t + 8, sizeof(hl), hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hs, seen, sizeof(hs), hipMemcpyDeviceToHost));
                }
            }
            unsigned long long f0 = ~0ull;
            for (int i = 0; i < 8; ++i) if (hf[i] < f0) f0 = hf[i];
            printf("round %d slab %d (%p): %.3f ms = %.0f GB/s; XCD finish times (ms after first start):", round, s, (void *)slab[s], best,
                   chunks * 1024.0 / best / 1e6);
            for (int i = 0; i < 8; ++i) printf(" %.2f", (double)(hl[i] - f0) / clock_khz);
            printf("   xcc of blocks 0..7:");
            for (int i = 0; i < 8; ++i) printf(" %d", hs[i]);
            printf("\n");
        }
    return 0;
}
