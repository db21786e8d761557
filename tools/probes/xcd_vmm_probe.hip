// xcd_vmm_probe.hip -- does SCATTERING a slab over physical memory with the virtual-memory API help? (tuning aid)
// Physical memory is created in granules (hipMemCreate) and mapped into one VA range either in creation order
// (what hipMalloc gives: physically near-contiguous) or strided through the whole pool, so that neighbouring
// VA granules lie far apart physically.  Then the eight-window store pattern of expand_flat_kernel is timed.
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_vmm_probe.hip -o tools/xcd_vmm_probe
//   tools/xcd_vmm_probe [granule MiB = 1024] [pool GiB = 192] [slab GiB = 16] [stride = 37]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
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

static float run(double *p, long slab_bytes)
{
    const int K = 6;
    const unsigned W = 2101;
    const long win_chunks = slab_bytes / 8 / 1024;          // eight windows back to back
    const long panels = (win_chunks + (long)K * W - 1) / ((long)K * W);
    const long blocks = 8 * ((panels * W + 3) / 4);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(pattern, dim3((unsigned)blocks), dim3(256), 0, 0, p, win_chunks, win_chunks, K, W);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const size_t gran = (size_t)(argc > 1 ? atol(argv[1]) : 1024) << 20;
    const size_t pool = (size_t)(argc > 2 ? atol(argv[2]) : 192) << 30;
    const size_t slab = (size_t)(argc > 3 ? atol(argv[3]) : 16) << 30;
    const long stride = argc > 4 ? atol(argv[4]) : 37;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t g_min = 0, g_rec = 0;
    CK(hipMemGetAllocationGranularity(&g_min, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&g_rec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu, recommended %zu; using granules of %zu MiB\n", g_min, g_rec, gran >> 20);
    const long nh = (long)(pool / gran), ns = (long)(slab / gran);
    std::vector<hipMemGenericAllocationHandle_t> h(nh);
    for (long i = 0; i < nh; ++i) CK(hipMemCreate(&h[i], gran, &prop, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int mode = 0; mode < 4; ++mode) {
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, slab, 0, nullptr, 0));
        for (long j = 0; j < ns; ++j) {
            long src = j;                                   // mode 0: creation order
            if (mode == 1) src = (j * stride) % nh;         // mode 1: strided through the pool
            if (mode == 2) src = (j % 8) * (nh / 8) + j / 8; // mode 2: round-robin over eight eighths of the pool
            if (mode == 3) { const long w = j * 8 / ns; src = w * (nh / 8) + (j - (w * ns + 7) / 8); }   // mode 3: window x of the slab = one contiguous run in eighth x of the pool
            CK(hipMemMap((char *)va + j * gran, gran, 0, h[src], 0));
        }
        CK(hipMemSetAccess(va, slab, &acc, 1));
        const float ms = run((double *)va, (long)slab);
        printf("mode %d (%s): %.3f ms  %.0f GB/s\n", mode,
               mode == 0 ? "creation order" : (mode == 1 ? "strided" : (mode == 2 ? "round-robin over eighths" : "window-aligned runs")), ms, slab / ms / 1e6);
        CK(hipMemUnmap(va, slab));
        CK(hipMemAddressFree(va, slab));
    }
    for (long i = 0; i < nh; ++i) CK(hipMemRelease(h[i]));
    return 0;
}
