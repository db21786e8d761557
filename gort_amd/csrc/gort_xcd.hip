// gort_xcd.hip -- what the flat kernels need to know about the eight XCDs of the part: is workgroup dispatch
// round-robin over them (then the static block -> XCD-range mapping is exact), how fast does each XCD write
// (duty weights), and a host-side self-test of the index arithmetic.  Tuning surface: include/gort_amd_tuning.h.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "gort_flat.h"

namespace gort {

bool expand_wants_xcd_slots(bool dispatch_round_robin)
{
    const int m = tuning().xcd_mode;
    return m == 2 || (m < 0 && !dispatch_round_robin);
}

// Host-side check of the index arithmetic the flat kernels rely on (no GPU needed; tests/test_host_abi.py):
// fast_div against '/', and the duty mapping as a bijection of the launch's workgroups onto the logical blocks.
// Returns 0, or the line of the first failed check.
int selftest_index_math()
{
    unsigned long long rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    const unsigned divisors[] = {1, 2, 3, 7, 128, 361, 2101, 2100, 4202, 32851, 65535, 65536, 1000003, 0x7fffffffu};
    for (unsigned d : divisors) {
        const FastDiv f = make_fast_div(d);
        const unsigned edge[] = {0u, 1u, d - 1, d, d + 1, 2 * d - 1 < 0x7fffffffu ? 2 * d - 1 : 0u, 0x7fffffffu, 0x7ffffffeu};
        for (unsigned n : edge) if (n <= 0x7fffffffu && fast_div(n, f) != n / d) return __LINE__;
        for (int k = 0; k < 200000; ++k) {
            const unsigned n = (unsigned)(next() >> 33);
            if (fast_div(n, f) != n / d) return __LINE__;
        }
    }
    for (int trial = 0; trial < 300; ++trial) {
        int w[8];
        for (int x = 0; x < 8; ++x) w[x] = trial == 0 ? 32 : (trial == 1 ? (x & 1 ? 27 : 32) : 8 + (int)(next() % 25));
        const long useful = trial < 2 ? 2044799 / (trial + 1) / 100 : 1 + (long)(next() % 20000);
        XcdDuty duty;
        const long grid = plan_xcd_duty(1, useful, w, duty);
        std::vector<unsigned char> seen((size_t)useful, 0);
        long hit = 0;
        for (long b = 0; b < grid; ++b) {
            const long blk = duty_logical_block(b, duty, useful);
            if (blk < 0) continue;
            if (blk >= useful || seen[(size_t)blk]) return __LINE__;
            seen[(size_t)blk] = 1;
            ++hit;
        }
        if (hit != useful) return __LINE__;
        XcdDuty plain;
        if (plan_xcd_duty(0, useful, w, plain) != useful || plan_xcd_duty(2, useful, nullptr, plain) != useful) return __LINE__;
    }
    return 0;
}

namespace {
// the store pattern of expand_flat_kernel (panels of K x W chunks, 16-B non-temporal stores) with the XCD
// mapping of mode 1; one workgroup in 64 reports when it started and ended
__global__ __launch_bounds__(256) void xcd_pattern_kernel(double *__restrict__ slab, long chunks, int K, unsigned W,
                                                          XcdDuty duty, long useful,
                                                          unsigned long long *__restrict__ t_first,
                                                          unsigned long long *__restrict__ t_last)
{
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long block = xcd_logical_block(1, duty, useful, nullptr);
    if (block < 0) return;
    const int xd = blockIdx.x & 7;
    const bool reports = t_first && (block & 63) == 0;
    if (reports && threadIdx.x == 0) atomicMin(&t_first[xd], wall_clock64());
    const unsigned wave = (unsigned)(block * 4 + wib);
    const unsigned panel = wave / W, w = wave - panel * W;
    const long c0 = (long)panel * K * W + w;
    const int lane = threadIdx.x & 63;
    dbl2 v;
    v.x = 0.0;
    v.y = 0.0;
    for (int k = 0; k < K; ++k) {
        const long c = c0 + (long)k * W;
        if (c < chunks) __builtin_nontemporal_store(v, reinterpret_cast<dbl2 *>(slab + c * CHUNK + EPL * lane));
    }
    __syncthreads();
    if (reports && threadIdx.x == 0) atomicMax(&t_last[xd], wall_clock64());
}
}  // namespace

int calibrate_xcd_weights(void *stream, double *slab, long n_doubles, int weights[8], double *pattern_gbs)
{
    if (pattern_gbs) *pattern_gbs = 0.0;
    for (int x = 0; x < 8; ++x) weights[x] = 32;
    // whole aligned chunks inside the slab
    const uintptr_t addr = reinterpret_cast<uintptr_t>(slab);
    const long skip = (long)(((addr + 1023) & ~(uintptr_t)1023) - addr) / 8;
    const long chunks = (n_doubles - skip) / CHUNK;
    constexpr int K = 6;
    constexpr unsigned W = 2101;
    if (chunks < 64L * K * W) return GORT_OK;                  // too small to say anything
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *dev = nullptr;
    if (hipMalloc(&dev, 16 * sizeof(*dev)) != hipSuccess) return fail(GORT_ENOMEM, "xcd calibration: hipMalloc failed");
    const long panels = (chunks + (long)K * W - 1) / ((long)K * W);
    const long useful = (panels * W + 3) / 4;
    int rc = GORT_OK;
    // ONE pass with equal weights: the rates of the XCDs while all of them run.  Iterating on the result drives
    // the slow XCDs' weights further down (to 25/32), which suits this bare store pattern but not the LUT kernel
    // (tools/probes/weights_sweep.py: 27-28 is its optimum, 25 already loses half the gain).
    for (int iter = 0; iter < 1 && rc == GORT_OK; ++iter) {
        XcdDuty duty;
        const long grid = plan_xcd_duty(1, useful, weights, duty);
        unsigned long long host[16];
        for (int x = 0; x < 8; ++x) { host[x] = ~0ull; host[8 + x] = 0; }
        hipError_t err = hipMemcpyAsync(dev, host, sizeof(host), hipMemcpyHostToDevice, s);
        if (err == hipSuccess) {
            hipLaunchKernelGGL(xcd_pattern_kernel, dim3((unsigned)grid), dim3(256), 0, s, slab + skip, chunks, K, W, duty,
                               useful, dev, dev + 8);
            if ((rc = check_launch("xcd_pattern_kernel"))) break;      // a failed launch must not be left for an unrelated check to find
            err = hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, s);
        }
        if (err == hipSuccess) err = hipStreamSynchronize(s);
        if (err != hipSuccess) { rc = fail(GORT_ENODEVICE, "xcd calibration: %s", hipGetErrorString(err)); break; }
        unsigned long long t0 = ~0ull;
        for (int x = 0; x < 8; ++x) if (host[x] < t0) t0 = host[x];
        // blocks per tick of every XCD in this run; the next weights are proportional to it
        double rate[8], rmax = 0.0;
        bool ok = true;
        for (int x = 0; x < 8; ++x) {
            if (host[8 + x] <= t0) { ok = false; break; }
            rate[x] = (double)weights[x] / (double)(host[8 + x] - t0);
            if (rate[x] > rmax) rmax = rate[x];
        }
        if (!ok) break;                                        // an XCD reported nothing: leave the weights alone
        if (pattern_gbs) {
            // the rate of the bare store pattern with equal shares, first start to last end (wall clock ticks)
            unsigned long long t1 = 0;
            for (int x = 0; x < 8; ++x) if (host[8 + x] > t1) t1 = host[8 + x];
            int khz = 0, device = 0;
            (void)hipGetDevice(&device);
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0 && t1 > t0)
                *pattern_gbs = (double)chunks * 1024.0 / ((double)(t1 - t0) / khz * 1e-3) / 1e9;
        }
        for (int x = 0; x < 8; ++x) {
            int w = (int)(32.0 * rate[x] / rmax + 0.5);
            weights[x] = w < 16 ? 16 : (w > 32 ? 32 : w);
        }
    }
    // What is being measured is a trait of the device - the XCDs of one XCC_ID parity write ~15 % slower than the
    // others on every MI355X seen so far (the odd XCC_IDs in every standalone probe, the even dispatch slots in one process) - under a few % of
    // run-to-run noise, and a weight that is off by one
    // costs more than it gains (tools/probes/weights_sweep.py).  So the eight results are averaged within each parity.
    if (rc == GORT_OK) {
        double mean[2] = {0.0, 0.0};
        for (int x = 0; x < 8; ++x) mean[x & 1] += 0.25 * weights[x];
        const double top = mean[0] > mean[1] ? mean[0] : mean[1];
        for (int x = 0; x < 8; ++x) {
            int w = (int)(32.0 * mean[x & 1] / top + 0.5);
            weights[x] = w < 16 ? 16 : (w > 32 ? 32 : w);
        }
    }
    (void)hipFree(dev);
    return rc;
}

// Are workgroups b, b+8, b+16, ... of a launch placed on one XCD each (round-robin dispatch, the documented
// behaviour of the multi-XCD dispatcher)?  Then the static mapping of the flat kernels is exact and needs no
// atomics.  A profiler or a partition mode may change the pattern, hence the probe rather than an assumption.
namespace {
__global__ void xcd_probe_kernel(int *__restrict__ xcc_of_block)
{
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc_of_block[blockIdx.x] = (int)(x & 7);
    }
}
}  // namespace

int probe_xcd_dispatch(void *stream, int *round_robin)
{
    constexpr int NB = 4096;
    *round_robin = 0;
    int *dev = nullptr;
    if (hipMalloc(&dev, sizeof(int) * NB) != hipSuccess) return fail(GORT_ENOMEM, "xcd probe: hipMalloc failed");
    int host[NB];
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(xcd_probe_kernel, dim3(NB), dim3(256), 0, s, dev);
    if (const int rc = check_launch("xcd_probe_kernel")) {
        (void)hipFree(dev);
        return rc;
    }
    hipError_t err = hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, s);
    if (err == hipSuccess) err = hipStreamSynchronize(s);
    (void)hipFree(dev);
    if (err != hipSuccess) return fail(GORT_ENODEVICE, "xcd probe: %s", hipGetErrorString(err));
    unsigned seen = 0;
    for (int b = 0; b < 8; ++b) seen |= 1u << host[b];
    bool ok = seen == 0xffu;
    for (int b = 8; b < NB && ok; ++b) ok = host[b] == host[b & 7];
    *round_robin = ok ? 1 : 0;
    return GORT_OK;
}

}  // namespace gort
