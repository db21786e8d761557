// gort_stamps.h -- phase stamps for the kernels whose time is a chain of latencies rather than a rate.
//
// A build with -DGORT_STAMPS (python -m gort_amd.build --stamps -> gort_amd/libgort_amd_stamps.so; never the product
// library) makes the kernels named below record the 100 MHz wall clock (s_memrealtime) at their phase boundaries, per
// workgroup or wave, together with the XCC and HW_ID of the writing wave; tools/stamps.py runs the BASELINE configs on
// that library and prints where the time of a workgroup goes and how the workgroups of a launch lie in time.  Round 4
// found with it that a grid launch 2.7 waves per SIMD deep ends in age order at 22 / 28 / 39 us per CU, that one lane on
// the horizon held two of BASELINE config 2's three waves for 15 us, and that a line of the `-energy` path was a chain of
// four latencies (profiles/r04/stamps.log).  Without the flag every macro is empty and the kernels are
// bit for bit what they were.
//
//   GORT_STAMPS_DEFINE(name)         once per translation unit: the buffer and its C accessor gort_debug_stamps_<name>
//   GORT_STAMPS_BEGIN()              in the kernel: the wave's stamp registers
//   GORT_STAMP(k)                    k = 0 .. 6, a compile-time constant: record now
//   GORT_STAMPS_END(name, unit, w)   the thread for which `w` holds writes the unit's eight slots (7 stamps + placement)
#ifndef GORT_STAMPS_H
#define GORT_STAMPS_H

#ifdef GORT_STAMPS
#include <hip/hip_runtime.h>

namespace gort {
constexpr int STAMP_SLOTS = 8;              // per unit: stamps 0..6, then (XCC_ID << 32) | HW_ID
constexpr long STAMP_UNITS = 16384;
}

#define GORT_STAMPS_DEFINE(name)                                                                                        \
    __device__ long long g_stamps_##name[gort::STAMP_UNITS * gort::STAMP_SLOTS];                                        \
    extern "C" int gort_debug_stamps_##name(long long *out, int clear)                                                  \
    {                                                                                                                   \
        const size_t bytes = sizeof(long long) * gort::STAMP_UNITS * gort::STAMP_SLOTS;                                 \
        if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_##name), bytes) != hipSuccess) return -1;               \
        if (clear) {                                                                                                    \
            void *p = nullptr;                                                                                          \
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps_##name)) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess || \
                hipDeviceSynchronize() != hipSuccess)          /* the engines' streams do not wait for the null stream */ \
                return -1;                                                                                              \
        }                                                                                                               \
        return 0;                                                                                                       \
    }
#define GORT_STAMPS_BEGIN() long long stamp_t_[gort::STAMP_SLOTS - 1] = {0, 0, 0, 0, 0, 0, 0}
#define GORT_STAMP(k) (stamp_t_[k] = wall_clock64())
#define GORT_STAMP_ANCHOR(x) asm volatile("" : "+v"(x))        /* the value is there before the stamp behind it is taken */
#define GORT_STAMPS_END(name, unit, writer)                                                                             \
    do {                                                                                                                \
        if ((writer) && (long)(unit) < gort::STAMP_UNITS) {                                                             \
            unsigned stamp_x_, stamp_h_;                                                                                \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(stamp_x_));                                     \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(stamp_h_));                                      \
            long long *stamp_o_ = g_stamps_##name + (long)(unit) * gort::STAMP_SLOTS;                                   \
            for (int stamp_k_ = 0; stamp_k_ < gort::STAMP_SLOTS - 1; ++stamp_k_) stamp_o_[stamp_k_] = stamp_t_[stamp_k_]; \
            stamp_o_[gort::STAMP_SLOTS - 1] = ((long long)(stamp_x_ & 15u) << 32) | stamp_h_;                           \
        }                                                                                                               \
    } while (0)
#else
#define GORT_STAMPS_DEFINE(name)
#define GORT_STAMPS_BEGIN() do { } while (0)
#define GORT_STAMP(k) do { } while (0)
#define GORT_STAMP_ANCHOR(x) do { } while (0)
#define GORT_STAMPS_END(name, unit, writer) do { } while (0)
#endif

#endif
