// gort_flat.h -- host-side planning shared by the two "flat" expansion kernels (gort_lut_expand.hip,
// gort_stream_expand.hip): the output is cut into 1-KiB chunks aligned in ABSOLUTE address, a wave takes chunks
// c, c + W, c + 2W, ... with W a multiple of nw / gcd(nw, 128) so that a lane keeps its bands for life, the chunks
// are worked through in panels of K steps x W waves, and every XCD owns one contiguous run of panels
// (DESIGN.md 5.1).  Device-side counterparts (fast_div, xcd_logical_block, CHUNK) live in gort_device.h.
#ifndef GORT_FLAT_H
#define GORT_FLAT_H

#include <cstdlib>
#include <cstring>

#include "gort_device.h"

namespace gort {
namespace {

// Tuning knobs of the flat kernels (A/B switches: the measuring build only, gort_internal.h; read once):
//   GORT_EXPAND_DEPTH    1 | 2 | 4                coefficient records in flight per lane (LUT kernel)
//   GORT_EXPAND_NT       1 | 0                    non-temporal stores
//   GORT_EXPAND_WAVES    target wave stride of expand_flat_kernel in chunks = waves per panel (rounded to a
//                        band-preserving multiple of nw/gcd(nw,128))
//   GORT_EXPAND_STEPS    steps per wave = panel height; 0 = one panel, every wave strides through the whole slab
//   GORT_EXPAND_XCD      0 | 1 | 2                XCD mapping, see xcd_logical_block(); default automatic
//   GORT_STREAM_WAVES    waves per panel of expand_flat_stream_kernel; GORT_STREAM_STEPS its steps per wave
// Measured on the 50.25 GB metric slab, four slabs held at once per run (profiles/r01/tune_panels*.log):
//   whole-slab strides (steps 0, stride 33616)   8.1-8.3 ms, 9.5 ms on some allocations
//   panels of 6 steps x 2101 waves, XCD mode 1   7.04-7.30 ms (6.9-7.1 TB/s), 7.8 ms on some allocations
//   the same with 4 steps                        7.8 ms on every allocation (prologue-bound)
//   the same, XCD mode 0 (interleaved)           8.2 ms
//   XCD mode 2 with panels                       one returning atomic per workgroup costs ~190 ns on its
//                                                counter's line: 9.2 ms at 16 steps before the counters were
//                                                spread over 8 lines, then fine from 8 steps up
// The spread between allocations of one size is a property of where the slab lies physically (the same slab
// is slow or fast at any offset and for the whole run); short panels narrow it from 18 % to 10 %.
struct ExpandTuning {
    bool nt = true;
    // GORT_EXPAND_XCD: 0 = logical blocks interleaved over the XCDs, 1 = one contiguous range per XCD assuming
    // round-robin dispatch, 2 = the same by the real XCC_ID through per-XCD slot counters;
    // -1 = automatic: 1 where dispatch is round-robin over the XCDs (probed once per engine), else 2
    int xcd_mode = -1;
    int depth = 2;
    int steps = -1;             // -1 = automatic: 6 with the static mapping, 16 with slot counters (fewer atomics)
    long waves = 2048;
    // the per-line stream kernel: automatic unless set (stream_panel_shape(), gort_stream_expand.hip)
    long stream_waves = 0;          // GORT_STREAM_WAVES: waves per panel of the per-line stream kernel (rounded like `waves`); 0 = ~2048
    int stream_steps = 0;           // GORT_STREAM_STEPS: steps per wave = panel height of the per-line stream kernel; 0 = from the stream's size
    ExpandTuning()
    {
        if (const char *v = ab_env("GORT_EXPAND_NT")) nt = atoi(v) != 0;
        if (const char *v = ab_env("GORT_EXPAND_DEPTH")) depth = atoi(v);
        if (const char *v = ab_env("GORT_EXPAND_WAVES")) waves = atol(v);
        if (const char *v = ab_env("GORT_STREAM_WAVES")) stream_waves = atol(v);
        if (const char *v = ab_env("GORT_STREAM_STEPS")) stream_steps = atoi(v);
        if (stream_steps < 0) stream_steps = 0;
        if (const char *v = ab_env("GORT_EXPAND_XCD")) xcd_mode = atoi(v);
        if (const char *v = ab_env("GORT_EXPAND_STEPS")) steps = atoi(v);
        if (xcd_mode < -1 || xcd_mode > 2) xcd_mode = -1;
        if (depth != 1 && depth != 2 && depth != 4) depth = 2;
        if (steps > 0) steps = (steps + depth - 1) / depth * depth;        // whole groups of DEPTH
        if (waves < 64) waves = 64;
        if (stream_waves != 0 && stream_waves < 64) stream_waves = 64;
    }
};

inline const ExpandTuning &tuning()
{
    static const ExpandTuning t;
    return t;
}

inline long gcd_long(long a, long b)
{
    while (b) { const long t = a % b; a = b; b = t; }
    return a;
}

// XCD mapping of a flat launch: without slot counters only the static forms are possible
inline int resolve_xcd_mode(const int *xcd_slots_dev)
{
    const int m = tuning().xcd_mode;
    if (m < 0) return xcd_slots_dev ? 2 : 1;
    return (m == 2 && !xcd_slots_dev) ? 1 : m;
}

inline FastDiv make_fast_div(unsigned d)
{
    FastDiv f;
    f.sh = 0;
    while ((1ull << f.sh) < d) ++f.sh;
    f.mul = (unsigned)((1ull << (31 + f.sh)) / d + 1);
    return f;
}

// Grid and ranges of a flat launch over `useful` logical blocks.  Mode 1: XCD x owns q w[x] blocks, q =
// ceil(useful / sum w), and gets through them in 32 q workgroup slots whatever its weight (the overshoot of at
// most sum w blocks falls off the end of the last range).
inline long plan_xcd_duty(int xcd_mode, long useful, const int *weights, XcdDuty &duty)
{
    long sumw = 0;
    duty.w8 = 0;
    for (int x = 0; x < 8; ++x) {
        int w = weights ? weights[x] : 32;
        w = w < 8 ? 8 : (w > 32 ? 32 : w);
        duty.w8 |= (unsigned long long)w << (8 * x);
        sumw += w;
    }
    duty.q = (useful + sumw - 1) / sumw;

    return xcd_mode == 1 ? 8 * 32 * duty.q : useful;
}

// chunk stride (in 1-KiB chunks) of a flat kernel: a multiple of nw/gcd(nw,CHUNK) close to the wave target
inline long flat_stride(int nw, long chunks, long target)
{
    const long unit = nw / gcd_long(nw, CHUNK);
    long mult = (target + unit / 2) / unit;
    if (mult < 1) mult = 1;
    long stride = unit * mult;
    if (stride > chunks) stride = unit * ((chunks + unit - 1) / unit);       // tiny slab: one step per wave
    return stride;
}

}  // namespace
}  // namespace gort
#endif
