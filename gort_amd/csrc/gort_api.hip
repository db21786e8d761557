// gort_api.hip -- C-ABI entry points of libgort_amd.so that touch the device.
//
// Host memory in, host memory out for the `gortt` CLI; `_dev` variants take device
// pointers so that a caller which already owns HBM buffers (bench.py, an ensemble
// driver) pays no PCIe traffic.  No CPU fallback: every entry point fails with
// GORT_ENODEVICE when HIP is unusable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <sched.h>
#include <thread>
#include <unordered_map>
#include <vector>

#include "gort_internal.h"

namespace gort {

#define GORT_HIP(call)                                                                              \
    do {                                                                                            \
        hipError_t err__ = (call);                                                                  \
        if (err__ != hipSuccess) {                                                                  \
            (void)hipGetLastError();     /* reported here: must not resurface in a later launch check */ \
            return fail(GORT_ENODEVICE, "%s: %s", #call, hipGetErrorString(err__));                 \
        }                                                                                           \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return GORT_OK;
        if (p) GORT_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        const hipError_t err = hipMalloc(&p, bytes);
        if (err != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
            return fail(err == hipErrorOutOfMemory ? GORT_ENOMEM : GORT_ENODEVICE, "hipMalloc of %zu bytes: %s", bytes,
                        hipGetErrorString(err));
        }
        cap = bytes;
        return GORT_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

}  // namespace gort

using namespace gort;

// The engine holds n_members >= 1 canopies with their spectra.  The single-canopy entry points
// (gort_engine_set_canopy / _set_spectra, stream, energy) work on member 0 of a 1-member engine.
struct gort_engine {
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev;          // start/stop pairs around the LUT expansion kernel
    size_t ev_used = 0;
    DevBuf canopy, spectra, L, coef, K, sun, nodes, angles, out, out2;
    DevBuf leaf, wl, tab_coef, tab_t12, tab_talf, tab_eof;
    DevBuf edup;                         // sun-direction table of the energy path (gort_energy.hip)
    DevBuf mbands;                       // gort_rsurf_members_stream through the line kernel: the members' StreamBand tables,
    bool mbands_current = false;         // built with the first call behind a setter (every setter ends in refresh_lambda_table)
    bool energy_dedup = true;            // GORT_ENERGY_DEDUP=0: every line evaluated (tests compare the two)
    int stream_form = 0;                 // kernel family of the last stream call: 0 narrow, 1 flat panels (gort_amd_tuning.h)
    hipEvent_t ev_stream[2] = {nullptr, nullptr};     // around the expansion of the last stream call, if asked for
    bool time_streams = false;           // gort_engine_time_streams: the two events cost a short call 6 us of its 18
    char *stage = nullptr;               // pinned staging of the setters' small uploads (stage_begin / stage_h2d)
    size_t stage_cap = 0, stage_off = 0;
    hipEvent_t ev_stage = nullptr;
    bool stage_busy = false;
    gort_pipe *hpipe = nullptr;          // pinned staging of the host-buffer entry points, created on first use
    int hpipe_nw = 0;
    unsigned hpipe_flags = 0;
    long hpipe_lines = 0;
    DevBuf xcd_slots;                    // XCD_SLOT_BYTES: per-XCD slot counters of the flat expansion kernels
    int xcd_round_robin = -1;            // probe_xcd_dispatch(): -1 not probed yet, 0 no, 1 yes
    // duty weights of the XCDs (32nds) for the static mapping; calibrated on the first LUT slab big enough
    int xcd_weights[8] = {32, 32, 32, 32, 32, 32, 32, 32};
    bool xcd_calibrated = false;
    bool xcd_weights_fixed = false;      // set by hand: never recalibrated
    int xcd_cal_class = -1;              // log2 size class of the slab the weights were measured on
    double xcd_pattern_gbs = 0.0;        // rate of the bare store pattern during the calibration pass
    double best_pattern_gbs[64] = {0};   // per size class (log2 of the doubles probed): the best rate gort_lut_alloc has seen
    // Small LUT slabs (the per-rank slabs of a multi-GPU run) are pipelined over two streams: geometry and sun
    // table of call i+1 run on `aux` into the other half of a double buffer while the expansion of call i is
    // still writing.  Only while both halves stay in the 256 MB Infinity Cache (PIPELINE_MAX_BYTES per half):
    // for the 191 MB of records of the full grid it costs 14 % instead (see grid_rows).
    hipStream_t aux = nullptr;
    hipEvent_t ev_side = nullptr;
    hipStream_t side = nullptr;          // gort_energy_members_dev beside the LUT chunks of an ensemble (gort_engine_energy_beside_grids)
    bool energy_on_side = false;
    bool time_expand = true;             // gort_engine_time_expand: two events around every LUT expansion launch
    hipEvent_t ev_tables = nullptr, ev_geom[2] = {nullptr, nullptr}, ev_expand[2] = {nullptr, nullptr};
    bool tables_recorded = false, expand_recorded[2] = {false, false};
    DevBuf gcoef[2], gsun[2];
    unsigned long grid_calls = 0;
    bool pipeline = true;                // GORT_GRID_PIPELINE=0: everything on `stream`
    int n_members = 1;
    bool have_canopy = false, have_spectra = false, have_nodes = false, have_tables = false;
    int nw = 0;
    int device = 0;                      // the HIP device the engine was created on: its LUT buffers live there
    int lut_slack_gib = 48;              // cap of the slack gort_lut_alloc keeps beside a placed buffer (GORT_LUT_SLACK_GIB at creation)
};

extern "C" int gort_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int gort_set_device(int device)
{
    GORT_HIP(hipSetDevice(device));
    return GORT_OK;
}

extern "C" int gort_get_device(void)
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) return fail(GORT_ENODEVICE, "gort_get_device: no HIP device");
    return d;
}

extern "C" void *gort_dev_malloc(size_t bytes)
{
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        fail(GORT_ENOMEM, "gort_dev_malloc: cannot allocate %zu bytes", bytes);
        return nullptr;
    }
    return p;
}

extern "C" void gort_dev_free(void *p_dev)
{
    if (p_dev) (void)hipFree(p_dev);
}

extern "C" int gort_memcpy_h2d(void *dst_dev, const void *src, size_t bytes)
{
    GORT_HIP(hipMemcpy(dst_dev, src, bytes, hipMemcpyHostToDevice));
    return GORT_OK;
}

extern "C" int gort_memcpy_d2h(void *dst, const void *src_dev, size_t bytes)
{
    GORT_HIP(hipMemcpy(dst, src_dev, bytes, hipMemcpyDeviceToHost));
    return GORT_OK;
}

// ------------------------------------------------------------ gap probabilities

extern "C" int gort_gap_probabilities_dev(gort_canopy *members_dev, int n_members, void *stream)
{
    if (!members_dev || n_members < 0) return fail(GORT_EINVAL, "gort_gap_probabilities_dev: bad argument");
    return launch_gap_probabilities(members_dev, n_members, stream);
}

// The tables of the crown geometries this process has seen (include/gort_amd.h): an ensemble filter re-submits the
// same members with new leaf/soil parameters every cycle, and `gortt` drivers loop over spectra for one stand.
namespace {
struct GapEntry {
    double geo[6];
    int q08;
    double p_n0[GORT_NTH], epgap[GORT_NTH], k_open, k_openep;
};
struct GapCache {
    std::mutex mu;
    std::unordered_multimap<uint64_t, GapEntry> map;
    std::deque<uint64_t> order;                              // insertion order: the oldest entry goes first
    long hits = 0, misses = 0;
    size_t cap = 4096;
    GapCache()
    {
        if (const char *v = getenv("GORT_GAP_CACHE")) cap = atol(v) > 0 ? (size_t)atol(v) : 0;
    }
};
GapCache &gap_cache()
{
    static GapCache c;
    return c;
}
bool same_geometry(const GapEntry &e, const gort_canopy &c)
{
    const double g[6] = {c.r, c.b, c.h1, c.h2, c.lambda, c.favd};
    return std::memcmp(e.geo, g, sizeof g) == 0 && e.q08 == (c.use_q08 ? 1 : 0);
}
}  // namespace

extern "C" void gort_gap_cache_stats(long *hits, long *misses, long *entries)
{
    GapCache &gc = gap_cache();
    std::lock_guard<std::mutex> lock(gc.mu);
    if (hits) *hits = gc.hits;
    if (misses) *misses = gc.misses;
    if (entries) *entries = (long)gc.map.size();
}

extern "C" void gort_gap_cache_clear(void)
{
    GapCache &gc = gap_cache();
    std::lock_guard<std::mutex> lock(gc.mu);
    gc.map.clear();
    gc.order.clear();
    gc.hits = gc.misses = 0;
}

extern "C" int gort_gap_probabilities(gort_canopy *members, int n_members)
{
    if (!members || n_members < 0) return fail(GORT_EINVAL, "gort_gap_probabilities: bad argument");
    if (n_members == 0) return GORT_OK;
    GapCache &gc = gap_cache();
    // 1) members whose geometry is known take their tables from the cache; the others are compacted for the device
    std::vector<int> todo;
    {
        std::lock_guard<std::mutex> lock(gc.mu);
        for (int m = 0; m < n_members; ++m) {
            bool hit = false;
            if (gc.cap) {
                auto range = gc.map.equal_range(gort_canopy_key(&members[m]));
                for (auto it = range.first; it != range.second && !hit; ++it)
                    if (same_geometry(it->second, members[m])) {
                        std::memcpy(members[m].p_n0, it->second.p_n0, sizeof it->second.p_n0);
                        std::memcpy(members[m].epgap, it->second.epgap, sizeof it->second.epgap);
                        members[m].k_open = it->second.k_open;
                        members[m].k_openep = it->second.k_openep;
                        hit = true;
                    }
            }
            if (hit) ++gc.hits;
            else todo.push_back(m);
        }
        gc.misses += (long)todo.size();
    }
    if (todo.empty()) return GORT_OK;
    for (int m : todo) {
        const int rc = gort_canopy_check_geometry(&members[m]);
        if (rc) return rc;
    }
    if (gort_device_count() <= 0) return fail(GORT_ENODEVICE, "gort_gap_probabilities: no HIP device");
    const int n = (int)todo.size();
    std::vector<gort_canopy> packed;
    gort_canopy *host = members;
    if (n != n_members) {
        packed.resize((size_t)n);
        for (int i = 0; i < n; ++i) packed[(size_t)i] = members[todo[(size_t)i]];
        host = packed.data();
    }
    gort_canopy *dev = nullptr;
    const size_t bytes = sizeof(gort_canopy) * (size_t)n;
    GORT_HIP(hipMalloc((void **)&dev, bytes));
    int rc = GORT_OK;
    hipError_t e = hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = launch_gap_probabilities(dev, n, nullptr);
        if (rc == GORT_OK) e = hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost);
    }
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(GORT_ENODEVICE, "gort_gap_probabilities: %s", hipGetErrorString(e));
    if (rc) return rc;
    // 2) hand the results back and remember them
    std::lock_guard<std::mutex> lock(gc.mu);
    for (int i = 0; i < n; ++i) {
        gort_canopy &c = members[todo[(size_t)i]];
        if (host != members) c = host[i];
        if (!gc.cap) continue;
        while (gc.map.size() >= gc.cap && !gc.order.empty()) {
            auto it = gc.map.find(gc.order.front());
            if (it != gc.map.end()) gc.map.erase(it);
            gc.order.pop_front();
        }
        GapEntry en;
        const double g[6] = {c.r, c.b, c.h1, c.h2, c.lambda, c.favd};
        std::memcpy(en.geo, g, sizeof g);
        en.q08 = c.use_q08 ? 1 : 0;
        std::memcpy(en.p_n0, c.p_n0, sizeof en.p_n0);
        std::memcpy(en.epgap, c.epgap, sizeof en.epgap);
        en.k_open = c.k_open;
        en.k_openep = c.k_openep;
        const uint64_t key = gort_canopy_key(&c);
        gc.map.emplace(key, en);
        gc.order.push_back(key);
    }
    return GORT_OK;
}

// ----------------------------------------------------------------------- engine

extern "C" int gort_engine_create(gort_engine **out)
{
    if (!out) return fail(GORT_EINVAL, "gort_engine_create: null out");
    *out = nullptr;
    if (gort_device_count() <= 0) return fail(GORT_ENODEVICE, "gort_engine_create: no HIP device");
    gort_engine *e = new (std::nothrow) gort_engine();
    if (!e) return fail(GORT_ENOMEM, "gort_engine_create: out of memory");
    bool ok = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&e->ev_tables, hipEventDisableTiming) == hipSuccess;
    for (int b = 0; b < 2 && ok; ++b)
        ok = hipEventCreateWithFlags(&e->ev_geom[b], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&e->ev_expand[b], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        gort_engine_destroy(e);
        return fail(GORT_ENODEVICE, "gort_engine_create: cannot create streams/events");
    }
    if (hipGetDevice(&e->device) != hipSuccess) { (void)hipGetLastError(); e->device = 0; }
    if (const char *v = getenv("GORT_LUT_SLACK_GIB")) {
        const long cap = atol(v);
        e->lut_slack_gib = cap < 0 ? 0 : (cap > 63 ? 63 : (int)cap);
    }
    // A/B switches, read once per engine in the measuring build only (gort_internal.h); a host sets the duty weights through
    // gort_engine_set_xcd_weights (include/gort_amd_tuning.h)
    if (const char *v = ab_env("GORT_GRID_PIPELINE")) e->pipeline = atoi(v) != 0;
    if (const char *v = ab_env("GORT_ENERGY_DEDUP")) e->energy_dedup = atoi(v) != 0;
    if (const char *v = ab_env("GORT_XCD_CALIBRATE")) e->xcd_calibrated = e->xcd_weights_fixed = atoi(v) == 0;   // 0: equal weights
    if (const char *v = ab_env("GORT_XCD_WEIGHTS")) {                                         // "32,25,32,25,..."
        int w[8];
        if (sscanf(v, "%d,%d,%d,%d,%d,%d,%d,%d", w, w + 1, w + 2, w + 3, w + 4, w + 5, w + 6, w + 7) == 8) {
            // the range gort_engine_set_xcd_weights accepts: what gort_engine_xcd_weights reports is what is used
            for (int x = 0; x < 8; ++x) e->xcd_weights[x] = w[x] < 8 ? 8 : (w[x] > 32 ? 32 : w[x]);
            e->xcd_calibrated = e->xcd_weights_fixed = true;
        }
    }
    *out = e;
    return GORT_OK;
}

extern "C" void gort_engine_destroy(gort_engine *e)
{
    if (!e) return;
    if (e->hpipe) gort_pipe_destroy(e->hpipe);
    e->hpipe = nullptr;
    if (e->aux) (void)hipStreamSynchronize(e->aux);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (DevBuf *b : {&e->gcoef[0], &e->gcoef[1], &e->gsun[0], &e->gsun[1], &e->mbands}) b->release();
    for (hipEvent_t ev : {e->ev_tables, e->ev_geom[0], e->ev_geom[1], e->ev_expand[0], e->ev_expand[1]})
        if (ev) (void)hipEventDestroy(ev);
    if (e->aux) (void)hipStreamDestroy(e->aux);
    if (e->side) { (void)hipStreamSynchronize(e->side); (void)hipStreamDestroy(e->side); }
    if (e->ev_side) (void)hipEventDestroy(e->ev_side);
    for (DevBuf *b : {&e->canopy, &e->spectra, &e->L, &e->coef, &e->K, &e->sun, &e->nodes, &e->angles, &e->out,
                      &e->out2, &e->leaf, &e->wl, &e->tab_coef, &e->tab_t12, &e->tab_talf, &e->tab_eof, &e->xcd_slots, &e->edup})
        b->release();
    for (hipEvent_t ev : e->ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->ev_stream) if (ev) (void)hipEventDestroy(ev);
    if (e->ev_stage) (void)hipEventDestroy(e->ev_stage);
    if (e->stage) (void)hipHostFree(e->stage);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

// Slot counters for one flat expansion launch, zeroed on the stream - or nullptr where workgroup dispatch is
// round-robin over the XCDs, probed once per engine (the static mapping is then exact and free of atomics).
static int xcd_slots_for_launch(gort_engine *e, int **slots)
{
    *slots = nullptr;
    if (e->xcd_round_robin < 0) {
        int rr = 0;
        const int rc = probe_xcd_dispatch(e->stream, &rr);
        if (rc) return rc;
        e->xcd_round_robin = rr;
    }
    if (!expand_wants_xcd_slots(e->xcd_round_robin == 1)) return GORT_OK;
    const int rc = e->xcd_slots.reserve(XCD_SLOT_BYTES);
    if (rc) return rc;
    GORT_HIP(hipMemsetAsync(e->xcd_slots.p, 0, XCD_SLOT_BYTES, e->stream));
    *slots = e->xcd_slots.as<int>();
    return GORT_OK;
}

extern "C" int gort_engine_xcd_mapping(gort_engine *e)
{
    if (!e) return fail(GORT_EINVAL, "gort_engine_xcd_mapping: null engine");
    int *slots = nullptr;
    const int rc = xcd_slots_for_launch(e, &slots);
    if (rc) return rc;                       // GORT_E* codes are negative already
    return slots ? 2 : 1;
}

extern "C" int gort_engine_xcd_weights(const gort_engine *e, int weights[8])
{
    if (!e || !weights) return fail(GORT_EINVAL, "gort_engine_xcd_weights: null argument");
    for (int x = 0; x < 8; ++x) weights[x] = e->xcd_weights[x];
    return e->xcd_calibrated ? 1 : 0;
}

extern "C" int gort_selftest_index_math(void) { return selftest_index_math(); }

extern "C" int gort_engine_set_lut_slack_gib(gort_engine *e, int gib)
{
    if (!e || gib < 0 || gib > 63) return fail(GORT_EINVAL, "gort_engine_set_lut_slack_gib: bad argument");
    e->lut_slack_gib = gib;
    return GORT_OK;
}

extern "C" double gort_engine_store_pattern_gbs(const gort_engine *e) { return e ? e->xcd_pattern_gbs : 0.0; }

extern "C" int gort_engine_set_xcd_weights(gort_engine *e, const int weights[8])
{
    if (!e) return fail(GORT_EINVAL, "gort_engine_set_xcd_weights: null engine");
    if (!weights) {                          // back to automatic: calibrate on the next big LUT slab
        for (int x = 0; x < 8; ++x) e->xcd_weights[x] = 32;
        e->xcd_calibrated = e->xcd_weights_fixed = false;
        return GORT_OK;
    }
    for (int x = 0; x < 8; ++x)
        if (weights[x] < 8 || weights[x] > 32) return fail(GORT_ERANGE, "gort_engine_set_xcd_weights: weight %d outside 8..32", weights[x]);
    for (int x = 0; x < 8; ++x) e->xcd_weights[x] = weights[x];
    e->xcd_calibrated = e->xcd_weights_fixed = true;
    return GORT_OK;
}

extern "C" void *gort_engine_stream(gort_engine *e) { return e ? (void *)e->stream : nullptr; }
extern "C" int gort_engine_nw(const gort_engine *e) { return e ? e->nw : 0; }
extern "C" int gort_engine_n_members(const gort_engine *e) { return e ? e->n_members : 0; }

extern "C" int gort_engine_synchronize(gort_engine *e)
{
    if (!e) return fail(GORT_EINVAL, "gort_engine_synchronize: null engine");
    GORT_HIP(hipStreamSynchronize(e->aux));
    if (e->side) GORT_HIP(hipStreamSynchronize(e->side));
    GORT_HIP(hipStreamSynchronize(e->stream));
    return GORT_OK;
}

// [r6] The albedo / fAPAR table of an ensemble is fp64-issue and latency bound (1000 members: 2 ms), its hemisphere LUTs are
// HBM-write bound (75 ms): asked for first and queued on a stream of its own, the table is evaluated under the LUT chunks
// instead of behind them.  The caller owns the ordering of what it hands in (angles, output) against its own streams, as ever;
// gort_engine_synchronize waits for this stream too.
extern "C" int gort_engine_energy_beside_grids(gort_engine *e, int on)
{
    if (!e) return fail(GORT_EINVAL, "gort_engine_energy_beside_grids: null engine");
    if (on && !e->side) GORT_HIP(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
    e->energy_on_side = on != 0;
    return GORT_OK;
}

extern "C" int gort_engine_time_expand(gort_engine *e, int on)
{
    if (!e) return fail(GORT_EINVAL, "gort_engine_time_expand: null engine");
    e->time_expand = on != 0;
    return GORT_OK;
}

// band-only tables of every member, L[n][11][nw]; every setter ends here, so this is also where the aux
// stream learns that canopies / spectra / tables changed
static int refresh_lambda_table(gort_engine *e)
{
    int rc = GORT_OK;
    if (e->have_canopy && e->have_spectra) {
        rc = e->L.reserve(sizeof(double) * lambda_table_doubles(e->nw, e->n_members));
        if (rc) return rc;
        rc = launch_lambda_table(e->canopy.as<gort_canopy>(), e->n_members, e->nw, e->spectra.as<double>(),
                                 e->L.as<double>(), e->stream);
    }
    e->mbands_current = false;
    GORT_HIP(hipEventRecord(e->ev_tables, e->stream));
    e->tables_recorded = true;
    return rc;
}

// Small host -> device uploads of the setters go through a pinned staging buffer and the engine's stream: ordered
// behind whatever still reads the previous contents, no host wait, and none of the stalls of pageable hipMemcpy
// (15-25 ms now and then for the 3 MB of a 1000-member ensemble).  stage_begin() once per setter call (waits only
// if the previous call's uploads are still in flight), stage_h2d() per piece, stage_end() at the end.
static int stage_begin(gort_engine *e, size_t total_bytes)
{
    if (!e->ev_stage) GORT_HIP(hipEventCreateWithFlags(&e->ev_stage, hipEventDisableTiming));
    if (e->stage_busy) {
        GORT_HIP(hipEventSynchronize(e->ev_stage));
        e->stage_busy = false;
    }
    total_bytes += 4096;
    if (total_bytes > e->stage_cap) {
        if (e->stage) GORT_HIP(hipHostFree(e->stage));
        e->stage = nullptr;
        e->stage_cap = 0;
        if (hipHostMalloc((void **)&e->stage, total_bytes, hipHostMallocDefault) != hipSuccess)
            return fail(GORT_ENOMEM, "cannot pin %zu bytes of staging memory", total_bytes);
        e->stage_cap = total_bytes;
    }
    e->stage_off = 0;
    return GORT_OK;
}

static int stage_h2d(gort_engine *e, void *dst_dev, const void *src, size_t bytes)
{
    if (e->stage_off + bytes > e->stage_cap) return fail(GORT_EINVAL, "staging buffer overrun");
    char *h = e->stage + e->stage_off;
    std::memcpy(h, src, bytes);
    e->stage_off += (bytes + 255) & ~(size_t)255;
    GORT_HIP(hipMemcpyAsync(dst_dev, h, bytes, hipMemcpyHostToDevice, e->stream));
    return GORT_OK;
}

static int stage_end(gort_engine *e)
{
    GORT_HIP(hipEventRecord(e->ev_stage, e->stream));
    e->stage_busy = true;
    return GORT_OK;
}

// caller: stage_begin() with room for n canopies, stage_end() afterwards
static int upload_canopies(gort_engine *e, const gort_canopy *members, int n, int compute_gaps)
{
    int rc;
    if (compute_gaps)
        for (int m = 0; m < n; ++m)
            if ((rc = gort_canopy_check_geometry(&members[m]))) return rc;       // before anything is uploaded
    if ((rc = e->canopy.reserve(sizeof(gort_canopy) * (size_t)n))) return rc;
    if ((rc = stage_h2d(e, e->canopy.p, members, sizeof(gort_canopy) * (size_t)n))) return rc;
    if (compute_gaps && (rc = launch_gap_probabilities(e->canopy.as<gort_canopy>(), n, e->stream))) return rc;
    if (n != e->n_members) e->have_spectra = false;    // spectra are per member
    e->n_members = n;
    e->have_canopy = true;
    return GORT_OK;
}

extern "C" int gort_engine_set_canopy(gort_engine *e, const gort_canopy *c)
{
    if (!e || !c) return fail(GORT_EINVAL, "gort_engine_set_canopy: bad argument");
    int rc = stage_begin(e, sizeof(gort_canopy));
    if (rc) return rc;
    if ((rc = upload_canopies(e, c, 1, 0))) return rc;
    if ((rc = stage_end(e))) return rc;
    return refresh_lambda_table(e);
}

extern "C" int gort_engine_set_spectra(gort_engine *e, int nw, const double *rsoil, const double *rleaf,
                                       const double *tleaf)
{
    if (!e || nw <= 0 || !rsoil || !rleaf || !tleaf) return fail(GORT_EINVAL, "gort_engine_set_spectra: bad argument");
    if (e->n_members != 1)
        return fail(GORT_EINVAL, "gort_engine_set_spectra: engine holds %d members; use gort_engine_set_members", e->n_members);
    int rc = e->spectra.reserve(sizeof(double) * 3 * (size_t)nw);
    if (rc) return rc;
    double *sp = e->spectra.as<double>();
    const size_t b = sizeof(double) * (size_t)nw;
    if ((rc = stage_begin(e, 3 * (b + 256)))) return rc;
    if ((rc = stage_h2d(e, sp, rsoil, b)) || (rc = stage_h2d(e, sp + nw, rleaf, b)) || (rc = stage_h2d(e, sp + 2 * nw, tleaf, b)))
        return rc;
    if ((rc = stage_end(e))) return rc;
    e->nw = nw;
    e->have_spectra = true;
    return refresh_lambda_table(e);
}

// ---- ensembles (BASELINE config 5) ----

extern "C" int gort_engine_set_members(gort_engine *e, const gort_canopy *members, int n_members, int compute_gaps,
                                       int nw, const double *spectra)
{
    if (!e || !members || n_members <= 0 || n_members > 65535 || nw <= 0 || !spectra)
        return fail(GORT_EINVAL, "gort_engine_set_members: bad argument");
    int rc = stage_begin(e, sizeof(gort_canopy) * (size_t)n_members);
    if (rc) return rc;
    if ((rc = upload_canopies(e, members, n_members, compute_gaps))) return rc;
    if ((rc = stage_end(e))) return rc;
    const size_t bytes = sizeof(double) * 3 * (size_t)nw * (size_t)n_members;
    if ((rc = e->spectra.reserve(bytes))) return rc;
    GORT_HIP(hipStreamSynchronize(e->stream));         // large and pageable: a plain copy, behind everything queued
    GORT_HIP(hipMemcpy(e->spectra.p, spectra, bytes, hipMemcpyHostToDevice));
    e->nw = nw;
    e->have_spectra = true;
    return refresh_lambda_table(e);
}

static int ensure_spectral_tables(gort_engine *e)
{
    if (e->have_tables) return GORT_OK;
    const double *t12, *talf;
    interface_transmissivity_tables(&t12, &talf);
    int rc;
    if ((rc = e->tab_coef.reserve(sizeof(float) * 7 * GORT_NBANDS))) return rc;
    if ((rc = e->tab_t12.reserve(sizeof(double) * GORT_NBANDS))) return rc;
    if ((rc = e->tab_talf.reserve(sizeof(double) * GORT_NBANDS))) return rc;
    if ((rc = e->tab_eof.reserve(sizeof(double) * 4 * 421))) return rc;
    GORT_HIP(hipMemcpy(e->tab_coef.p, prospect_coeff_table(), sizeof(float) * 7 * GORT_NBANDS, hipMemcpyHostToDevice));
    GORT_HIP(hipMemcpy(e->tab_t12.p, t12, sizeof(double) * GORT_NBANDS, hipMemcpyHostToDevice));
    GORT_HIP(hipMemcpy(e->tab_talf.p, talf, sizeof(double) * GORT_NBANDS, hipMemcpyHostToDevice));
    GORT_HIP(hipMemcpy(e->tab_eof.p, price_eof_table(), sizeof(double) * 4 * 421, hipMemcpyHostToDevice));
    e->have_tables = true;
    return GORT_OK;
}

extern "C" int gort_engine_set_members_leaf(gort_engine *e, const gort_canopy *members, const gort_leaf_soil *leaf,
                                            int n_members, int compute_gaps, const double *wl_nm, int nw)
{
    if (!e || !members || !leaf || n_members <= 0 || n_members > 65535 || nw <= 0 || !wl_nm)
        return fail(GORT_EINVAL, "gort_engine_set_members_leaf: bad argument");
    for (int i = 0; i < nw; ++i)
        if (!(wl_nm[i] >= 400 && wl_nm[i] <= 2500))
            return fail(GORT_ERANGE, "gortt_price_soil: wavlength out of range (400-2500)");
    int rc = stage_begin(e, (sizeof(gort_canopy) + sizeof(gort_leaf_soil)) * (size_t)n_members + sizeof(double) * (size_t)nw + 1024);
    if (rc) return rc;
    if ((rc = upload_canopies(e, members, n_members, compute_gaps))) return rc;
    if ((rc = ensure_spectral_tables(e))) return rc;
    if ((rc = e->leaf.reserve(sizeof(gort_leaf_soil) * (size_t)n_members))) return rc;
    if ((rc = e->wl.reserve(sizeof(double) * (size_t)nw))) return rc;
    if ((rc = e->spectra.reserve(sizeof(double) * 3 * (size_t)nw * (size_t)n_members))) return rc;
    if ((rc = stage_h2d(e, e->leaf.p, leaf, sizeof(gort_leaf_soil) * (size_t)n_members))) return rc;
    if ((rc = stage_h2d(e, e->wl.p, wl_nm, sizeof(double) * (size_t)nw))) return rc;
    if ((rc = stage_end(e))) return rc;
    rc = launch_member_spectra(e->leaf.as<gort_leaf_soil>(), n_members, nw, e->wl.as<double>(),
                               e->tab_coef.as<float>(), e->tab_t12.as<double>(), e->tab_talf.as<double>(),
                               e->tab_eof.as<double>(), e->spectra.as<double>(), e->stream);
    if (rc) return rc;
    e->nw = nw;
    e->have_spectra = true;
    return refresh_lambda_table(e);
}

// capacity for an ensemble of n_members x nw bands: every buffer the member setters and the band tables need, the pinned
// staging of their uploads and the spectral tables - so that a setter call costs its copies and kernels (~1.5 ms for 1000
// members) and not the process's first 240 MB of hipMalloc and hipHostMalloc (15 - 30 ms, by the box)
extern "C" int gort_engine_reserve_members(gort_engine *e, int n_members, int nw)
{
    if (!e || n_members <= 0 || n_members > 65535 || nw <= 0) return fail(GORT_EINVAL, "gort_engine_reserve_members: bad argument");
    int rc = stage_begin(e, (sizeof(gort_canopy) + sizeof(gort_leaf_soil)) * (size_t)n_members + sizeof(double) * (size_t)nw + 1024);
    if (rc) return rc;
    if ((rc = ensure_spectral_tables(e))) return rc;
    // growing a buffer discards what it held (DevBuf::reserve frees and allocates): an engine that was configured before is
    // "not ready" again until its canopies and spectra are set once more - never a stream of garbage
    void *const held[3] = {e->canopy.p, e->spectra.p, e->L.p};
    if ((rc = e->canopy.reserve(sizeof(gort_canopy) * (size_t)n_members)) == GORT_OK &&
        (rc = e->leaf.reserve(sizeof(gort_leaf_soil) * (size_t)n_members)) == GORT_OK &&
        (rc = e->wl.reserve(sizeof(double) * (size_t)nw)) == GORT_OK &&
        (rc = e->spectra.reserve(sizeof(double) * 3 * (size_t)nw * (size_t)n_members)) == GORT_OK)
        rc = e->L.reserve(sizeof(double) * lambda_table_doubles(nw, n_members));
    if (held[0] != e->canopy.p || held[2] != e->L.p) e->have_canopy = false;
    if (held[1] != e->spectra.p || held[2] != e->L.p) e->have_spectra = false;
    return rc;
}

extern "C" int gort_engine_get_member(gort_engine *e, int member, gort_canopy *canopy, double *rsoil, double *rleaf,
                                      double *tleaf)
{
    if (!e || member < 0 || member >= e->n_members) return fail(GORT_EINVAL, "gort_engine_get_member: bad member");
    GORT_HIP(hipStreamSynchronize(e->stream));
    if (canopy) {
        if (!e->have_canopy) return fail(GORT_EINVAL, "gort_engine_get_member: no canopy set");
        GORT_HIP(hipMemcpy(canopy, e->canopy.as<gort_canopy>() + member, sizeof(gort_canopy), hipMemcpyDeviceToHost));
    }
    if (rsoil || rleaf || tleaf) {
        if (!e->have_spectra) return fail(GORT_EINVAL, "gort_engine_get_member: no spectra set");
        const double *sp = e->spectra.as<double>() + (size_t)member * 3 * e->nw;
        const size_t b = sizeof(double) * (size_t)e->nw;
        if (rsoil) GORT_HIP(hipMemcpy(rsoil, sp, b, hipMemcpyDeviceToHost));
        if (rleaf) GORT_HIP(hipMemcpy(rleaf, sp + e->nw, b, hipMemcpyDeviceToHost));
        if (tleaf) GORT_HIP(hipMemcpy(tleaf, sp + 2 * e->nw, b, hipMemcpyDeviceToHost));
    }
    return GORT_OK;
}

static int require_ready(const gort_engine *e, const char *who)
{
    if (!e) return fail(GORT_EINVAL, "%s: null engine", who);
    if (!e->have_canopy) return fail(GORT_EINVAL, "%s: no canopy set", who);
    if (!e->have_spectra) return fail(GORT_EINVAL, "%s: no spectra set", who);
    return GORT_OK;
}

// ----------------------------------------------------------------- BRDF stream

extern "C" int gort_rsurf_stream_dev(gort_engine *e, const double *angles_dev, long nA, double *rsurf_dev,
                                     double *scomp_dev, double *K_dev)
{
    int rc;
    // The viewed proportions alone (K_dev, no reflectance): they depend on the canopy and the angles only, and the
    // reference prints them for a header without wavelengths too (`N 0`: gortt_rsurf computes Kc, Kg, Kt, Kz in front
    // of its wavelength loop, gortt.c:424-449) - no spectra needed, one launch of the geometry kernel
    if (e && e->have_canopy && !rsurf_dev && !scomp_dev && K_dev) {
        if (nA < 0 || (nA > 0 && !angles_dev)) return fail(GORT_EINVAL, "gort_rsurf_stream_dev: bad argument");
        if (nA == 0) return GORT_OK;
        return launch_geometry_stream_fused(e->canopy.as<gort_canopy>(), 1, nullptr, 0, angles_dev, nA, nullptr, K_dev, e->stream);
    }
    if ((rc = require_ready(e, "gort_rsurf_stream_dev"))) return rc;
    if (nA < 0 || (nA > 0 && (!angles_dev || !rsurf_dev))) return fail(GORT_EINVAL, "gort_rsurf_stream_dev: bad argument");
    if (nA == 0) return GORT_OK;
    const bool timed = e->time_streams;
    for (int i = 0; timed && i < 2; ++i)
        if (!e->ev_stream[i]) GORT_HIP(hipEventCreate(&e->ev_stream[i]));
    // 17 ... ~250 bands (all the reference's command line can read): one kernel from the angle line to its row
    if (stream_takes_lines_kernel(e->nw, nA, scomp_dev != nullptr)) {
        if (timed) GORT_HIP(hipEventRecord(e->ev_stream[0], e->stream));
        rc = launch_stream_lines(e->canopy.as<gort_canopy>(), 1, stream_band_table(e->L.as<double>(), e->nw, e->n_members), e->nw,
                                 angles_dev, nA, rsurf_dev, K_dev, e->stream);
        if (timed) GORT_HIP(hipEventRecord(e->ev_stream[1], e->stream));
        e->stream_form = 2;
        return rc;
    }
    // line records with one pad record in front and a tail pad (the aligned flat expansion prefetches)
    const long tail = expand_stream_tail_pad_records(e->nw, nA);
    const size_t coef_bytes = sizeof(double) * GORT_COEF_STRIDE * (size_t)(nA + 1 + tail);
    const bool fresh = coef_bytes > e->coef.cap;
    if ((rc = e->coef.reserve(coef_bytes))) return rc;
    if (fresh) GORT_HIP(hipMemsetAsync(e->coef.p, 0, coef_bytes, e->stream));      // pads: readable, contents don't-care (padded lanes are never stored)
    double *coef = e->coef.as<double>() + GORT_COEF_STRIDE;
    int *xcd_slots = nullptr;
    if ((rc = xcd_slots_for_launch(e, &xcd_slots))) return rc;
    const gort_canopy *c = e->canopy.as<gort_canopy>();          // member 0
    const bool large = stream_is_large(e->nw, nA, scomp_dev != nullptr);
    if (stream_fuses(e->nw, scomp_dev != nullptr)) {
        if (timed) GORT_HIP(hipEventRecord(e->ev_stream[0], e->stream));
        rc = launch_geometry_stream_fused(c, 1, e->L.as<double>(), e->nw, angles_dev, nA, rsurf_dev, K_dev, e->stream);
        if (timed) GORT_HIP(hipEventRecord(e->ev_stream[1], e->stream));
        e->stream_form = 0;
        return rc;
    }
    if ((rc = launch_geometry_stream(c, 1, angles_dev, nA, coef, K_dev, large ? 1 : 0, e->stream, K_dev != nullptr || scomp_dev != nullptr))) return rc;
    if (timed) GORT_HIP(hipEventRecord(e->ev_stream[0], e->stream));
    rc = launch_expand_stream(c, e->L.as<double>(), stream_band_table(e->L.as<double>(), e->nw, e->n_members), e->nw, coef, nA, rsurf_dev,
                              scomp_dev, xcd_slots, e->stream, false);
    if (timed) GORT_HIP(hipEventRecord(e->ev_stream[1], e->stream));
    e->stream_form = large ? 1 : 0;
    return rc;
}

// ---- host buffers in, host buffers out ----
// Pageable memory cannot be the target of a DMA: the runtime stages such copies itself at ~10 GB/s.  Results
// therefore go either straight into the caller's buffer when that is pinned (gort_host_malloc, hipHostMalloc,
// hipHostRegister: one copy at the PCIe rate) or through the pinned slots of an internal gort_pipe, chunk by
// chunk with `depth` chunks in flight, and are copied out of the slots by several host threads while the next
// chunks are still on their way.

static bool is_pinned_host(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();            // unregistered host memory is reported as an error: clear it
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

static void parallel_copy(void *dst, const void *src, size_t bytes)
{
    constexpr size_t PER_THREAD = 8u << 20;
    unsigned hw = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) hw = (unsigned)CPU_COUNT(&set);
    unsigned nt = (unsigned)(bytes / PER_THREAD);
    if (nt > hw) nt = hw;
    if (nt > 8) nt = 8;
    if (nt <= 1) { std::memcpy(dst, src, bytes); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t a = bytes * t / nt / 64 * 64, b = t + 1 == nt ? bytes : bytes * (t + 1) / nt / 64 * 64;
        pool.emplace_back([=] { std::memcpy((char *)dst + a, (const char *)src + a, b - a); });
    }
    for (auto &th : pool) th.join();
}

// lines per slot of the internal pipe: ~32 MiB of rsurf, a few chunks per call
static long host_chunk_lines(int nw, bool scomp, bool energy)
{
    const size_t per_line = sizeof(double) * (size_t)(nw > 0 ? nw : 1) * (1 + (scomp ? 4 : 0) + (energy ? 3 : 0));
    long n = (long)((32u << 20) / per_line);
    return n < 256 ? 256 : (n > 65536 ? 65536 : n);
}

static int host_pipe(gort_engine *e, unsigned flags, gort_pipe **out)
{
    const long lines = host_chunk_lines(e->nw, flags & GORT_PIPE_SCOMP, flags & GORT_PIPE_ENERGY);
    if (e->hpipe && (e->hpipe_nw != e->nw || e->hpipe_flags != flags)) {
        gort_pipe_destroy(e->hpipe);
        e->hpipe = nullptr;
    }
    if (!e->hpipe) {
        const int rc = gort_pipe_create(e, lines, 3, flags, &e->hpipe);
        if (rc) return rc;
        e->hpipe_nw = e->nw;
        e->hpipe_flags = flags;
        e->hpipe_lines = lines;
    }
    *out = e->hpipe;
    return GORT_OK;
}

// nA lines through the internal pipe; `energy` selects what is collected
static int staged_stream(gort_engine *e, const double *angles, long nA, double *rsurf, double *scomp, double *K,
                         double *energy)
{
    gort_pipe *p = nullptr;
    const unsigned flags = (scomp ? GORT_PIPE_SCOMP : 0u) | (energy ? (rsurf ? GORT_PIPE_ENERGY : GORT_PIPE_ENERGY_ONLY) : 0u);
    int rc = host_pipe(e, flags, &p);
    if (rc) return rc;
    const long L = e->hpipe_lines;
    const size_t nw = (size_t)e->nw, D = sizeof(double);
    long sent = 0, got = 0;
    int in_flight = 0;
    auto collect = [&]() -> int {
        gort_pipe_chunk c;
        int r = gort_pipe_wait(p, &c);
        if (r == GORT_OK) {
            const size_t o = (size_t)got;
            if (rsurf) parallel_copy(rsurf + o * nw, c.rsurf, D * (size_t)c.n * nw);
            if (scomp) parallel_copy(scomp + o * nw * 4, c.scomp, D * 4 * (size_t)c.n * nw);
            if (K) std::memcpy(K + o * 4, c.K, D * 4 * (size_t)c.n);
            if (energy) parallel_copy(energy + o * nw * 3, c.energy, D * 3 * (size_t)c.n * nw);
            got += c.n;
        }
        const int r2 = gort_pipe_release(p);
        --in_flight;
        return r ? r : r2;
    };
    while (sent < nA) {
        if (in_flight == 3 && (rc = collect())) break;
        double *slot = nullptr;
        if ((rc = gort_pipe_acquire(p, &slot))) break;
        const long n = nA - sent < L ? nA - sent : L;
        std::memcpy(slot, angles + 4 * sent, D * 4 * (size_t)n);
        rc = gort_pipe_submit(p, n);
        ++in_flight;
        sent += n;
        if (rc) break;
    }
    while (in_flight > 0) {
        const int r = collect();
        if (rc == GORT_OK) rc = r;
    }
    return rc;
}

extern "C" int gort_rsurf_stream(gort_engine *e, const double *angles, long nA, double *rsurf, double *scomp,
                                 double *K)
{
    int rc = require_ready(e, "gort_rsurf_stream");
    if (rc) return rc;
    if (nA < 0 || (nA > 0 && (!angles || !rsurf))) return fail(GORT_EINVAL, "gort_rsurf_stream: bad argument");
    if (nA == 0) return GORT_OK;
    const size_t nw = (size_t)e->nw, n = (size_t)nA;
    const size_t out_bytes = sizeof(double) * n * nw * (scomp ? 5 : 1);
    const bool direct = out_bytes < (4u << 20) || (is_pinned_host(rsurf) && (!scomp || is_pinned_host(scomp)));
    if (!direct) return staged_stream(e, angles, nA, rsurf, scomp, K, nullptr);
    // small calls, and callers with pinned buffers: one copy in, the kernels, one copy out
    if ((rc = e->angles.reserve(sizeof(double) * 4 * n))) return rc;
    if ((rc = e->out.reserve(sizeof(double) * n * nw))) return rc;
    if (scomp && (rc = e->out2.reserve(sizeof(double) * 4 * n * nw))) return rc;
    if (K && (rc = e->K.reserve(sizeof(double) * 4 * n))) return rc;
    GORT_HIP(hipMemcpyAsync(e->angles.p, angles, sizeof(double) * 4 * n, hipMemcpyHostToDevice, e->stream));
    rc = gort_rsurf_stream_dev(e, e->angles.as<double>(), nA, e->out.as<double>(),
                               scomp ? e->out2.as<double>() : nullptr, K ? e->K.as<double>() : nullptr);
    if (rc) return rc;
    GORT_HIP(hipMemcpyAsync(rsurf, e->out.p, sizeof(double) * n * nw, hipMemcpyDeviceToHost, e->stream));
    if (scomp) GORT_HIP(hipMemcpyAsync(scomp, e->out2.p, sizeof(double) * 4 * n * nw, hipMemcpyDeviceToHost, e->stream));
    if (K) GORT_HIP(hipMemcpyAsync(K, e->K.p, sizeof(double) * 4 * n, hipMemcpyDeviceToHost, e->stream));
    GORT_HIP(hipStreamSynchronize(e->stream));
    return GORT_OK;
}

extern "C" int gort_engine_stream_form(gort_engine *e)
{
    if (!e) return fail(GORT_EINVAL, "gort_engine_stream_form: null engine");
    return e->stream_form;
}

extern "C" int gort_engine_time_streams(gort_engine *e, int on)
{
    if (!e) return fail(GORT_EINVAL, "gort_engine_time_streams: null engine");
    e->time_streams = on != 0;
    return GORT_OK;
}

extern "C" double gort_engine_last_stream_ms(gort_engine *e)
{
    if (!e || !e->time_streams || !e->ev_stream[1]) return -1.0;
    float ms = 0.f;
    if (hipEventSynchronize(e->ev_stream[1]) != hipSuccess ||
        hipEventElapsedTime(&ms, e->ev_stream[0], e->ev_stream[1]) != hipSuccess)
        return -1.0;
    return ms;
}

// The same angle lines for ensemble members [member_begin, member_end): what an ensemble filter compares with
// its observations (a few sun/view geometries x a few bands per member, every member in one launch pair).
extern "C" int gort_rsurf_members_stream_dev(gort_engine *e, const double *angles_dev, long nA, int member_begin,
                                             int member_end, double *rsurf_dev)
{
    int rc = require_ready(e, "gort_rsurf_members_stream_dev");
    if (rc) return rc;
    if (member_begin < 0 || member_end > e->n_members || member_begin > member_end)
        return fail(GORT_EINVAL, "gort_rsurf_members_stream_dev: members [%d,%d) outside [0,%d)", member_begin,
                    member_end, e->n_members);
    if (nA < 0 || (nA > 0 && member_end > member_begin && (!angles_dev || !rsurf_dev)))
        return fail(GORT_EINVAL, "gort_rsurf_members_stream_dev: bad argument");
    const int nm = member_end - member_begin;
    if (nA == 0 || nm == 0) return GORT_OK;
    // from 17 bands: the line kernel, the member in blockIdx.y - geometry and samples in one launch, rows as whole cache lines: a
    // thousand members x 2000 lines x 100 bands in 0.58 ms where records + one thread per sample took 1.9 (a cliff at 17 bands:
    // 16 bands, the fused kernel, 0.19 ms; 17 bands 0.52).  Any band count: the flat-panel kernel has no member dimension
    if (members_stream_takes_lines_kernel(e->nw, nA, nm)) {
        // the members' StreamBand tables (96 B per member and band: 200 MB for 1000 x 2101): of the whole ensemble, once per setter
        // call - a filter asks for its observations many times per cycle, and the 95 us of this kernel were 2 % of every call
        const size_t per_member = STREAM_BAND_TABLE_DOUBLES * (size_t)e->nw;
        if (!e->mbands_current) {
            if ((rc = e->mbands.reserve(sizeof(double) * per_member * (size_t)e->n_members))) return rc;
            if ((rc = launch_member_stream_bands(e->L.as<double>(), e->n_members, e->nw, e->mbands.as<double>(), e->stream))) return rc;
            e->mbands_current = true;
        }
        return launch_stream_lines(e->canopy.as<gort_canopy>() + member_begin, nm, e->mbands.as<double>() + per_member * (size_t)member_begin,
                                   e->nw, angles_dev, nA, rsurf_dev, nullptr, e->stream);
    }
    if ((rc = e->coef.reserve(sizeof(double) * GORT_COEF_STRIDE * (size_t)nA * (size_t)nm))) return rc;
    return launch_members_stream(e->canopy.as<gort_canopy>() + member_begin, nm,
                                 e->L.as<double>() + (size_t)member_begin * L_NSLOT * e->nw, e->nw, angles_dev, nA,
                                 e->coef.as<double>(), rsurf_dev, e->stream);
}

extern "C" int gort_rsurf_members_stream(gort_engine *e, const double *angles, long nA, int member_begin,
                                         int member_end, double *rsurf)
{
    int rc = require_ready(e, "gort_rsurf_members_stream");
    if (rc) return rc;
    if (nA < 0 || member_begin < 0 || member_begin > member_end || (nA > 0 && member_end > member_begin && (!angles || !rsurf)))
        return fail(GORT_EINVAL, "gort_rsurf_members_stream: bad argument");
    if (nA == 0 || member_begin == member_end) return GORT_OK;
    const size_t n = (size_t)nA, total = n * (size_t)e->nw * (size_t)(member_end - member_begin);
    if ((rc = e->angles.reserve(sizeof(double) * 4 * n))) return rc;
    if ((rc = e->out.reserve(sizeof(double) * total))) return rc;
    GORT_HIP(hipMemcpyAsync(e->angles.p, angles, sizeof(double) * 4 * n, hipMemcpyHostToDevice, e->stream));
    if ((rc = gort_rsurf_members_stream_dev(e, e->angles.as<double>(), nA, member_begin, member_end, e->out.as<double>())))
        return rc;
    GORT_HIP(hipMemcpyAsync(rsurf, e->out.p, sizeof(double) * total, hipMemcpyDeviceToHost, e->stream));
    GORT_HIP(hipStreamSynchronize(e->stream));
    return GORT_OK;
}

// ------------------------------------------------------------------ LUT (grid)

static int size_class(long doubles)
{
    int cls = 0;
    for (long v = doubles; v > 1; v >>= 1) ++cls;
    return cls;
}

// rows are GLOBAL: member * (nsza*nvza) + isza * nvza + ivza
constexpr size_t PIPELINE_MAX_BYTES = 64u << 20;        // records + sun table of a call whose geometry runs under the previous expansion

static int grid_rows(gort_engine *e, const gort_grid *g, long row_begin, long row_end, double *lut_dev)
{
    int rc;
    const long rows = row_end - row_begin, nA = rows * g->nphi;
    const int nw = e->nw;
    const gort_canopy *c = e->canopy.as<gort_canopy>();
    // below 128 bands a chunk of the aligned LUT kernel (128 doubles) spans more than two angles: small grids and up to 64 bands the
    // geometry kernel writes the samples itself; [r6] 65 ... 127 bands of a grid worth three launches: records + expand_flat_few_kernel
    const bool few_flat = grid_takes_few_flat_kernel(nw, nA * (long)nw);
    const bool few_bands = nw < 128 && !few_flat;
    if (few_bands) {
        // up to 8 bands (config 3 has one): no records at all, the geometry kernel writes the samples itself - every row with its
        // own member's canopy and band constants (rows are global: member * rows_per_member + ...)
        static const bool fuse = !(ab_env("GORT_GRID_FUSE") && atoi(ab_env("GORT_GRID_FUSE")) == 0);
        if (nw <= 8 && fuse) return launch_geometry_grid_fused(c, e->L.as<double>(), nw, *g, row_begin, row_end, lut_dev, e->stream);
        // 9 ... 127 bands: the same kernel - lanes as bands, whole rows per store - with the LUT path's (sun zenith, band) table
        // beside it.  [r5] A hemisphere x 100 bands in 0.50 ms (0.65 through the stream kernels on the nodes written out as
        // angle lines, 2.7 through records + one thread per sample), x 16 bands 0.09 ms (0.29, 0.62); the LUT family's sample
        // like every other LUT path
        if (fuse) {
            const int q0 = (int)(row_begin / g->nvza), q1 = (int)((row_end - 1) / g->nvza) + 1;   // sun rows q = member*nsza + isza
            if ((rc = e->sun.reserve(sizeof(double) * 5 * (size_t)nw * (size_t)(q1 - q0)))) return rc;
            if ((rc = launch_sun_table(c, e->L.as<double>(), nw, *g, q0, q1, e->sun.as<double>(), e->stream))) return rc;
            return launch_geometry_grid_fused(c, e->L.as<double>(), nw, *g, row_begin, row_end, lut_dev, e->stream, e->sun.as<double>(), q0);
        }
        // GORT_GRID_FUSE=0 (measuring build): the two-kernel path the fused forms are held to - full angle records, then one
        // thread per sample (the LUT family's five-term sample), member = blockIdx.z.  Several members: whole members only
        const long rpm = (long)g->nsza * g->nvza;
        const long m0 = row_begin / rpm, m1 = (row_end - 1) / rpm + 1;
        if (m1 - m0 > 1 && (row_begin != m0 * rpm || row_end != m1 * rpm))
            return fail(GORT_EINVAL, "grid of %d bands: rows [%ld,%ld) are not whole members", nw, row_begin, row_end);
        if ((rc = e->coef.reserve(sizeof(double) * GORT_COEF_STRIDE * (size_t)nA))) return rc;
        if ((rc = launch_geometry_grid(c, *g, row_begin, row_end, e->coef.as<double>(), false, e->stream))) return rc;
        const double *Lm = e->L.as<double>() + (size_t)m0 * L_NSLOT * nw;
        if (m1 - m0 == 1)
            return launch_expand_stream(c + m0, Lm, nullptr, nw, e->coef.as<double>(), nA, lut_dev, nullptr, nullptr, e->stream, true);
        return launch_expand_grid_members(c + m0, Lm, nw, e->coef.as<double>(), nA / (m1 - m0), (int)(m1 - m0), lut_dev, e->stream);
    }
    // compact 64-B records, one pad record in front and a tail pad (see expand_flat_kernel).
    // Full-size slabs: ONE buffer, reused by every call - the 191 MB of records the geometry kernel writes are
    // still in the 256 MB Infinity Cache when the expansion fetches them.  Double-buffering them to overlap the
    // next call's geometry with this call's expansion cost 14 % there (8.0 against 7.0 ms): two buffers do not
    // fit, the record and sun-term fetches in the waves' prologues then come from HBM, and those short-lived
    // waves are latency-bound.  Small slabs (the per-rank slabs of a multi-GPU run): both halves fit, and the
    // pipeline hides the 0.04-0.06 ms of geometry, sun table and launch gaps per call (4 % at N = 8).
    long front = 1, tail = expand_grid_tail_pad_records(nw, nA * (long)nw);
    if (few_flat) expand_grid_few_pad_records(nw, nA * (long)nw, &front, &tail);
    const size_t coef_bytes = sizeof(double) * 8 * (size_t)(nA + front + tail);
    const int q0 = (int)(row_begin / g->nvza), q1 = (int)((row_end - 1) / g->nvza) + 1;   // sun rows q = member*nsza + isza
    const size_t sun_bytes = sizeof(double) * 5 * (size_t)nw * (size_t)(q1 - q0);
    const bool piped = e->pipeline && coef_bytes + sun_bytes <= PIPELINE_MAX_BYTES;
    const int half = piped ? (int)(e->grid_calls++ & 1) : 0;
    DevBuf &coef_buf = piped ? e->gcoef[half] : e->coef, &sun_buf = piped ? e->gsun[half] : e->sun;
    hipStream_t gs = piped ? e->aux : e->stream;
    if (piped) {
        if (e->tables_recorded) GORT_HIP(hipStreamWaitEvent(gs, e->ev_tables, 0));
        if (e->expand_recorded[half]) GORT_HIP(hipStreamWaitEvent(gs, e->ev_expand[half], 0));   // last reader of this half
    }
    const bool fresh = coef_bytes > coef_buf.cap;
    if ((rc = coef_buf.reserve(coef_bytes))) return rc;
    if (fresh) GORT_HIP(hipMemsetAsync(coef_buf.p, 0, coef_bytes, gs));      // pads: readable, contents don't-care (padded lanes are never stored)
    double *coef8 = coef_buf.as<double>() + 8 * front;
    if ((rc = launch_geometry_grid(c, *g, row_begin, row_end, coef8, true, gs))) return rc;
    if ((rc = sun_buf.reserve(sun_bytes))) return rc;
    if ((rc = launch_sun_table(c, e->L.as<double>(), nw, *g, q0, q1, sun_buf.as<double>(), gs))) return rc;
    if (piped) {
        GORT_HIP(hipEventRecord(e->ev_geom[half], gs));
        GORT_HIP(hipStreamWaitEvent(e->stream, e->ev_geom[half], 0));
    }
    int *xcd_slots = nullptr;
    if ((rc = xcd_slots_for_launch(e, &xcd_slots))) return rc;
    // slabs of 1 GiB or more: time the XCDs' write rates on the slab itself (it is overwritten right after) - once per
    // size class (power of two of the slab's size), since how unevenly the XCDs write depends on how far apart their
    // windows lie; weights set by hand (GORT_XCD_WEIGHTS, gort_engine_set_xcd_weights) are left alone
    if (!xcd_slots && !e->xcd_weights_fixed && nA * (long)nw >= (1L << 27)) {
        const int cls = size_class(nA * (long)nw);
        if (!e->xcd_calibrated || cls != e->xcd_cal_class) {
            if ((rc = calibrate_xcd_weights(e->stream, lut_dev, nA * (long)nw, e->xcd_weights, &e->xcd_pattern_gbs))) return rc;
            e->xcd_calibrated = true;
            e->xcd_cal_class = cls;
        }
    }
    // HIP events on the launch stream bracket the dominant kernel (bench.py roofline); up to
    // 512 launches are kept between two gort_engine_last_expand_ms() calls
    const bool timed = e->time_expand && e->ev_used + 2 <= 1024;
    if (timed) {
        while (e->ev.size() < e->ev_used + 2) {
            hipEvent_t ev;
            GORT_HIP(hipEventCreate(&ev));
            e->ev.push_back(ev);
        }
        GORT_HIP(hipEventRecord(e->ev[e->ev_used], e->stream));
    }
    rc = few_flat ? launch_expand_grid_few(sun_buf.as<double>(), q0, q1 - q0, coef8, nw, g->nvza, g->nphi, row_begin, row_end, lut_dev,
                                           xcd_slots, e->xcd_weights, e->stream)
                  : launch_expand_grid(sun_buf.as<double>(), q0, coef8, nw, g->nvza, g->nphi, row_begin, row_end, lut_dev,
                                       xcd_slots, e->xcd_weights, e->stream);
    if (timed) {
        GORT_HIP(hipEventRecord(e->ev[e->ev_used + 1], e->stream));
        e->ev_used += 2;
    }
    if (piped) {
        GORT_HIP(hipEventRecord(e->ev_expand[half], e->stream));
        e->expand_recorded[half] = true;
    }
    return rc;
}

// (Round 6 tried cutting a call whose records do not fit the pipeline's double buffer - a hemisphere: 191 MB - into slabs that
// do, so that the geometry of slab i + 1 runs under the expansion of slab i as it does between the calls of a slabbed grid: five
// slabs of a hemisphere x 100 / 128 / 512 bands took 545 / 640 / 1930 us against 450 / 480 / 1690 in one piece - every slab's
// expansion has a head and a tail of its own, and the geometry kernel is sized for the machine, not for a fifth of it.
// profiles/r06/few_band_flat_ab.log)

// ---- LUT buffers with a measured placement (include/gort_amd.h) ----
// buffers handed out as a pointer INTO their allocation (shifted windows): pointer -> what hipFree needs
static std::mutex &lut_bases_mu()
{
    static std::mutex m;
    return m;
}
static std::unordered_map<void *, void *> &lut_bases()
{
    static std::unordered_map<void *, void *> m;
    return m;
}


extern "C" int gort_lut_alloc(gort_engine *e, size_t bytes, size_t win_offset, size_t win_bytes, int max_draws,
                              void **out_dev, gort_lut_placement *info)
{
    if (out_dev) *out_dev = nullptr;
    if (info) std::memset(info, 0, sizeof *info);
    if (!e || !out_dev || bytes == 0 || win_offset % sizeof(double) || win_bytes % sizeof(double) || win_offset > bytes ||
        win_bytes > bytes - win_offset)
        return fail(GORT_EINVAL, "gort_lut_alloc: bad argument");
    if (win_bytes == 0) { win_offset = 0; win_bytes = bytes / sizeof(double) * sizeof(double); }
    // the engine's device, whatever the calling thread had current - and the caller's device again on every way out
    struct DeviceGuard {
        int prev = -1;
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } guard;
    if (hipGetDevice(&guard.prev) != hipSuccess) { (void)hipGetLastError(); guard.prev = -1; }
    if (guard.prev == e->device) guard.prev = -1;      // nothing to restore
    GORT_HIP(hipSetDevice(e->device));
    if (max_draws < 1) max_draws = 1;
    if (max_draws > GORT_LUT_MAX_DRAWS) max_draws = GORT_LUT_MAX_DRAWS;
    const long doubles = (long)(win_bytes / sizeof(double));
    // windows below 1 GiB have no stable rate to select on (and the pattern probe needs 64 panels): first draw
    int *slots = nullptr;
    int rc = xcd_slots_for_launch(e, &slots);          // also probes the dispatch order once
    if (rc) return rc;
    // (round 3 stopped selecting at 32 GiB: the boxes it had seen wrote such windows at 7.1-7.5 TB/s wherever they lay.
    // Round 4 met boxes where two of four 50 GB allocations alive together run the LUT kernel at 7.45 ms and two at 6.72,
    // every time, and the probe tells them apart - 6.2 against 7.15 TB/s: profiles/r04/placement_select.log)
    const bool select = max_draws > 1 && doubles >= (1L << 27) && !slots;
    const int cls = size_class(doubles);
    const double accept = select ? 0.985 * e->best_pattern_gbs[cls] : 0.0;
    void *cand[GORT_LUT_MAX_DRAWS] = {nullptr};
    int weights[GORT_LUT_MAX_DRAWS][8];
    double gbs[GORT_LUT_MAX_DRAWS] = {0.0};
    int n = 0, best = 0;
    // passes of the LUT kernel's bare store pattern over a candidate window (the first one touches the pages)
    auto probe = [&](void *buffer, int i, int passes) -> int {
        double *win = reinterpret_cast<double *>(static_cast<char *>(buffer) + win_offset);
        for (int pass = 0; pass < passes; ++pass) {
            double g = 0.0;
            int w[8];
            const int prc = calibrate_xcd_weights(e->stream, win, doubles, w, &g);
            if (prc) return prc;
            if ((pass > 0 || passes == 1) && g > gbs[i]) { gbs[i] = g; std::memcpy(weights[i], w, sizeof w); }
        }
        if (gbs[i] > gbs[best]) best = i;
        return GORT_OK;
    };
    // A window that is at most half the buffer (a rank's slab of a gatherable LUT: the other ranks' windows are only
    // ever written by the all-gather) is placed by a SCAN inside one allocation.  Measured (profiles/r03/
    // placement_scan.log): the rate of a 6 - 25 GB window as a function of where it lies in a big allocation is a
    // comb - 7.2-7.3 TB/s on plateaus 3-5 GiB wide that recur every 8 to 48 GiB (wherever the window straddles a
    // boundary of the physical extents behind the allocation), 6.1-6.3 TB/s in between - so a handful of random
    // placements mostly land in between, and a scan in 1-GiB steps over up to 48 GiB of slack finds a plateau.  The
    // slack (at most what leaves 8 GiB of the device free) stays allocated with the buffer.
    const bool scan = select && win_bytes * 2 <= bytes;
    void *base = nullptr;
    size_t slack_bytes = 0;
    if (scan) {
        constexpr size_t GIB = (size_t)1 << 30;
        // An allocation that lies inside ONE physical extent has no plateau anywhere (a fresh process's first 76 GiB:
        // flat at 6.0-6.35 TB/s over 59 GiB, placement_scan_rotate.log).  Then - the best candidate is not 8 % above the
        // median - the scan is repeated on a new allocation made behind a blocker of a few GiB, which the driver backs
        // with other extents (the same virtual range, re-allocated behind a 5 GiB blocker: a dense comb); twice at most.
        void *blockers[2] = {nullptr, nullptr};
        for (int attempt = 0; attempt < 3 && rc == GORT_OK; ++attempt) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
            // the cap of the slack that stays allocated with the buffer (GORT_LUT_SLACK_GIB when the engine was created,
            // gort_engine_set_lut_slack_gib; default 48; 0 = no scan, a plain allocation).  Ranks that share a GPU, or a
            // caching allocator beside this one, want it small.
            size_t slack_gib = (size_t)e->lut_slack_gib;
            while (slack_gib > 0 && bytes + slack_gib * GIB + 8 * GIB > free_b) slack_gib /= 2;
            slack_bytes = 0;
            for (;; slack_gib /= 2) {
                if (hipMalloc(&base, bytes + slack_gib * GIB) == hipSuccess) break;
                (void)hipGetLastError();
                base = nullptr;
                if (slack_gib == 0) break;
            }
            if (!base) { rc = fail(GORT_ENOMEM, "gort_lut_alloc: cannot allocate %zu bytes", bytes); break; }
            slack_bytes = slack_gib * GIB;
            int cands = (int)slack_gib + 1;
            if (cands > GORT_LUT_MAX_DRAWS) cands = GORT_LUT_MAX_DRAWS;
            const size_t step = cands > 1 ? slack_gib * GIB / (size_t)(cands - 1) / (2u << 20) * (2u << 20) : 0;
            best = 0;
            for (n = 0; n < cands && rc == GORT_OK; ++n) {
                gbs[n] = 0.0;
                cand[n] = static_cast<char *>(base) + (size_t)n * step;
                rc = probe(cand[n], n, n == 0 ? 3 : 2);      // neighbours overlap: only the first one meets untouched pages
                if (rc == GORT_OK && accept > 0.0 && gbs[n] >= accept) { ++n; break; }
            }
            if (rc) break;
            std::vector<double> sorted(gbs, gbs + n);
            std::sort(sorted.begin(), sorted.end());
            const bool plateau = (accept > 0.0 && gbs[best] >= accept) || n < 8 || gbs[best] >= 1.08 * sorted[(size_t)n / 2];
            if (plateau || attempt == 2) break;
            (void)hipFree(base);                         // a flat comb: once more, on other extents
            base = nullptr;
            if (hipMalloc(&blockers[attempt], (size_t)(attempt ? 11 : 5) * GIB) != hipSuccess) {
                (void)hipGetLastError();
                blockers[attempt] = nullptr;
            }
            if (info) ++info->rescans;
        }
        for (void *b : blockers) if (b) (void)hipFree(b);
        if (rc) {
            if (base) (void)hipFree(base);
            return rc;
        }
        if (cand[best] != base) {                       // an interior pointer: gort_lut_free must find the allocation
            std::lock_guard<std::mutex> lock(lut_bases_mu());
            lut_bases()[cand[best]] = base;
        }
    } else {
        // the window is (most of) the buffer: separate allocations, alive together (a freed candidate's memory would
        // come straight back) as far as the device has room for them beside 64 GiB for everybody else
        const int draws = select ? (max_draws < 5 ? max_draws : 5) : 1;
        for (; n < draws; ++n) {
            size_t free_b = 0, total_b = 0;
            if (n > 0 && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + ((size_t)64 << 30))) {
                (void)hipGetLastError();
                break;
            }
            if (hipMalloc(&cand[n], bytes) != hipSuccess) {
                (void)hipGetLastError();
                cand[n] = nullptr;
                if (n == 0) return fail(GORT_ENOMEM, "gort_lut_alloc: cannot allocate %zu bytes", bytes);
                break;                                      // make do with the draws we have
            }
            if (!select) continue;
            if ((rc = probe(cand[n], n, 3))) break;
            if (accept > 0.0 && gbs[n] >= accept) { ++n; break; }      // as good as anything this engine has seen
        }
        if (rc) {
            for (int i = 0; i < GORT_LUT_MAX_DRAWS; ++i) if (cand[i]) (void)hipFree(cand[i]);
            return rc;
        }
        for (int i = 0; i < n; ++i)
            if (i != best && cand[i]) (void)hipFree(cand[i]);
    }
    if (select) {
        if (gbs[best] > e->best_pattern_gbs[cls]) e->best_pattern_gbs[cls] = gbs[best];
        if (!e->xcd_weights_fixed && gbs[best] > 0.0) {        // the XCD duty weights of this size class come with the probe
            std::memcpy(e->xcd_weights, weights[best], sizeof e->xcd_weights);
            e->xcd_calibrated = true;
            e->xcd_cal_class = cls;
            e->xcd_pattern_gbs = gbs[best];
        }
    }
    if (info) {
        info->draws = n;
        info->picked = best;
        info->accept_gbs = accept;
        info->shifted = scan ? 1 : 0;
        info->slack_bytes = (uint64_t)slack_bytes;
        for (int i = 0; i < n; ++i) info->probe_gbs[i] = gbs[i];
    }
    *out_dev = cand[best];
    return GORT_OK;
}

// the all-gather of a row-sharded LUT on the engine's stream (gort_rccl.cpp): asynchronous like every device entry point
extern "C" int gort_lut_allgather_on(void *stream, void *lut_dev, size_t rows_per_rank, size_t row_bytes, int rank, int world, void *comm);
extern "C" int gort_lut_allgather(gort_engine *e, void *lut_dev, size_t rows_per_rank, size_t row_bytes, int rank, int world, void *comm)
{
    if (!e) return fail(GORT_EINVAL, "gort_lut_allgather: null engine");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    GORT_HIP(hipSetDevice(e->device));
    const int rc = gort_lut_allgather_on(e->stream, lut_dev, rows_per_rank, row_bytes, rank, world, comm);
    if (prev >= 0 && prev != e->device) (void)hipSetDevice(prev);      // the caller's device, as it was
    return rc;
}

// the probe of gort_lut_alloc on memory the caller owns (contents destroyed): include/gort_amd_tuning.h
extern "C" double gort_engine_probe_store_pattern(gort_engine *e, void *dev, size_t bytes)
{
    if (!e || !dev || bytes < sizeof(double)) return (double)fail(GORT_EINVAL, "gort_engine_probe_store_pattern: bad argument");
    double best = 0.0;
    for (int pass = 0; pass < 3; ++pass) {
        double g = 0.0;
        int w[8];
        const int rc = calibrate_xcd_weights(e->stream, static_cast<double *>(dev), (long)(bytes / sizeof(double)), w, &g);
        if (rc) return (double)rc;
        if (pass > 0 && g > best) best = g;
    }
    return best;
}

extern "C" void gort_lut_free(void *lut_dev)
{
    if (!lut_dev) return;
    void *base = lut_dev;
    {
        std::lock_guard<std::mutex> lock(lut_bases_mu());
        auto it = lut_bases().find(lut_dev);
        if (it != lut_bases().end()) {
            base = it->second;
            lut_bases().erase(it);
        }
    }
    (void)hipFree(base);
}

static int check_grid(const gort_grid *g, const char *who)
{
    if (!g || g->nsza <= 0 || g->nvza <= 0 || g->nphi <= 0) return fail(GORT_EINVAL, "%s: bad grid", who);
    // the kernels track (view zenith, azimuth) nodes per sun row in 32-bit
    if ((long)g->nvza * g->nphi >= (1L << 30)) return fail(GORT_EINVAL, "%s: nvza*nphi too large", who);
    return GORT_OK;
}

extern "C" int gort_rsurf_grid_dev(gort_engine *e, const gort_grid *g, long row_begin, long row_end, double *lut_dev)
{
    int rc = require_ready(e, "gort_rsurf_grid_dev");
    if (rc) return rc;
    if ((rc = check_grid(g, "gort_rsurf_grid_dev"))) return rc;
    const long rows_total = (long)g->nsza * g->nvza;       // member 0
    if (row_begin < 0 || row_end > rows_total || row_begin > row_end)
        return fail(GORT_EINVAL, "gort_rsurf_grid_dev: rows [%ld,%ld) outside [0,%ld)", row_begin, row_end, rows_total);
    if (row_begin == row_end) return GORT_OK;
    if (!lut_dev) return fail(GORT_EINVAL, "gort_rsurf_grid_dev: null output");
    return grid_rows(e, g, row_begin, row_end, lut_dev);
}

extern "C" int gort_rsurf_members_grid_dev(gort_engine *e, const gort_grid *g, int member_begin, int member_end,
                                           double *lut_dev)
{
    int rc = require_ready(e, "gort_rsurf_members_grid_dev");
    if (rc) return rc;
    if ((rc = check_grid(g, "gort_rsurf_members_grid_dev"))) return rc;
    if (member_begin < 0 || member_end > e->n_members || member_begin > member_end)
        return fail(GORT_EINVAL, "gort_rsurf_members_grid_dev: members [%d,%d) outside [0,%d)", member_begin, member_end,
                    e->n_members);
    if (member_begin == member_end) return GORT_OK;
    if (!lut_dev) return fail(GORT_EINVAL, "gort_rsurf_members_grid_dev: null output");
    const long rpm = (long)g->nsza * g->nvza;
    return grid_rows(e, g, member_begin * rpm, member_end * rpm, lut_dev);
}

extern "C" double gort_engine_last_expand_ms(gort_engine *e)
{
    if (!e || e->ev_used == 0) return -1.0;
    double sum = 0.0;
    const size_t n = e->ev_used / 2;
    for (size_t i = 0; i < n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(e->ev[2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms, e->ev[2 * i], e->ev[2 * i + 1]) != hipSuccess)
            return -1.0;
        sum += ms;
    }
    e->ev_used = 0;
    return sum / (double)n;
}

// ---------------------------------------------------------------------- energy

static int ensure_nodes(gort_engine *e)
{
    if (e->have_nodes) return GORT_OK;
    const int np = GORT_NPOINTS;
    double x[GORT_NPOINTS], w[GORT_NPOINTS];
    gort_gauleg(-1., 1., x, w, np);
    // outer: azimuth nodes vaa = pi + pi*x_i; inner: zenith nodes from the positive half of the
    // same rule, vza = acos(x_j), weight w_j |x_j| (gortt_albedo.c:82-133), result / pi (:136)
    const double xm = 0.5 * (1. - 1.), xr = 0.5 * (1. + 1.);
    const double ym = 0.5 * (2. * M_PI - 0.), yr = 0.5 * (2. * M_PI + 0.);
    std::vector<double> nodes(3 * 512);
    int n = 0;
    for (int i = 0; i < np; ++i)
        for (int j = np / 2; j < np; ++j, ++n) {
            const double xx = xm + xr * x[j];
            nodes[3 * n] = ym + yr * x[i];
            nodes[3 * n + 1] = std::acos(xx);
            nodes[3 * n + 2] = (w[j] * std::fabs(xx) * xr) * (w[i] * yr) / M_PI;
        }
    int rc = e->nodes.reserve(sizeof(double) * nodes.size());
    if (rc) return rc;
    GORT_HIP(hipMemcpy(e->nodes.p, nodes.data(), sizeof(double) * nodes.size(), hipMemcpyHostToDevice));
    e->have_nodes = true;
    return GORT_OK;
}

// scratch of the sun-direction table for nA lines (nullptr: few lines, or switched off)
static int energy_workspace(gort_engine *e, long nA, void **ws)
{
    *ws = nullptr;
    const size_t bytes = e->energy_dedup ? energy_dedup_workspace(nA) : 0;
    if (!bytes) return GORT_OK;
    const int rc = e->edup.reserve(bytes);
    if (rc) return rc;
    *ws = e->edup.p;
    return GORT_OK;
}

extern "C" int gort_energy_stream_dev(gort_engine *e, const double *angles_dev, long nA, double *energy_dev)
{
    int rc = require_ready(e, "gort_energy_stream_dev");
    if (rc) return rc;
    if (nA < 0 || (nA > 0 && (!angles_dev || !energy_dev))) return fail(GORT_EINVAL, "gort_energy_stream_dev: bad argument");
    if (nA == 0) return GORT_OK;
    if ((rc = ensure_nodes(e))) return rc;
    void *ws = nullptr;
    if ((rc = energy_workspace(e, nA, &ws))) return rc;
    int *slots = nullptr;
    if ((rc = xcd_slots_for_launch(e, &slots))) return rc;              // probes the dispatch order once
    return launch_energy(e->canopy.as<gort_canopy>(), 1, e->L.as<double>(), e->nw, angles_dev, nA,
                         e->nodes.as<double>(), energy_dev, ws, e->xcd_round_robin == 1, e->stream);
}

// ---- the indexed form: distinct rows + one index per line ----

int gort_engine_energy_table(gort_engine *e, const double *angles_dev, long nA, void *ws_dev, uint32_t *n_rows_dev, void *stream)
{
    if (!e || nA < 0 || (nA > 0 && (!angles_dev || !ws_dev))) return fail(GORT_EINVAL, "gort_engine_energy_table: bad argument");
    return launch_energy_table(angles_dev, nA, ws_dev, n_rows_dev, stream);
}

int gort_engine_energy_rows(gort_engine *e, const double *angles_dev, long nA, const void *ws_dev, long n_rows, double *rows_dev)
{
    int rc = require_ready(e, "gort_engine_energy_rows");
    if (rc) return rc;
    if (nA <= 0 || n_rows <= 0) return GORT_OK;
    if ((rc = ensure_nodes(e))) return rc;
    return launch_energy_rows(e->canopy.as<gort_canopy>(), 1, e->L.as<double>(), e->nw, angles_dev, nA, e->nodes.as<double>(),
                              rows_dev, n_rows, ws_dev, n_rows, e->stream);
}

extern "C" int gort_energy_stream_indexed_dev(gort_engine *e, const double *angles_dev, long nA, double *rows_dev, long rows_cap,
                                              uint32_t *index_dev, uint32_t *n_rows_dev)
{
    int rc = require_ready(e, "gort_energy_stream_indexed_dev");
    if (rc) return rc;
    if (nA < 0 || rows_cap < 0 || (nA > 0 && (!angles_dev || !index_dev || (rows_cap > 0 && !rows_dev))))
        return fail(GORT_EINVAL, "gort_energy_stream_indexed_dev: bad argument");
    if (nA == 0) {
        if (n_rows_dev) GORT_HIP(hipMemsetAsync(n_rows_dev, 0, sizeof(uint32_t), e->stream));
        return GORT_OK;
    }
    if ((rc = ensure_nodes(e))) return rc;
    if ((rc = e->edup.reserve(energy_table_workspace(nA)))) return rc;
    if ((rc = launch_energy_table(angles_dev, nA, e->edup.p, n_rows_dev, e->stream))) return rc;
    GORT_HIP(hipMemcpyAsync(index_dev, energy_table_index(e->edup.p, nA), sizeof(uint32_t) * (size_t)nA, hipMemcpyDeviceToDevice, e->stream));
    return launch_energy_rows(e->canopy.as<gort_canopy>(), 1, e->L.as<double>(), e->nw, angles_dev, nA, e->nodes.as<double>(),
                              rows_dev, rows_cap, e->edup.p, -1, e->stream);
}

extern "C" int gort_energy_stream_indexed(gort_engine *e, const double *angles, long nA, double *rows, long rows_cap,
                                          uint32_t *index, long *n_rows)
{
    int rc = require_ready(e, "gort_energy_stream_indexed");
    if (rc) return rc;
    if (n_rows) *n_rows = 0;
    if (nA < 0 || rows_cap < 0 || !n_rows || (nA > 0 && (!angles || !index || (rows_cap > 0 && !rows))))
        return fail(GORT_EINVAL, "gort_energy_stream_indexed: bad argument");
    if (nA == 0) return GORT_OK;
    const size_t n = (size_t)nA, row_bytes = sizeof(double) * 3 * (size_t)e->nw;
    if ((rc = ensure_nodes(e))) return rc;
    if ((rc = e->angles.reserve(sizeof(double) * 4 * n))) return rc;
    if ((rc = e->edup.reserve(energy_table_workspace(nA)))) return rc;
    GORT_HIP(hipMemcpyAsync(e->angles.p, angles, sizeof(double) * 4 * n, hipMemcpyHostToDevice, e->stream));
    if ((rc = launch_energy_table(e->angles.as<double>(), nA, e->edup.p, nullptr, e->stream))) return rc;
    unsigned count = 0;
    GORT_HIP(hipMemcpyAsync(&count, energy_table_count(e->edup.p, nA), sizeof count, hipMemcpyDeviceToHost, e->stream));
    GORT_HIP(hipMemcpyAsync(index, energy_table_index(e->edup.p, nA), sizeof(uint32_t) * n, hipMemcpyDeviceToHost, e->stream));
    GORT_HIP(hipStreamSynchronize(e->stream));
    *n_rows = (long)count;
    if ((long)count > rows_cap)
        return fail(GORT_ERANGE, "gort_energy_stream_indexed: %u distinct sun directions, room for %ld rows", count, rows_cap);
    if ((rc = e->out.reserve(row_bytes * count))) return rc;
    if ((rc = launch_energy_rows(e->canopy.as<gort_canopy>(), 1, e->L.as<double>(), e->nw, e->angles.as<double>(), nA,
                                 e->nodes.as<double>(), e->out.as<double>(), (long)count, e->edup.p, (long)count, e->stream)))
        return rc;
    GORT_HIP(hipMemcpyAsync(rows, e->out.p, row_bytes * count, hipMemcpyDeviceToHost, e->stream));
    GORT_HIP(hipStreamSynchronize(e->stream));
    return GORT_OK;
}

extern "C" int gort_energy_members_dev(gort_engine *e, const double *angles_dev, long nA, int member_begin,
                                       int member_end, double *energy_dev)
{
    int rc = require_ready(e, "gort_energy_members_dev");
    if (rc) return rc;
    if (member_begin < 0 || member_end > e->n_members || member_begin > member_end)
        return fail(GORT_EINVAL, "gort_energy_members_dev: members [%d,%d) outside [0,%d)", member_begin, member_end,
                    e->n_members);
    if (nA < 0 || (nA > 0 && member_end > member_begin && (!angles_dev || !energy_dev)))
        return fail(GORT_EINVAL, "gort_energy_members_dev: bad argument");
    if (nA == 0 || member_begin == member_end) return GORT_OK;
    if ((rc = ensure_nodes(e))) return rc;
    void *ws = nullptr;
    if ((rc = energy_workspace(e, nA, &ws))) return rc;
    int *slots = nullptr;
    if ((rc = xcd_slots_for_launch(e, &slots))) return rc;
    hipStream_t s = e->stream;
    if (e->energy_on_side && e->side) {
        // behind the tables (recorded on the main stream by the setters) and behind whatever the main stream holds now: the
        // nodes' upload, an earlier call's workspace
        s = e->side;
        if (!e->ev_side) GORT_HIP(hipEventCreateWithFlags(&e->ev_side, hipEventDisableTiming));
        GORT_HIP(hipEventRecord(e->ev_side, e->stream));
        GORT_HIP(hipStreamWaitEvent(s, e->ev_side, 0));
    }
    return launch_energy(e->canopy.as<gort_canopy>() + member_begin, member_end - member_begin,
                         e->L.as<double>() + (size_t)member_begin * L_NSLOT * e->nw, e->nw, angles_dev, nA,
                         e->nodes.as<double>(), energy_dev, ws, e->xcd_round_robin == 1, s);
}

extern "C" int gort_energy_stream(gort_engine *e, const double *angles, long nA, double *energy)
{
    int rc = require_ready(e, "gort_energy_stream");
    if (rc) return rc;
    if (nA < 0 || (nA > 0 && (!angles || !energy))) return fail(GORT_EINVAL, "gort_energy_stream: bad argument");
    if (nA == 0) return GORT_OK;
    const size_t n = (size_t)nA, nw = (size_t)e->nw;
    if (sizeof(double) * 3 * n * nw >= (4u << 20) && !is_pinned_host(energy))
        return staged_stream(e, angles, nA, nullptr, nullptr, nullptr, energy);
    if ((rc = e->angles.reserve(sizeof(double) * 4 * n))) return rc;
    if ((rc = e->out.reserve(sizeof(double) * 3 * n * nw))) return rc;
    GORT_HIP(hipMemcpyAsync(e->angles.p, angles, sizeof(double) * 4 * n, hipMemcpyHostToDevice, e->stream));
    if ((rc = gort_energy_stream_dev(e, e->angles.as<double>(), nA, e->out.as<double>()))) return rc;
    GORT_HIP(hipMemcpyAsync(energy, e->out.p, sizeof(double) * 3 * n * nw, hipMemcpyDeviceToHost, e->stream));
    GORT_HIP(hipStreamSynchronize(e->stream));
    return GORT_OK;
}
