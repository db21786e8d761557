// gort_pipe.hip -- chunks of an angle stream in flight: host -> device copies, kernels and device -> host copies
// of consecutive chunks overlap, and the host keeps parsing / formatting meanwhile.
//
// The reference evaluates one line at a time between a sscanf and a printf (gortt.c:232-329).  Here a chunk of
// lines travels through three engines that run side by side:
//
//     copy-in stream    angles of chunk i+1   (pinned host -> HBM)
//     engine stream     kernels of chunk i    (gort_rsurf_stream_dev, gort_energy_stream_dev)
//     copy-out stream   results of chunk i-1  (HBM -> pinned host)
//
// ordered by events, never by host waits.  Every slot owns pinned host buffers and device buffers for one chunk;
// the host side fills `angles` of an acquired slot, submits it, and later collects the oldest chunk's results
// (gort_pipe_wait sleeps on the copy-out event).  acquire/submit and wait/release may run on two threads (a
// producer that parses and a consumer that formats): the `gortt` executable does exactly that.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "gort_internal.h"

using namespace gort;

#define PIPE_HIP(call)                                                                               \
    do {                                                                                            \
        hipError_t err__ = (call);                                                                  \
        if (err__ != hipSuccess)                                                                    \
            return fail(err__ == hipErrorOutOfMemory ? GORT_ENOMEM : GORT_ENODEVICE, "%s: %s", #call, \
                        hipGetErrorString(err__));                                                  \
    } while (0)

namespace {

struct Slot {
    double *h_ang = nullptr, *h_rsurf = nullptr, *h_scomp = nullptr, *h_K = nullptr, *h_energy = nullptr;   // pinned
    double *d_ang = nullptr, *d_rsurf = nullptr, *d_scomp = nullptr, *d_K = nullptr, *d_energy = nullptr;   // HBM
    // GORT_PIPE_ENERGY_INDEXED: the sun-direction table of the chunk (device), its line index and row count (pinned host)
    void *d_table = nullptr;
    uint32_t *h_index = nullptr, *h_rows = nullptr;
    hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_out = nullptr;
    long n = 0, energy_rows = 0;
    long energy_cap = 0;         // rows h_energy / d_energy hold (indexed mode: grown when a chunk has more distinct rows)
    int rc = GORT_OK;            // error raised while the chunk was submitted; reported by gort_pipe_wait
    std::string err;             // its message
};

}  // namespace

struct gort_pipe {
    gort_engine *e = nullptr;
    int device = 0;              // the device that was current at creation: every entry point selects it
    long max_lines = 0;
    int nw = 0, depth = 0;
    unsigned flags = 0;
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<Slot> slots;
    std::mutex mu;
    std::condition_variable cv;
    long acquired = 0, submitted = 0, waited = 0, released = 0;     // running counts; slot = count % depth
};

extern "C" void *gort_host_malloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        fail(GORT_ENOMEM, "gort_host_malloc: cannot pin %zu bytes", bytes);
        return nullptr;
    }
    return p;
}

extern "C" void gort_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

extern "C" void gort_pipe_destroy(gort_pipe *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->s_in) (void)hipStreamSynchronize(p->s_in);
    if (p->e) (void)gort_engine_synchronize(p->e);
    if (p->s_out) (void)hipStreamSynchronize(p->s_out);
    for (Slot &s : p->slots) {
        for (double *h : {s.h_ang, s.h_rsurf, s.h_scomp, s.h_K, s.h_energy}) if (h) (void)hipHostFree(h);
        for (double *d : {s.d_ang, s.d_rsurf, s.d_scomp, s.d_K, s.d_energy}) if (d) (void)hipFree(d);
        if (s.d_table) (void)hipFree(s.d_table);
        if (s.h_index) (void)hipHostFree(s.h_index);
        if (s.h_rows) (void)hipHostFree(s.h_rows);
        for (hipEvent_t ev : {s.ev_in, s.ev_k, s.ev_out}) if (ev) (void)hipEventDestroy(ev);
    }
    if (p->s_in) (void)hipStreamDestroy(p->s_in);
    if (p->s_out) (void)hipStreamDestroy(p->s_out);
    delete p;
}

static int pipe_alloc(gort_pipe *p)
{
    const size_t n = (size_t)p->max_lines, nw = (size_t)p->nw, D = sizeof(double);
    PIPE_HIP(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
    PIPE_HIP(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
    p->slots.resize((size_t)p->depth);
    for (Slot &s : p->slots) {
        auto both = [&](double **h, double **d, size_t bytes) -> int {
            if (bytes == 0) bytes = D;
            PIPE_HIP(hipHostMalloc((void **)h, bytes, hipHostMallocDefault));
            PIPE_HIP(hipMalloc((void **)d, bytes));
            return GORT_OK;
        };
        int rc;
        if ((rc = both(&s.h_ang, &s.d_ang, D * 4 * n))) return rc;
        if (!(p->flags & GORT_PIPE_ENERGY_ONLY)) {
            if ((rc = both(&s.h_rsurf, &s.d_rsurf, D * n * nw))) return rc;
            if ((rc = both(&s.h_K, &s.d_K, D * 4 * n))) return rc;
        }
        if ((p->flags & GORT_PIPE_SCOMP) && (rc = both(&s.h_scomp, &s.d_scomp, D * 4 * n * nw))) return rc;
        if (p->flags & GORT_PIPE_ENERGY) {
            // indexed: room for the distinct rows of a typical chunk (16 MiB; a million lines of a 1-degree sun grid have 91) - a
            // chunk with more makes its slot grow (gort_pipe_submit knows the count before anything is queued for the rows)
            s.energy_cap = (long)n;
            if (p->flags & GORT_PIPE_ENERGY_INDEXED) {
                const long typical = (long)(((size_t)16 << 20) / (D * 3 * (nw ? nw : 1)));
                if (typical < s.energy_cap) s.energy_cap = typical < 1 ? 1 : typical;
            }
            if ((rc = both(&s.h_energy, &s.d_energy, D * 3 * (size_t)s.energy_cap * nw))) return rc;
        }
        if (p->flags & GORT_PIPE_ENERGY_INDEXED) {
            PIPE_HIP(hipMalloc(&s.d_table, energy_table_workspace(p->max_lines)));
            PIPE_HIP(hipHostMalloc((void **)&s.h_index, sizeof(uint32_t) * n, hipHostMallocDefault));
            PIPE_HIP(hipHostMalloc((void **)&s.h_rows, 64, hipHostMallocDefault));
        }
        PIPE_HIP(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
        PIPE_HIP(hipEventCreateWithFlags(&s.ev_k, hipEventDisableTiming));
        PIPE_HIP(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming | hipEventBlockingSync));
    }
    return GORT_OK;
}

extern "C" int gort_pipe_create(gort_engine *e, long max_lines, int depth, unsigned flags, gort_pipe **out)
{
    if (!out) return fail(GORT_EINVAL, "gort_pipe_create: null out");
    *out = nullptr;
    if (!e || max_lines <= 0 || depth < 1 || depth > 16) return fail(GORT_EINVAL, "gort_pipe_create: bad argument");
    if ((flags & GORT_PIPE_ENERGY_ONLY) && (flags & GORT_PIPE_SCOMP)) return fail(GORT_EINVAL, "gort_pipe_create: ENERGY_ONLY with SCOMP");
    if (flags & GORT_PIPE_ENERGY_ONLY) flags |= GORT_PIPE_ENERGY;
    if ((flags & GORT_PIPE_ENERGY_INDEXED) && !(flags & GORT_PIPE_ENERGY))
        return fail(GORT_EINVAL, "gort_pipe_create: ENERGY_INDEXED without ENERGY or ENERGY_ONLY");
    gort_pipe *p = new (std::nothrow) gort_pipe();
    if (!p) return fail(GORT_ENOMEM, "gort_pipe_create: out of memory");
    p->e = e;
    if (hipGetDevice(&p->device) != hipSuccess) p->device = 0;
    p->max_lines = max_lines;
    p->nw = gort_engine_nw(e);
    p->depth = depth;
    p->flags = flags;
    const int rc = pipe_alloc(p);
    if (rc) {
        p->e = nullptr;              // nothing of ours is queued on the engine
        gort_pipe_destroy(p);
        return rc;
    }
    *out = p;
    return GORT_OK;
}

extern "C" int gort_pipe_acquire(gort_pipe *p, double **angles)
{
    if (!p || !angles) return fail(GORT_EINVAL, "gort_pipe_acquire: bad argument");
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->acquired != p->submitted) return fail(GORT_EINVAL, "gort_pipe_acquire: the acquired slot was not submitted");
    p->cv.wait(lk, [&] { return p->acquired - p->released < p->depth; });
    *angles = p->slots[(size_t)(p->acquired % p->depth)].h_ang;
    ++p->acquired;
    return GORT_OK;
}

extern "C" int gort_pipe_submit(gort_pipe *p, long n)
{
    if (!p) return fail(GORT_EINVAL, "gort_pipe_submit: null pipe");
    Slot *sp;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (p->acquired != p->submitted + 1) return fail(GORT_EINVAL, "gort_pipe_submit: no slot acquired");
        sp = &p->slots[(size_t)(p->submitted % p->depth)];
    }
    Slot &s = *sp;
    if (n < 0 || n > p->max_lines) {
        // recorded as a failed, empty chunk: every acquired slot is submitted, so a consumer that counts chunks
        // (the CLI does) meets the error in gort_pipe_wait instead of waiting for a chunk that never comes
        s.n = 0;
        s.rc = fail(GORT_EINVAL, "gort_pipe_submit: %ld lines in a slot of %ld", n, p->max_lines);
        s.err = gort_last_error();
        {
            std::lock_guard<std::mutex> lk(p->mu);
            ++p->submitted;
        }
        p->cv.notify_all();
        return s.rc;
    }
    s.n = n;
    s.energy_rows = (p->flags & GORT_PIPE_ENERGY) && !(p->flags & GORT_PIPE_ENERGY_INDEXED) && p->nw > 0 ? n : 0;
    s.rc = GORT_OK;
    s.err.clear();
    (void)hipSetDevice(p->device);
    hipStream_t ks = (hipStream_t)gort_engine_stream(p->e);
    const size_t nn = (size_t)n, nw = (size_t)p->nw, D = sizeof(double);
    auto enqueue = [&]() -> int {
        if (n == 0) {
            PIPE_HIP(hipEventRecord(s.ev_out, p->s_out));
            return GORT_OK;
        }
        PIPE_HIP(hipMemcpyAsync(s.d_ang, s.h_ang, D * 4 * nn, hipMemcpyHostToDevice, p->s_in));
        const bool indexed = s.d_table && nw > 0;
        int rc = GORT_OK;
        if (indexed) {
            // the chunk's sun directions are counted behind its copy in, on the copy-in stream - not behind the kernels of the
            // chunks in front - and the host waits for the count: it sizes the evaluation and the copy out
            // (the count is STORED into pinned host memory by the table's last kernel: a 4-byte device -> host copy would queue
            // behind the previous chunk's 50 MB of results on the copy engine, a millisecond per chunk)
            if ((rc = gort_engine_energy_table(p->e, s.d_ang, n, s.d_table, s.h_rows, p->s_in))) return rc;
        }
        PIPE_HIP(hipEventRecord(s.ev_in, p->s_in));
        if (indexed) {
            PIPE_HIP(hipEventSynchronize(s.ev_in));
            s.energy_rows = (long)*static_cast<volatile uint32_t *>(s.h_rows);
            if (s.energy_rows > s.energy_cap) {
                // more distinct rows than the slot has room for: nothing of this slot is in flight (the chunk that used it last has
                // been released), so its buffers can be exchanged for bigger ones here.  The new ones first: a slot whose allocation
                // fails keeps what it had (the chunk fails with GORT_ENOMEM, the chunks in flight and the slot stay whole)
                long cap = 2 * s.energy_cap > s.energy_rows ? 2 * s.energy_cap : s.energy_rows;
                if (cap > p->max_lines) cap = p->max_lines;
                double *h_new = nullptr, *d_new = nullptr;
                const size_t bytes = D * 3 * (size_t)cap * nw;
                const bool mock_failure = ab_env("GORT_PIPE_FAIL_GROW") != nullptr;          // measuring build: the test of this path
                hipError_t err = mock_failure ? hipErrorOutOfMemory : hipHostMalloc((void **)&h_new, bytes, hipHostMallocDefault);
                if (err == hipSuccess && (err = hipMalloc((void **)&d_new, bytes)) != hipSuccess) (void)hipHostFree(h_new);
                if (err != hipSuccess) {
                    (void)hipGetLastError();
                    return fail(err == hipErrorOutOfMemory ? GORT_ENOMEM : GORT_ENODEVICE, "gort_pipe_submit: %ld distinct sun directions in "
                                "one chunk need %zu bytes of row buffers twice (pinned and device): %s", s.energy_rows, bytes, hipGetErrorString(err));
                }
                PIPE_HIP(hipHostFree(s.h_energy));
                s.h_energy = h_new;
                PIPE_HIP(hipFree(s.d_energy));
                s.d_energy = d_new;
                s.energy_cap = cap;
            }
        }
        PIPE_HIP(hipStreamWaitEvent(ks, s.ev_in, 0));
        if (nw > 0 && s.d_rsurf) rc = gort_rsurf_stream_dev(p->e, s.d_ang, n, s.d_rsurf, s.d_scomp, s.d_K);
        else if (nw == 0 && s.d_K) rc = gort_rsurf_stream_dev(p->e, s.d_ang, n, nullptr, nullptr, s.d_K);   // proportions only
        if (rc == GORT_OK && indexed) rc = gort_engine_energy_rows(p->e, s.d_ang, n, s.d_table, s.energy_rows, s.d_energy);
        else if (rc == GORT_OK && s.d_energy && nw > 0) rc = gort_energy_stream_dev(p->e, s.d_ang, n, s.d_energy);
        if (rc) return rc;
        PIPE_HIP(hipEventRecord(s.ev_k, ks));
        PIPE_HIP(hipStreamWaitEvent(p->s_out, s.ev_k, 0));
        if (nw == 0 && s.d_K) PIPE_HIP(hipMemcpyAsync(s.h_K, s.d_K, D * 4 * nn, hipMemcpyDeviceToHost, p->s_out));
        if (nw > 0) {
            if (s.d_rsurf) {
                PIPE_HIP(hipMemcpyAsync(s.h_rsurf, s.d_rsurf, D * nn * nw, hipMemcpyDeviceToHost, p->s_out));
                PIPE_HIP(hipMemcpyAsync(s.h_K, s.d_K, D * 4 * nn, hipMemcpyDeviceToHost, p->s_out));
            }
            if (s.d_scomp) PIPE_HIP(hipMemcpyAsync(s.h_scomp, s.d_scomp, D * 4 * nn * nw, hipMemcpyDeviceToHost, p->s_out));
            if (s.d_energy && s.energy_rows > 0)
                PIPE_HIP(hipMemcpyAsync(s.h_energy, s.d_energy, D * 3 * (size_t)s.energy_rows * nw, hipMemcpyDeviceToHost, p->s_out));
            if (indexed)
                PIPE_HIP(hipMemcpyAsync(s.h_index, energy_table_index(s.d_table, n), sizeof(uint32_t) * nn, hipMemcpyDeviceToHost, p->s_out));
        }
        PIPE_HIP(hipEventRecord(s.ev_out, p->s_out));
        return GORT_OK;
    };
    s.rc = enqueue();
    if (s.rc) s.err = gort_last_error();         // the cause, for gort_pipe_wait (another thread, another last-error slot)
    {
        std::lock_guard<std::mutex> lk(p->mu);
        ++p->submitted;
    }
    p->cv.notify_all();
    return s.rc;
}

extern "C" int gort_pipe_wait(gort_pipe *p, gort_pipe_chunk *out)
{
    if (!p || !out) return fail(GORT_EINVAL, "gort_pipe_wait: bad argument");
    Slot *sp;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        if (p->waited != p->released) return fail(GORT_EINVAL, "gort_pipe_wait: the previous chunk was not released");
        p->cv.wait(lk, [&] { return p->submitted > p->waited; });
        sp = &p->slots[(size_t)(p->waited % p->depth)];
        ++p->waited;
    }
    Slot &s = *sp;
    out->n = s.n;
    out->angles = s.h_ang;
    out->rsurf = s.h_rsurf;
    out->scomp = s.h_scomp;
    out->K = s.h_K;
    out->energy = s.h_energy;
    out->energy_index = s.d_table && p->nw > 0 ? s.h_index : nullptr;
    out->energy_rows = s.energy_rows;
    (void)hipSetDevice(p->device);
    if (s.rc) {
        // part of the chunk may have been queued before the failure (the copy in, some kernels): nothing of it may
        // still be running when the caller releases the slot and its pinned and device buffers are handed out again
        (void)hipStreamSynchronize(p->s_in);
        (void)gort_engine_synchronize(p->e);
        (void)hipStreamSynchronize(p->s_out);
        (void)hipGetLastError();
        return fail(s.rc, "gort_pipe_wait: the chunk failed when it was submitted: %s", s.err.c_str());
    }
    PIPE_HIP(hipEventSynchronize(s.ev_out));
    return GORT_OK;
}

extern "C" int gort_pipe_release(gort_pipe *p)
{
    if (!p) return fail(GORT_EINVAL, "gort_pipe_release: null pipe");
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (p->waited != p->released + 1) return fail(GORT_EINVAL, "gort_pipe_release: no chunk to release");
        ++p->released;
    }
    p->cv.notify_all();
    return GORT_OK;
}
