// gort_rccl.cpp -- the one exchange step of the multi-GPU path in the C ABI: an in-place RCCL all-gather of a
// row-sharded LUT over xGMI (`ncclAllGather`, one exchange, no reductions), plus what a C host needs to
// get a communicator without any Python: the unique id rank 0 hands to the others through whatever bootstrap the host
// has (MPI, a socket, torch.distributed), ncclCommInitRank, and ncclCommInitAll for one process driving several devices.
// The reference has no counterpart (single process, README.md:30-36).
//
// librccl is bound at run time (dlopen), not at link time: single-GPU users need no RCCL at all, and a process that has
// PyTorch loaded already holds ITS copy of librccl.so - the one already in the process is the one used, so that there
// are never two RCCL runtimes behind one set of devices.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "gort_internal.h"

namespace gort {
namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy already in the process first (RTLD_NOLOAD), then the usual names
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        for (const char *n : names) {
            if (r.lib) break;
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) {
            const char *e = dlerror();
            std::snprintf(r.why, sizeof r.why, "librccl.so not found (%s)", e ? e : "dlopen failed");
            return;
        }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
            std::snprintf(r.why, sizeof r.why, "librccl.so lacks an expected symbol");
            r.lib = nullptr;
        }
    });
    return r;
}

int need_rccl(const char *who)
{
    if (!rccl().lib) return fail(GORT_ENODEVICE, "%s: %s", who, rccl().why);
    return GORT_OK;
}

int check(ncclResult_t rc, const char *who)
{
    if (rc == ncclSuccess) return GORT_OK;
    return fail(GORT_ENODEVICE, "%s: RCCL: %s", who, rccl().GetErrorString(rc));
}

}  // namespace
}  // namespace gort

using namespace gort;

static_assert(GORT_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as GORT_RCCL_ID_BYTES opaque bytes");

extern "C" int gort_rccl_unique_id(unsigned char *id)
{
    if (!id) return fail(GORT_EINVAL, "gort_rccl_unique_id: null id");
    int rc = need_rccl("gort_rccl_unique_id");
    if (rc) return rc;
    ncclUniqueId u;
    if ((rc = check(rccl().GetUniqueId(&u), "ncclGetUniqueId"))) return rc;
    std::memcpy(id, u.internal, GORT_RCCL_ID_BYTES);
    return GORT_OK;
}

extern "C" int gort_rccl_comm_init_rank(int world, const unsigned char *id, int rank, void **comm)
{
    if (comm) *comm = nullptr;
    if (!id || !comm || world < 1 || rank < 0 || rank >= world) return fail(GORT_EINVAL, "gort_rccl_comm_init_rank: bad argument");
    int rc = need_rccl("gort_rccl_comm_init_rank");
    if (rc) return rc;
    ncclUniqueId u;
    std::memcpy(u.internal, id, GORT_RCCL_ID_BYTES);
    ncclComm_t c = nullptr;
    if ((rc = check(rccl().CommInitRank(&c, world, u, rank), "ncclCommInitRank"))) return rc;
    *comm = c;
    return GORT_OK;
}

extern "C" int gort_rccl_comm_init_all(int n_devices, const int *devices, void **comms)
{
    if (!comms || n_devices < 1) return fail(GORT_EINVAL, "gort_rccl_comm_init_all: bad argument");
    int rc = need_rccl("gort_rccl_comm_init_all");
    if (rc) return rc;
    int current = 0;
    (void)hipGetDevice(&current);                       // ncclCommInitAll walks the devices
    rc = check(rccl().CommInitAll(reinterpret_cast<ncclComm_t *>(comms), n_devices, devices), "ncclCommInitAll");
    (void)hipSetDevice(current);
    return rc;
}

extern "C" int gort_rccl_comm_destroy(void *comm)
{
    if (!comm) return GORT_OK;
    int rc = need_rccl("gort_rccl_comm_destroy");
    if (rc) return rc;
    return check(rccl().CommDestroy(static_cast<ncclComm_t>(comm)), "ncclCommDestroy");
}

// In place: rank r has written rows [r * rows_per_rank, (r + 1) * rows_per_rank) of lut_dev, which holds
// world * rows_per_rank rows of row_bytes bytes (gort_lut_alloc of the gatherable size); its window is the send buffer.
extern "C" int gort_lut_allgather_on(void *stream, void *lut_dev, size_t rows_per_rank, size_t row_bytes, int rank, int world, void *comm)
{
    if (!lut_dev || !comm || world < 1 || rank < 0 || rank >= world) return fail(GORT_EINVAL, "gort_lut_allgather: bad argument");
    int rc = need_rccl("gort_lut_allgather");
    if (rc) return rc;
    const size_t bytes = rows_per_rank * row_bytes;
    if (bytes == 0) return GORT_OK;
    char *base = static_cast<char *>(lut_dev);
    // doubles where the window is made of them (it is: LUT rows), bytes otherwise
    const bool dbl = bytes % sizeof(double) == 0 && reinterpret_cast<uintptr_t>(lut_dev) % sizeof(double) == 0;
    return check(rccl().AllGather(base + (size_t)rank * bytes, base, dbl ? bytes / sizeof(double) : bytes, dbl ? ncclDouble : ncclChar,
                                  static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)),
                 "ncclAllGather");
}
