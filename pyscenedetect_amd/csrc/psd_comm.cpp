// psd_comm.cpp -- the one exchange step of the multi-GPU path behind the C-ABI: an RCCL all-gather of per-frame score
// records.  The pixel work shards with no data-path collective (clips are independent; a frame range
// plus a one-frame halo is self-contained); what every rank needs afterwards is every clip's records (<= 1064 B per
// frame) to run the deterministic epilogues, so cut lists do not depend on the GPU count.  Records are KBs to a few MBs:
// the exchange is latency-bound, one fused ncclAllGather of padded per-rank blocks over the xGMI mesh.
//
// RCCL is loaded at run time (dlopen): a host that already carries one (PyTorch bundles its own librccl.so) keeps
// exactly one copy in the process; the library itself does not link against RCCL, so single-GPU hosts need none.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "psd_internal.h"

extern "C" void psd_set_error(const char* fmt, ...);

namespace psd {
int engine_device(psd_engine* e);
hipStream_t engine_stream(psd_engine* e);
}  // namespace psd

namespace {

struct NcclUniqueId { char internal[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* NcclComm;
typedef int (*fn_get_unique_id)(NcclUniqueId*);
typedef int (*fn_comm_init_rank)(NcclComm*, int, NcclUniqueId, int);
typedef int (*fn_all_gather)(const void*, void*, size_t, int /*ncclDataType_t*/, NcclComm, hipStream_t);
typedef int (*fn_comm_destroy)(NcclComm);
typedef const char* (*fn_error_string)(int);

struct Rccl {
    void* handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_error_string error_string = nullptr;
};

std::string g_rccl_why;   // why loading failed (dlerror() captured ONCE: a second call returns NULL)

void rccl_load(Rccl& r)
{
    // a copy that is already in the process first (RTLD_NOLOAD), then the system's
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)
        if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* p : paths)
        if (!r.handle) {
            r.handle = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
            if (!r.handle) { const char* why = dlerror(); g_rccl_why = why ? why : "not found"; }
        }
    if (!r.handle) return;
    r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
    r.all_gather = (fn_all_gather)dlsym(r.handle, "ncclAllGather");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
    r.error_string = (fn_error_string)dlsym(r.handle, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy) {
        g_rccl_why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
        dlclose(r.handle);
        r.handle = nullptr;
    }
}

Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_load(r); });
    return r.handle ? &r : nullptr;
}

int rccl_fail(Rccl* r, const char* what, int rc)
{
    psd_set_error("%s failed: %s", what, r && r->error_string ? r->error_string(rc) : "RCCL error");
    return PSD_ERR_HIP;
}

}  // namespace

struct psd_comm {
    psd_engine* engine = nullptr;
    NcclComm comm = nullptr;
    int n_ranks = 0, rank = 0;
    uint8_t* d_send = nullptr;   // cap records
    uint8_t* d_recv = nullptr;   // n_ranks * cap records
    uint8_t* h_recv = nullptr;   // pinned mirror of d_recv
    size_t cap = 0;
};

#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            psd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            (void)hipGetLastError(); /* (the failure is reported here: do not leave it for the next launch check) */ \
            return PSD_ERR_HIP;                                                                      \
        }                                                                                            \
    } while (0)

extern "C" {

int psd_comm_unique_id(void* id128)
{
    if (!id128) { psd_set_error("psd_comm_unique_id: null argument"); return PSD_ERR_INVALID; }
    Rccl* r = rccl();
    if (!r) { psd_set_error("RCCL (librccl.so) is not available: %s", g_rccl_why.c_str()); return PSD_ERR_UNSUPPORTED; }
    NcclUniqueId id;
    const int rc = r->get_unique_id(&id);
    if (rc != 0) return rccl_fail(r, "ncclGetUniqueId", rc);
    memcpy(id128, &id, sizeof id);
    return PSD_OK;
}

int psd_comm_create(psd_engine* e, int n_ranks, int rank, const void* id128, psd_comm** out)
{
    if (!e || !id128 || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        psd_set_error("psd_comm_create: invalid argument");
        return PSD_ERR_INVALID;
    }
    *out = nullptr;
    Rccl* r = rccl();
    if (!r) { psd_set_error("RCCL (librccl.so) is not available: %s", g_rccl_why.c_str()); return PSD_ERR_UNSUPPORTED; }
    HIP_TRY(hipSetDevice(psd::engine_device(e)));
    psd_comm* c = new (std::nothrow) psd_comm();
    if (!c) { psd_set_error("out of memory"); return PSD_ERR_NOMEM; }
    c->engine = e; c->n_ranks = n_ranks; c->rank = rank;
    NcclUniqueId id;
    memcpy(&id, id128, sizeof id);
    const int rc = r->comm_init_rank(&c->comm, n_ranks, id, rank);
    if (rc != 0) { delete c; return rccl_fail(r, "ncclCommInitRank", rc); }
    *out = c;
    return PSD_OK;
}

void psd_comm_destroy(psd_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(psd::engine_device(c->engine));
    (void)hipStreamSynchronize(psd::engine_stream(c->engine));
    Rccl* r = rccl();
    if (r && c->comm) (void)r->comm_destroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    delete c;
}

int psd_allgather_scores(psd_comm* c, const psd_frame_scores* d_local, int n_local, const int* counts, psd_frame_scores* h_all)
{
    // Argument errors that only THIS rank can see must not keep it out of the collective (the other ranks would hang in
    // ncclAllGather): as long as the communicator and the counts are usable the rank takes part -- with what it has,
    // zero-filled -- and reports its error afterwards.
    if (!c || !counts) { psd_set_error("psd_allgather_scores: invalid argument"); return PSD_ERR_INVALID; }
    bool local_error = false;
    char local_msg[160] = "";
    if (n_local < 0 || (n_local > 0 && !d_local)) {
        snprintf(local_msg, sizeof local_msg, "psd_allgather_scores: invalid local records (n_local = %d)", n_local);
        local_error = true; n_local = 0;
    } else if (counts[c->rank] != n_local) {
        snprintf(local_msg, sizeof local_msg, "psd_allgather_scores: counts[%d] = %d but this rank contributes %d records",
                 c->rank, counts[c->rank], n_local);
        local_error = true;
        if (counts[c->rank] >= 0 && n_local > counts[c->rank]) n_local = counts[c->rank];
    }
    Rccl* r = rccl();
    if (!r) { psd_set_error("RCCL (librccl.so) is not available: %s", g_rccl_why.c_str()); return PSD_ERR_UNSUPPORTED; }
    size_t cap = 1, total = 0;
    for (int i = 0; i < c->n_ranks; i++) {
        if (counts[i] < 0) { psd_set_error("psd_allgather_scores: negative count"); return PSD_ERR_INVALID; }
        if ((size_t)counts[i] > cap) cap = (size_t)counts[i];
        total += (size_t)counts[i];
    }
    if (total > 0 && !h_all) { psd_set_error("psd_allgather_scores: null output"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(psd::engine_device(c->engine)));
    hipStream_t stream = psd::engine_stream(c->engine);
    const size_t rec = sizeof(psd_frame_scores);
    if (c->cap < cap) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (c->d_send) HIP_TRY(hipFree(c->d_send));
        if (c->d_recv) HIP_TRY(hipFree(c->d_recv));
        if (c->h_recv) HIP_TRY(hipHostFree(c->h_recv));
        c->d_send = c->d_recv = c->h_recv = nullptr; c->cap = 0;
        HIP_TRY(hipMalloc((void**)&c->d_send, cap * rec));
        HIP_TRY(hipMalloc((void**)&c->d_recv, cap * rec * c->n_ranks));
        HIP_TRY(hipHostMalloc((void**)&c->h_recv, cap * rec * c->n_ranks, hipHostMallocDefault));
        c->cap = cap;
    }
    // ragged blocks padded to the largest: one fused collective (the point-to-point xGMI mesh makes a ring pay a hop per
    // rank; at these sizes latency is everything)
    if (local_error) HIP_TRY(hipMemsetAsync(c->d_send, 0, c->cap * rec, stream));
    if (n_local > 0) HIP_TRY(hipMemcpyAsync(c->d_send, d_local, (size_t)n_local * rec, hipMemcpyDeviceToDevice, stream));
    const int rc = r->all_gather(c->d_send, c->d_recv, c->cap * rec, 0 /* ncclChar */, c->comm, stream);
    if (rc != 0) return rccl_fail(r, "ncclAllGather", rc);
    HIP_TRY(hipMemcpyAsync(c->h_recv, c->d_recv, c->cap * rec * c->n_ranks, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (local_error) { psd_set_error("%s (the collective was completed with zero-filled records)", local_msg); return PSD_ERR_INVALID; }
    size_t off = 0;
    for (int i = 0; i < c->n_ranks; i++) {
        memcpy(h_all + off, c->h_recv + (size_t)i * c->cap * rec, (size_t)counts[i] * rec);
        off += (size_t)counts[i];
    }
    return PSD_OK;
}

// The same exchange for records that are already on the HOST -- what a rank holds after a corpus pass: its clips were scored in
// several submissions (one per resolution, pieces of long runs) and collected as they finished, so no single device buffer has
// them.  n_local elements of elem_bytes each (40: psd_frame_sums; 1064: psd_frame_scores) go through a page-locked staging
// buffer to the device, ONE ncclAllGather of padded per-rank blocks, and back; h_all receives counts[r] elements of every rank
// r in rank order.  counts must be the same array on every rank -- in the sharded flow every rank derives it from the plan
// (distributed.assign_clips is deterministic and every rank knows every clip's length), so no collective is spent on it.
int psd_allgather_host(psd_comm* c, const void* h_local, int n_local, size_t elem_bytes, const int* counts, void* h_all)
{
    if (!c || !counts || elem_bytes == 0) { psd_set_error("psd_allgather_host: invalid argument"); return PSD_ERR_INVALID; }
    bool local_error = false;
    char local_msg[160] = "";
    if (n_local < 0 || (n_local > 0 && !h_local)) {
        snprintf(local_msg, sizeof local_msg, "psd_allgather_host: invalid local records (n_local = %d)", n_local);
        local_error = true; n_local = 0;
    } else if (counts[c->rank] != n_local) {
        snprintf(local_msg, sizeof local_msg, "psd_allgather_host: counts[%d] = %d but this rank contributes %d records",
                 c->rank, counts[c->rank], n_local);
        local_error = true;
        if (counts[c->rank] >= 0 && n_local > counts[c->rank]) n_local = counts[c->rank];
    }
    Rccl* r = rccl();
    if (!r) { psd_set_error("RCCL (librccl.so) is not available: %s", g_rccl_why.c_str()); return PSD_ERR_UNSUPPORTED; }
    size_t cap_el = 1, total = 0;
    for (int i = 0; i < c->n_ranks; i++) {
        if (counts[i] < 0) { psd_set_error("psd_allgather_host: negative count"); return PSD_ERR_INVALID; }
        if ((size_t)counts[i] > cap_el) cap_el = (size_t)counts[i];
        total += (size_t)counts[i];
    }
    if (total > 0 && !h_all) { psd_set_error("psd_allgather_host: null output"); return PSD_ERR_INVALID; }
    HIP_TRY(hipSetDevice(psd::engine_device(c->engine)));
    hipStream_t stream = psd::engine_stream(c->engine);
    // (the buffers are sized in records of the widest kind: a block of cap_el elements fits cap records whenever
    //  cap * sizeof(psd_frame_scores) >= cap_el * elem_bytes)
    const size_t rec = sizeof(psd_frame_scores);
    const size_t block = cap_el * elem_bytes;
    const size_t cap = (block + rec - 1) / rec;
    if (c->cap < cap) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (c->d_send) HIP_TRY(hipFree(c->d_send));
        if (c->d_recv) HIP_TRY(hipFree(c->d_recv));
        if (c->h_recv) HIP_TRY(hipHostFree(c->h_recv));
        c->d_send = c->d_recv = c->h_recv = nullptr; c->cap = 0;
        HIP_TRY(hipMalloc((void**)&c->d_send, cap * rec));
        HIP_TRY(hipMalloc((void**)&c->d_recv, cap * rec * c->n_ranks));
        HIP_TRY(hipHostMalloc((void**)&c->h_recv, cap * rec * c->n_ranks, hipHostMallocDefault));
        c->cap = cap;
    }
    // this rank's block travels through the front of the pinned mirror (it is overwritten by the gathered blocks afterwards)
    memset(c->h_recv, 0, block);
    if (n_local > 0) memcpy(c->h_recv, h_local, (size_t)n_local * elem_bytes);
    HIP_TRY(hipMemcpyAsync(c->d_send, c->h_recv, block, hipMemcpyHostToDevice, stream));
    const int rc = r->all_gather(c->d_send, c->d_recv, block, 0 /* ncclChar */, c->comm, stream);
    if (rc != 0) return rccl_fail(r, "ncclAllGather", rc);
    HIP_TRY(hipMemcpyAsync(c->h_recv, c->d_recv, block * c->n_ranks, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (local_error) { psd_set_error("%s (the collective was completed with zero-filled records)", local_msg); return PSD_ERR_INVALID; }
    size_t off = 0;
    for (int i = 0; i < c->n_ranks; i++) {
        memcpy((uint8_t*)h_all + off, c->h_recv + (size_t)i * block, (size_t)counts[i] * elem_bytes);
        off += (size_t)counts[i] * elem_bytes;
    }
    return PSD_OK;
}

}  // extern "C"
