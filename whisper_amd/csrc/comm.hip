// Multi-GPU combine behind the C ABI: one context per GPU (one process or thread each), RCCL over xGMI.
//
// replaces: the merge of the 32 partial aggregates (server-batch-prio3/src/main.rs:327-335, AggregateShare::merge,
// vdaf.rs:658-676) when the batch is sharded over the GPUs of one server, and the element-wise sum of per-GPU sketches
// (server-hh/src/main.rs:374-394 does the same between the two servers with CountSketch::insert).
//
// Field128 partial aggregates cannot use RCCL's sum (it is not mod p): the ranks all-gather the raw 16-byte limbs and every
// rank sums the rows mod p on its GPU (wh_f128_sum_rows_dev).  Heavy-hitter counters live in Z/2^32, where ncclSum on
// ncclUint32 is exact.  RCCL is bound at run time (dlopen), so a single-GPU server never loads it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include "../../include/whisper_gfx950.h"

namespace wh {
int ctx_device(wh_ctx *c);
hipStream_t ctx_stream(wh_ctx *c);
void **ctx_comm_slot(wh_ctx *c);
std::recursive_mutex &ctx_mutex(wh_ctx *c);
void ctx_set_error(wh_ctx *c, const std::string &what);
uint32_t ctx_output_len(wh_ctx *c);
}
using namespace wh;

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string why;       // why it could not be loaded
};

// one binding per process; a copy of RCCL that is already in the process (torch's) is reused.  WH_RCCL_LIBRARY (read once, at
// the first use) names the library file for installations that keep RCCL outside the loader's path; when it is set, only
// that file is tried.
Rccl &rccl_state() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        const char *override_name = getenv("WH_RCCL_LIBRARY");
        const char *const *names = override_name ? &override_name : defaults;
        const int n_names = override_name ? 1 : 4;
        for (int k = 0; k < n_names && !r.handle; k++) r.handle = dlopen(names[k], RTLD_NOW | RTLD_NOLOAD);
        std::string last;
        for (int k = 0; k < n_names && !r.handle; k++) {
            r.handle = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
            if (!r.handle) { const char *e = dlerror(); last = e ? e : "dlopen failed"; }    // dlerror() clears what it returns
        }
        if (!r.handle) { r.why = "RCCL not found: " + last; return; }
#define WH_SYM(field, name)                                                                    \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name));                   \
        if (!r.field && r.why.empty()) r.why = std::string("RCCL lacks ") + name
        WH_SYM(get_unique_id, "ncclGetUniqueId");
        WH_SYM(comm_init_rank, "ncclCommInitRank");
        WH_SYM(comm_destroy, "ncclCommDestroy");
        WH_SYM(all_gather, "ncclAllGather");
        WH_SYM(all_reduce, "ncclAllReduce");
        WH_SYM(error_string, "ncclGetErrorString");
#undef WH_SYM
    });
    return r;
}
Rccl *rccl() { Rccl &r = rccl_state(); return r.why.empty() ? &r : nullptr; }
const char *rccl_why() { return rccl_state().why.c_str(); }

struct Comm {
    ncclComm_t comm = nullptr;   // nullptr: a lone GPU, no RCCL
    int rank = 0, world = 1;
    uint8_t *d_gather = nullptr; uint64_t gather_cap = 0;     // [world][len] rows of the all-gather
    uint8_t *d_part = nullptr; uint64_t part_cap = 0;         // this GPU's partial aggregate (host-buffer entry point)
};

Comm *comm_of(wh_ctx *c) { return static_cast<Comm *>(*ctx_comm_slot(c)); }

int32_t fail_hip(wh_ctx *c, const char *what, hipError_t e) { ctx_set_error(c, std::string(what) + ": " + hipGetErrorString(e)); return WH_ERR_HIP; }
int32_t fail_nccl(wh_ctx *c, const char *what, ncclResult_t e) {
    Rccl *r = rccl();
    ctx_set_error(c, std::string(what) + ": " + (r ? r->error_string(e) : "RCCL error"));
    return WH_ERR_COMM;
}
int32_t grow(wh_ctx *c, uint8_t *&p, uint64_t &cap, uint64_t want) {
    if (want <= cap && p) return WH_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want);
    if (e != hipSuccess) return fail_hip(c, "hipMalloc (gather rows)", e);
    cap = want;
    return WH_OK;
}

}  // namespace

extern "C" {

const char *wh_comm_why(void) { return rccl_why(); }

int32_t wh_comm_unique_id(uint8_t id[WH_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == WH_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id) return WH_ERR_PARAM;
    Rccl *r = rccl();
    if (!r) return WH_ERR_UNSUPPORTED;
    ncclUniqueId u;
    if (r->get_unique_id(&u) != ncclSuccess) return WH_ERR_COMM;
    memcpy(id, &u, sizeof u);
    return WH_OK;
}

int32_t wh_comm_init(wh_ctx *c, const uint8_t *id, int32_t rank, int32_t world) {
    if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    if (comm_of(c)) return WH_ERR_STATE;
    hipError_t e = hipSetDevice(ctx_device(c));
    if (e != hipSuccess) return fail_hip(c, "hipSetDevice", e);
    Comm *m = new Comm();
    m->rank = rank; m->world = world;
    if (id) {        // also for world == 1: the same code path as a node of GPUs, through RCCL
        Rccl *r = rccl();
        if (!r) { delete m; ctx_set_error(c, rccl_why()); return WH_ERR_UNSUPPORTED; }
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        ncclResult_t ne = r->comm_init_rank(&m->comm, world, u, rank);
        if (ne != ncclSuccess) { delete m; return fail_nccl(c, "ncclCommInitRank", ne); }
    }
    *ctx_comm_slot(c) = m;
    return WH_OK;
}

int32_t wh_comm_destroy(wh_ctx *c) {
    if (!c) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    Comm *m = comm_of(c);
    if (!m) return WH_OK;
    (void)hipSetDevice(ctx_device(c));
    (void)hipStreamSynchronize(ctx_stream(c));
    if (m->comm) { Rccl *r = rccl(); if (r) (void)r->comm_destroy(m->comm); }
    if (m->d_gather) (void)hipFree(m->d_gather);
    if (m->d_part) (void)hipFree(m->d_part);
    delete m;
    *ctx_comm_slot(c) = nullptr;
    return WH_OK;
}

int32_t wh_comm_rank(wh_ctx *c, int32_t *rank, int32_t *world) {
    if (!c || !rank || !world) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    Comm *m = comm_of(c);
    *rank = m ? m->rank : 0; *world = m ? m->world : 1;
    return WH_OK;
}

int32_t wh_f128_allgather_reduce_dev(wh_ctx *c, const uint8_t *d_partial, uint32_t len, uint8_t *d_out) {
    if (!c || !d_partial || !d_out || !len) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    Comm *m = comm_of(c);
    if (!m) return WH_ERR_STATE;
    hipError_t e = hipSetDevice(ctx_device(c));
    if (e != hipSuccess) return fail_hip(c, "hipSetDevice", e);
    const uint64_t row = 16ull * len;
    if (!m->comm) return wh_f128_sum_rows_dev(c, d_partial, 1, len, nullptr, d_out);     // a lone GPU: the sum of one row
    int32_t rc;
    if ((rc = grow(c, m->d_gather, m->gather_cap, row * m->world))) return rc;
    // raw limbs: RCCL's own sum is not mod p
    ncclResult_t ne = rccl()->all_gather(d_partial, m->d_gather, row, ncclUint8, m->comm, ctx_stream(c));
    if (ne != ncclSuccess) return fail_nccl(c, "ncclAllGather", ne);
    return wh_f128_sum_rows_dev(c, m->d_gather, (uint64_t)m->world, len, nullptr, d_out);
}

int32_t wh_prio3_allgather_reduce_dev(wh_ctx *c, const uint8_t *d_partial, uint8_t *d_agg) {
    if (!c) return WH_ERR_PARAM;
    return wh_f128_allgather_reduce_dev(c, d_partial, ctx_output_len(c), d_agg);
}

int32_t wh_prio3_allgather_reduce(wh_ctx *c, const uint8_t *keep, uint8_t *agg) {
    if (!c || !agg) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    Comm *m = comm_of(c);
    if (!m) return WH_ERR_STATE;
    hipError_t e = hipSetDevice(ctx_device(c));
    if (e != hipSuccess) return fail_hip(c, "hipSetDevice", e);
    const uint64_t row = 16ull * ctx_output_len(c);
    int32_t rc;
    if ((rc = grow(c, m->d_part, m->part_cap, 2 * row))) return rc;
    // this GPU's masked aggregate (host keep mask), left on the device
    uint8_t *d_part = m->d_part, *d_all = m->d_part + row;
    if ((rc = wh_prio3_aggregate(c, keep, agg))) return rc;           // also validates the context's state
    if ((e = hipMemcpyAsync(d_part, agg, row, hipMemcpyHostToDevice, ctx_stream(c))) != hipSuccess) return fail_hip(c, "hipMemcpyAsync", e);
    if ((rc = wh_prio3_allgather_reduce_dev(c, d_part, d_all))) return rc;
    if ((e = hipMemcpyAsync(agg, d_all, row, hipMemcpyDeviceToHost, ctx_stream(c))) != hipSuccess) return fail_hip(c, "hipMemcpyAsync", e);
    if ((e = hipStreamSynchronize(ctx_stream(c))) != hipSuccess) return fail_hip(c, "hipStreamSynchronize", e);
    return WH_OK;
}

int32_t wh_hh_allreduce_u32_dev(wh_ctx *c, uint32_t *d_sketch, uint64_t words) {
    if (!c || !d_sketch || !words) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    Comm *m = comm_of(c);
    if (!m) return WH_ERR_STATE;
    if (!m->comm) return WH_OK;                       // a lone GPU: its sketch is the sum
    hipError_t e = hipSetDevice(ctx_device(c));
    if (e != hipSuccess) return fail_hip(c, "hipSetDevice", e);
    // counters are Z/2^32: the integer sum wraps exactly as u32::wrapping_add does (common/src/group.rs:88-91)
    ncclResult_t ne = rccl()->all_reduce(d_sketch, d_sketch, words, ncclUint32, ncclSum, m->comm, ctx_stream(c));
    if (ne != ncclSuccess) return fail_nccl(c, "ncclAllReduce", ne);
    return WH_OK;
}

}  // extern "C"
