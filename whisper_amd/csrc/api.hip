// C-ABI layer of libwhisper_gfx950.so (include/whisper_gfx950.h): context, HBM workspace,
// constant tables and kernel launches.  One context = one GPU = one HIP stream.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <mutex>
#include <thread>
#include "host_math.hpp"
#include "aes_host.hpp"
#include "hh_dev.hpp"
#include "prove_dev.hpp"
#include "ntt_dev.hpp"
#include "shard_dev.hpp"
#include "sort_dev.hpp"
#include "kernels.hip"

using namespace wh;

// Scratch of wh_sort_by_testing_id_noalloc_dev, the one entry point that allocates nothing: the second permutation
// buffer (SORT_NOALLOC_MAX_N words) followed by the count table of the largest sort (256 bins x tiles, then 256 bin
// totals).  In host code the name is only a handle: the address on a device comes from hipGetSymbolAddress, once per
// device (scratch_of_device below), and every use goes through that pointer.
constexpr uint64_t SORT_NOALLOC_MAX_N = 1ull << 20;
constexpr uint64_t SORT_NOALLOC_HIST_WORDS = 256 * (SORT_NOALLOC_MAX_N / SORT_TILE) + 256;
static __device__ __align__(256) uint32_t scratch[SORT_NOALLOC_MAX_N + SORT_NOALLOC_HIST_WORDS];

namespace wh {
void launch_expand(const Prio3Dev &P, const Workspace &W, uint32_t npad, uint32_t sel, hipStream_t stream);
void launch_meas(const Prio3Dev &P, const Workspace &W, uint32_t mode, uint32_t npad, int32_t *status, hipStream_t stream);
void launch_chain(const Prio3Dev &P, const Workspace &W, uint32_t npad, bool prefetch, uint64_t vk0, uint64_t vk1, int32_t *status,
                  hipStream_t stream);
}

// Tuning and test knobs of a context (wh_ctx_set_option).  The environment variables of the same purpose are read ONCE, in
// wh_ctx_create, as the initial values: no entry point looks at the environment afterwards.
struct Knobs {
    uint64_t v[WH_NUM_OPTS] = {0};
};
static void knobs_from_env(Knobs &k) {
    k.v[WH_OPT_MAX_SUBBATCH] = 0;                 // 0: the library's choice (pick_cap)
    k.v[WH_OPT_HOST_CHUNK_MB] = 2048;
    k.v[WH_OPT_HOST_COPY_THREADS] = 0;            // 0: hardware concurrency, at most 8
    k.v[WH_OPT_HH_SUBBATCH] = 0;
    k.v[WH_OPT_HH_SUM_LDS] = 120u << 10;
    k.v[WH_OPT_SIDE_STREAM] = 1;
    k.v[WH_OPT_HASH_B_SIDE] = 1;
    k.v[WH_OPT_THIRD_STREAM] = 1;
    k.v[WH_OPT_BASE_PAIRED] = 1;
    k.v[WH_OPT_CHAIN_MAX_INPUT_LEN] = WH_CHAIN_MAX_INPUT_LEN_DEFAULT;
    k.v[WH_OPT_CHAIN_PREFETCH] = 1;                // 3 waves per SIMD with the next block in flight: 1-7 % ahead of 4 without (profiles/r3_chain_experiments.txt)
    k.v[WH_OPT_QUERY_TINY] = 1;
    auto num = [](const char *name, uint64_t &out) { if (const char *e = getenv(name)) out = strtoull(e, nullptr, 10); };
    num("WH_MAX_SUBBATCH", k.v[WH_OPT_MAX_SUBBATCH]);
    num("WH_HOST_CHUNK_MB", k.v[WH_OPT_HOST_CHUNK_MB]);
    num("WH_HOST_COPY_THREADS", k.v[WH_OPT_HOST_COPY_THREADS]);
    num("WH_HH_SUBBATCH", k.v[WH_OPT_HH_SUBBATCH]);
    num("WH_HH_SUM_LDS", k.v[WH_OPT_HH_SUM_LDS]);
    num("WH_CHAIN_MAX_INPUT_LEN", k.v[WH_OPT_CHAIN_MAX_INPUT_LEN]);
    num("WH_CHAIN_PREFETCH", k.v[WH_OPT_CHAIN_PREFETCH]);
    num("WH_QUERY_TINY", k.v[WH_OPT_QUERY_TINY]);
    if (getenv("WH_NO_SIDE_STREAM")) k.v[WH_OPT_SIDE_STREAM] = 0;
    if (getenv("WH_HASH_B_MAIN")) k.v[WH_OPT_HASH_B_SIDE] = 0;
    if (getenv("WH_NO_THIRD_STREAM")) k.v[WH_OPT_THIRD_STREAM] = 0;
    if (getenv("WH_BASE_NO_MEAS")) k.v[WH_OPT_BASE_PAIRED] = 0;
}

struct wh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // side stream for the short kernels that do not depend on the measurement sponges (proof-share
    // expansion, public-parts hash / RLC / testing id): they slip in beside k_meas
    hipStream_t stream2 = nullptr, stream3 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr;
    // the sort by testing id beside the sponge kernels (wh_prio3_prepare_sorted_batch_dev)
    hipStream_t stream_sort = nullptr;
    hipEvent_t ev_ids = nullptr, ev_sorted = nullptr;
    // host-buffer entry points: copy stream, "chunk b has arrived" / "chunk b has been verified" events, pinned staging
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    uint8_t *h_stage = nullptr; uint64_t h_stage_cap = 0;
    wh_prio3_params params{};
    HostSizes z{};
    Prio3Dev P{};
    Workspace W{};
    uint64_t cap = 0;            // reports the workspace holds (one sub-batch)
    // buffers owned by the context
    F128 *d_omega = nullptr, *d_stab = nullptr;
    uint8_t *d_aes = nullptr;    // 176 B round keys + 256 B sbox
    uint8_t d_aes_key[16] = {0};
    bool aes_valid = false;
    F128 *d_out_shares = nullptr;  // [out_cap][output_len]
    uint64_t out_cap = 0, n_last = 0;
    // which entry point produced the out shares the context holds: aggregate / out_shares / prepare_finish refuse anything else
    enum Producer { NONE = 0, PREPARE_BATCH, PREPARE_INIT, FLP_QUERY } producer = NONE;
    bool init_finished = false;  // PREPARE_INIT: wh_prio3_prepare_finish_batch has decided the batch (prepare_next, prio3.rs:2396-2442)
    int32_t *d_status_last = nullptr;  // status of the last batch: always context-owned memory (d_status_keep or d_status_own)
    int32_t *d_status_keep = nullptr; uint64_t status_keep_cap = 0;   // copy of a caller-owned status array (the _dev entry points)
    int32_t *d_status_own = nullptr; uint64_t status_cap = 0;
    uint8_t *d_keep_eff = nullptr; uint64_t keep_cap = 0;
    F128 *d_partials = nullptr; uint64_t partials_cap = 0;
    // staging for the host-buffer entry points
    uint8_t *d_blob = nullptr; uint64_t blob_cap = 0;
    uint64_t *d_offsets = nullptr; uint64_t off_cap = 0;
    uint8_t *d_tags = nullptr, *d_ids = nullptr; uint64_t tags_cap = 0, ids_cap = 0;
    uint8_t *d_prep = nullptr; uint64_t prep_cap = 0;
    F128 *d_jr_seeds = nullptr; uint64_t jr_seeds_cap = 0;   // prepare_init's joint-rand seed per report (prepare state)
    // optional per-kernel timing with HIP events on the context stream (wh_ctx_set_profiling)
    bool profiling = false;
    struct Ev { hipEvent_t a, b; int id; };
    std::vector<Ev> evs;
    std::vector<hipEvent_t> ev_pool;
    // prover (prove.hip): transform tables, built on first use
    F128 *d_omega2m = nullptr, *d_twist = nullptr;
    // heavy-hitter key chains (vdpf.hip)
    HhTables *d_hh_tab = nullptr; bool hh_tab_valid = false; uint8_t hh_vk[16] = {0};
    HhWork hh_w{}; uint64_t hh_cap = 0; uint32_t hh_cap_depth = 0;
    uint32_t *d_hh_dat = nullptr; uint64_t hh_dat_cap = 0;      // [n][2^depth][258] of the last batch, when the caller gave no buffer
    const uint32_t *d_hh_dat_last = nullptr; uint64_t hh_n_last = 0; uint32_t hh_depth_last = 0;
    uint8_t *d_hh_keys = nullptr; uint64_t hh_keys_cap = 0;
    uint8_t *d_hh_tags = nullptr; uint64_t hh_tags_cap = 0;
    uint8_t *d_hh_ids = nullptr; uint64_t hh_ids_cap = 0;
    int32_t *d_hh_status = nullptr; uint64_t hh_status_cap = 0;
    uint8_t *d_hh_keep = nullptr; uint64_t hh_keep_cap = 0;
    uint32_t *d_hh_acc = nullptr; uint64_t hh_acc_cap = 0;
    uint32_t *d_hh_part = nullptr; uint64_t hh_part_cap = 0;   // one sub-batch of bucket vectors on their way into sketches (wh_hh_eval_and_sketch_batch_dev)
    // testing-id sort (sort.hip)
    uint32_t *d_sort_tmp = nullptr; uint64_t sort_tmp_cap = 0;
    uint32_t *d_sort_hist = nullptr; uint64_t sort_hist_cap = 0;
    uint64_t *d_bad_ids = nullptr; uint64_t bad_ids_cap = 0;
    hipEvent_t ev_bad = nullptr; bool bad_sorted_busy = false;
    unsigned __int128 *h_bad = nullptr; uint64_t h_bad_cap = 0;   // pinned host staging of wh_keep_mask_dev: lives until the next call, so its copy stays asynchronous
    // scratch of the host-buffer entry points (grow on demand, freed with the context)
    uint8_t *d_tmp_a = nullptr; uint64_t tmp_a_cap = 0;
    uint8_t *d_tmp_b = nullptr; uint64_t tmp_b_cap = 0;
    uint8_t *d_tmp_c = nullptr; uint64_t tmp_c_cap = 0;
    uint32_t *d_scratch = nullptr;      // this device's copy of the file-scope `scratch` array (not owned)
    std::string scratch_err;            // why d_scratch is empty, if it is
    // every entry point holds this while it works on the context, so threads may share a context; a caller that needs a
    // sequence of calls undisturbed (prepare ... aggregate) brackets it with wh_ctx_lock / wh_ctx_unlock (recursive)
    std::recursive_mutex mu;
    void *comm = nullptr;        // multi-GPU state (comm.hip)
    void *count = nullptr;       // Prio3Count / Field64 state (count64.hip)
    void *grouptest = nullptr;   // job and result staging of the batch-proof kernel (grouptest.hip)
    Knobs knobs;
    double k_ms[WH_NUM_KERNELS] = {0};
    uint64_t k_launches[WH_NUM_KERNELS] = {0};
    std::string err;
};

namespace {

#define WH_LOCK(ctx) std::lock_guard<std::recursive_mutex> wh_lock_guard_(ctx->mu)
#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return WH_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

// Address of the file-scope `scratch` array on the current device.  Every GPU holds its own copy at its own address, so
// the look-up is made once per device, with that device current, and kept.
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
uint32_t *g_scratch_ptr[MAX_DEVICES] = {nullptr};
// "last user" of a device's scratch: every allocation-free sort waits for it on its own stream before it launches and records
// it afterwards, so two contexts on one GPU (two streams) take their turns on the array instead of racing.  Created with the
// address, once per device: the sort itself creates nothing.
hipEvent_t g_scratch_ev[MAX_DEVICES] = {nullptr};
int32_t scratch_of_device(wh_ctx *c) {
    if (c->device < 0 || c->device >= MAX_DEVICES) { c->err = "device index beyond the scratch table"; return WH_ERR_NO_DEVICE; }
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (!g_scratch_ptr[c->device]) {
        void *p = nullptr;
        HIPCHK(c, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) { c->err = "hipGetSymbolAddress(scratch) returned a null address"; return WH_ERR_HIP; }
        hipEvent_t ev = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g_scratch_ev[c->device] = ev;
        g_scratch_ptr[c->device] = static_cast<uint32_t *>(p);
    }
    c->d_scratch = g_scratch_ptr[c->device];
    return WH_OK;
}

// The context remembers the per-report status of its last batch for the aggregation's keep mask.  A caller-owned array
// may be freed or reused between the two calls, so its contents are copied (in stream order, after the kernels that
// write it) into context memory; `keep` must have been sized before the launches.
int32_t remember_status(wh_ctx *c, const int32_t *d_status, uint64_t n) {
    if (!n) { c->d_status_last = nullptr; return WH_OK; }
    if (d_status == c->d_status_own) { c->d_status_last = c->d_status_own; return WH_OK; }
    HIPCHK(c, hipMemcpyAsync(c->d_status_keep, d_status, 4 * n, hipMemcpyDeviceToDevice, c->stream));
    c->d_status_last = c->d_status_keep;
    return WH_OK;
}

hipEvent_t ev_get(wh_ctx *c) {
    if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
// brackets one kernel launch with events when profiling is on
struct KTimer {
    wh_ctx *c; hipEvent_t a = nullptr, b = nullptr; int id; hipStream_t st;
    KTimer(wh_ctx *c_, int id_, hipStream_t st_ = nullptr) : c(c_), id(id_), st(st_ ? st_ : c_->stream) {
        if (c->profiling) { a = ev_get(c); b = ev_get(c); if (a) (void)hipEventRecord(a, st); }
    }
    ~KTimer() {
        if (c->profiling && a && b) { (void)hipEventRecord(b, st); c->evs.push_back({a, b, id}); }
    }
};

template <typename T>
int32_t ensure(wh_ctx *c, T *&ptr, uint64_t &cap, uint64_t want) {
    if (want <= cap && ptr) return WH_OK;
    if (ptr) HIPCHK(c, hipFree(ptr));
    ptr = nullptr; cap = 0;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&ptr), want * sizeof(T)));
    cap = want;
    return WH_OK;
}

void aes_tables(const uint8_t key[16], uint8_t out[432]) {
    uint8_t *rk = out, *sbox = out + 176;
    auto xt = [](uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); };
    auto gm = [&](uint8_t a, uint8_t b) { uint8_t r = 0; while (b) { if (b & 1) r ^= a; a = xt(a); b >>= 1; } return r; };
    for (int x = 0; x < 256; x++) {
        uint8_t inv = 0;
        if (x) for (int y = 1; y < 256; y++) if (gm((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        uint8_t s = inv, r = inv;
        for (int i = 0; i < 4; i++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        sbox[x] = s ^ 0x63;
    }
    memcpy(rk, key, 16);
    uint8_t rcon = 1;
    for (int i = 16; i < 176; i += 4) {
        uint8_t t[4] = {rk[i - 4], rk[i - 3], rk[i - 2], rk[i - 1]};
        if (i % 16 == 0) {
            uint8_t u = t[0];
            t[0] = sbox[t[1]] ^ rcon; t[1] = sbox[t[2]]; t[2] = sbox[t[3]]; t[3] = sbox[u];
            rcon = xt(rcon);
        }
        for (int k = 0; k < 4; k++) rk[i + k] = rk[i - 16 + k] ^ t[k];
    }
}

// workspace for a sub-batch of `cap` reports
int32_t alloc_workspace(wh_ctx *c, uint64_t cap) {
    if (cap <= c->cap) return WH_OK;
    Workspace &W = c->W;
    void *ptrs[] = {W.desc, W.scal, W.weights, W.wlimbs, W.pows, W.rlc, W.hrows, W.hlist, W.llist, W.dirty, W.counters};
    for (void *p : ptrs) if (p) HIPCHK(c, hipFree(p));
    W = Workspace{};
    c->cap = 0;
    const HostSizes &z = c->z;
    W.hrow_bytes = 16ull * (z.input_len + 2ull * z.proof_len);
    HIPCHK(c, hipMalloc((void **)&W.desc, cap * sizeof(RepDesc)));
    HIPCHK(c, hipMalloc((void **)&W.scal, cap * sizeof(RepScal)));
    HIPCHK(c, hipMalloc((void **)&W.weights, cap * (uint64_t)(z.calls + 1) * sizeof(F128)));
    HIPCHK(c, hipMalloc((void **)&W.rlc, cap * (uint64_t)(z.verifier_len + 4 + (c->params.num_aggregators - 1)) * sizeof(F128)));
    if (c->params.circuit != WH_AVERAGE) HIPCHK(c, hipMalloc((void **)&W.wlimbs, cap * (uint64_t)z.calls * 48));
    HIPCHK(c, hipMalloc((void **)&W.pows, cap * sizeof(RepPow)));
    HIPCHK(c, hipMalloc((void **)&W.hrows, cap * W.hrow_bytes));
    HIPCHK(c, hipMalloc((void **)&W.hlist, cap * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void **)&W.llist, cap * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void **)&W.dirty, cap * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void **)&W.counters, 64));
    c->cap = cap;
    return WH_OK;
}

// sub-batch size: enough lanes to fill the chip several times over, bounded by HBM scratch
uint64_t pick_cap(const wh_ctx *c, uint64_t n) {
    const uint64_t hrow = 16ull * (c->z.input_len + 2ull * c->z.proof_len);
    uint64_t by_mem = (64ull << 30) / (hrow + 4096);   // <= 64 GiB of expansion rows (288 GB HBM)
    uint64_t cap = 1ull << 19;
    if (cap > by_mem) cap = by_mem;
    if (const uint64_t v = c->knobs.v[WH_OPT_MAX_SUBBATCH]) {      // tuning / test knob: reports per sub-batch
        if (v >= 64 && v < cap) cap = v;
    }
    if (cap < 64) cap = 64;
    if (n <= cap) return n < 64 ? 64 : n;
    // equal-sized sub-batches, so that the last one does not leave the chip half empty
    const uint64_t parts = (n + cap - 1) / cap;
    return (n + parts - 1) / parts;
}

// query launch shape.  Average: four reports (wavefronts) per workgroup.  ParallelSum circuits:
// one workgroup per report, one wavefront per 64 gadget slots (at most four); LDS holds 48 B of
// weight limbs per gadget call, the hand-over area and the truncate staging of each wave.
typedef void (*query_fn)(Prio3Dev, Workspace, uint8_t *, uint8_t *, int32_t *);
struct QueryLaunch { query_fn fn; uint32_t waves; size_t lds; uint32_t per_block; };     // per_block: reports per workgroup
QueryLaunch query_launch(const wh_ctx *c, bool base) {
    const HostSizes &z = c->z;
    QueryLaunch q;
    const bool polyeval = c->params.circuit == WH_AVERAGE;
    const bool hist = c->params.circuit == WH_HISTOGRAM;
    // a few gadget calls of at most 64 slots: one wavefront per report, everything in registers (k_query_psum_tiny)
    const bool tiny = !polyeval && c->params.chunk <= 64 && z.calls <= QTINY_MAX_CALLS && c->knobs.v[WH_OPT_QUERY_TINY];
    q.per_block = polyeval || tiny ? 4 : 1;
    q.waves = polyeval || tiny ? 4 : (c->params.chunk <= 64 ? 1 : (c->params.chunk <= 128 ? 2 : 4));
    q.lds = polyeval ? (4 * (size_t)z.m + z.m / 2 + 1) * sizeof(F128)
                     : (tiny ? 0 : 48 * (size_t)z.calls + sizeof(QueryShared) + q.waves * (size_t)QT_STAGE_U4 * 16);
    if (polyeval) q.fn = base ? k_query_polyeval<true> : k_query_polyeval<false>;
    else if (tiny) q.fn = base ? (hist ? k_query_psum_tiny<true, true> : k_query_psum_tiny<false, true>)
                               : (hist ? k_query_psum_tiny<true, false> : k_query_psum_tiny<false, false>);
    else if (base) q.fn = hist ? k_query_psum<true, true> : k_query_psum<false, true>;
    else q.fn = hist ? k_query_psum<true, false> : k_query_psum<false, false>;
    // more than the 64 KiB every kernel may ask for: gfx950 gives a workgroup up to 160 KiB on request (then one workgroup
    // per CU: thousands of gadget calls, i.e. a chunk length far below sqrt(input_len), are slow but verified all the same)
    if (q.lds > 64 * 1024 &&
        (q.lds > 160 * 1024 ||
         hipFuncSetAttribute(reinterpret_cast<const void *>(q.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds) != hipSuccess))
        q.fn = nullptr;
    return q;
}
static inline dim3 query_grid(const QueryLaunch &q, uint32_t nb) { return dim3((nb + q.per_block - 1) / q.per_block); }

}  // namespace

// what comm.hip needs of a context
namespace wh {
int ctx_device(wh_ctx *c) { return c->device; }
hipStream_t ctx_stream(wh_ctx *c) { return c->stream; }
void **ctx_comm_slot(wh_ctx *c) { return &c->comm; }
void **ctx_count_slot(wh_ctx *c) { return &c->count; }
void count64_free(void *state);
void **ctx_grouptest_slot(wh_ctx *c) { return &c->grouptest; }
void grouptest_free(void *state);
std::recursive_mutex &ctx_mutex(wh_ctx *c) { return c->mu; }
void ctx_set_error(wh_ctx *c, const std::string &what) { c->err = what; }
uint32_t ctx_output_len(wh_ctx *c) { return c->z.output_len; }
}

extern "C" {

int32_t wh_prio3_sizes_of(const wh_prio3_params *p, wh_prio3_sizes *out) {
    HostSizes z;
    if (!p || !out) return WH_ERR_PARAM;
    int32_t rc = host_sizes(*p, z);
    if (rc) return rc;
    *out = wh_prio3_sizes{z.input_len, z.proof_len, z.verifier_len, z.output_len, z.jr_len,
                          z.calls, z.m, z.arity, z.leader_len, z.helper_len};
    return WH_OK;
}

int32_t wh_ctx_create(int32_t device_id, const wh_prio3_params *params, wh_ctx **out) {
    if (!params || !out) return WH_ERR_PARAM;
    *out = nullptr;
    HostSizes z;
    int32_t rc = host_sizes(*params, z);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return WH_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= ndev) return WH_ERR_NO_DEVICE;
    wh_ctx *c = new wh_ctx();
    c->device = device_id;
    c->params = *params;
    c->z = z;
    auto fail = [&](int32_t code) { wh_ctx_destroy(c); return code; };
    if (hipSetDevice(device_id) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipStreamCreate(&c->stream) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipStreamCreate(&c->stream2) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipStreamCreate(&c->stream3) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipEventCreateWithFlags(&c->ev_join3, hipEventDisableTiming) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipEventCreateWithFlags(&c->ev_bad, hipEventDisableTiming) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipStreamCreate(&c->copy_stream) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipStreamCreate(&c->stream_sort) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipEventCreateWithFlags(&c->ev_ids, hipEventDisableTiming) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipEventCreateWithFlags(&c->ev_sorted, hipEventDisableTiming) != hipSuccess) return fail(WH_ERR_HIP);
    for (int b = 0; b < 2; b++)
        if (hipEventCreateWithFlags(&c->ev_h2d[b], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_done[b], hipEventDisableTiming) != hipSuccess) return fail(WH_ERR_HIP);
    c->err.reserve(256);                 // later error texts need no host allocation either
    knobs_from_env(c->knobs);            // the only look at the environment
    // the allocation-free sort's scratch: a context that never calls that entry point must not depend on it, so a failure
    // here only leaves d_scratch empty (wh_sort_by_testing_id_noalloc_dev then reports WH_ERR_STATE with this text)
    if (scratch_of_device(c) != WH_OK) { c->d_scratch = nullptr; c->scratch_err = c->err; c->err.clear(); }

    // constant tables: w^j and s_i = sum_{j=1..calls} w^(ij)
    std::vector<F128> omega(z.m), stab(z.gpoly_len);
    F128 w = root_of_unity(z.log2m);
    omega[0] = f_one();
    for (uint32_t j = 1; j < z.m; j++) omega[j] = f_mul(omega[j - 1], w);
    for (uint32_t i = 0; i < z.gpoly_len; i++) {
        F128 acc = f_zero();
        for (uint32_t j = 1; j <= z.calls; j++) acc = f_add(acc, omega[((uint64_t)i * j) % z.m]);
        stab[i] = acc;
    }
    if (hipMalloc((void **)&c->d_omega, z.m * sizeof(F128)) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipMalloc((void **)&c->d_stab, z.gpoly_len * sizeof(F128)) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipMalloc((void **)&c->d_aes, 432) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipMemcpy(c->d_omega, omega.data(), z.m * sizeof(F128), hipMemcpyHostToDevice) != hipSuccess) return fail(WH_ERR_HIP);
    if (hipMemcpy(c->d_stab, stab.data(), z.gpoly_len * sizeof(F128), hipMemcpyHostToDevice) != hipSuccess) return fail(WH_ERR_HIP);

    Prio3Dev &P = c->P;
    P.circuit = params->circuit; P.bits = params->bits; P.len = params->len; P.chunk = params->chunk;
    P.input_len = z.input_len; P.proof_len = z.proof_len; P.verifier_len = z.verifier_len; P.output_len = z.output_len;
    P.jr_len = z.jr_len; P.calls = z.calls; P.m = z.m; P.log2m = z.log2m; P.arity = z.arity; P.gpoly_len = z.gpoly_len;
    P.alg_id = z.alg_id; P.leader_len = z.leader_len; P.helper_len = z.helper_len; P.n_weights = z.calls + 1;
    P.num_aggregators = params->num_aggregators; P.mode = 0;
    P.half = f_inv(f_small(params->num_aggregators));
    P.m_inv = f_inv(f_small(z.m));
    P.sampler_bound = F128{P_LO, P_HI};
    P.omega = c->d_omega; P.s_tab = c->d_stab;
    *out = c;
    return WH_OK;
}

void wh_ctx_destroy(wh_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    wh_comm_destroy(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    wh::count64_free(c->count); c->count = nullptr;
    wh::grouptest_free(c->grouptest); c->grouptest = nullptr;
    if (c->stream2) hipStreamSynchronize(c->stream2);
    if (c->stream3) hipStreamSynchronize(c->stream3);
    if (c->stream_sort) { hipStreamSynchronize(c->stream_sort); hipStreamDestroy(c->stream_sort); }
    if (c->ev_ids) hipEventDestroy(c->ev_ids);
    if (c->ev_sorted) hipEventDestroy(c->ev_sorted);
    void *ptrs[] = {c->d_omega, c->d_stab, c->d_aes, c->d_out_shares, c->d_status_own, c->d_status_keep, c->d_tmp_a, c->d_tmp_b, c->d_tmp_c, c->d_keep_eff, c->d_partials,
                    c->d_blob, c->d_offsets, c->d_tags, c->d_ids, c->d_prep, c->d_jr_seeds, c->W.desc, c->W.scal, c->W.weights, c->W.wlimbs, c->W.pows, c->W.rlc,
                    c->W.hrows, c->W.hlist, c->W.llist, c->W.dirty, c->W.counters,
                    c->d_omega2m, c->d_twist, c->d_hh_tab, c->hh_w.tpc, c->hh_w.proofblk, c->hh_w.supd, c->hh_w.buckg, c->d_hh_dat, c->d_hh_keys, c->d_hh_tags,
                    c->d_hh_ids, c->d_hh_status, c->d_hh_keep, c->d_hh_acc, c->d_hh_part, c->d_sort_tmp, c->d_sort_hist, c->d_bad_ids};
    for (void *p : ptrs) if (p) hipFree(p);
    if (c->h_bad) hipHostFree(c->h_bad);
    if (c->copy_stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
    for (int b = 0; b < 2; b++) { if (c->ev_h2d[b]) hipEventDestroy(c->ev_h2d[b]); if (c->ev_done[b]) hipEventDestroy(c->ev_done[b]); }
    if (c->h_stage) hipHostFree(c->h_stage);
    for (auto &e : c->evs) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (auto e : c->ev_pool) hipEventDestroy(e);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->ev_bad) hipEventDestroy(c->ev_bad);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->ev_join3) hipEventDestroy(c->ev_join3);
    if (c->stream3) hipStreamDestroy(c->stream3);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

// The text is copied under the context's lock into a buffer of the CALLING thread: the pointer stays valid until that thread
// asks again, whatever other threads do to the context meanwhile.  It is the context's last error: a thread that wants the
// text of its own failed call brackets call and query with wh_ctx_lock / wh_ctx_unlock.
const char *wh_last_error(const wh_ctx *c) {
    if (!c) return "null context";
    static thread_local std::string mine;
    wh_ctx *m = const_cast<wh_ctx *>(c);
    std::lock_guard<std::recursive_mutex> g(m->mu);
    mine = m->err;
    return mine.c_str();
}
// the funnel for concurrent callers (32 tokio tasks, a rayon pool): tasks queue on the context; recursive, so the calls
// inside a bracket still take it
int32_t wh_ctx_lock(wh_ctx *c) { if (!c) return WH_ERR_PARAM; c->mu.lock(); return WH_OK; }
int32_t wh_ctx_unlock(wh_ctx *c) { if (!c) return WH_ERR_PARAM; c->mu.unlock(); return WH_OK; }
void *wh_ctx_stream(wh_ctx *c) { return c ? (void *)c->stream : nullptr; }
int32_t wh_ctx_synchronize(wh_ctx *c) {
    if (!c) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

// One batched verification in two parts, so that the host-buffer entry point can feed it chunk by chunk while the next chunk
// is still crossing PCIe: prepare_begin sizes the context for n reports (all allocations happen here), prepare_range enqueues
// the kernel sequence for reports [base, base + nb) whose encodings sit at d_blob + d_off[0 .. nb].
struct SortPlan { uint64_t n; uint32_t *d_order; uint8_t *d_sorted_ids; };
struct PrepPlan { uint64_t vk0 = 0, vk1 = 0; QueryLaunch q{}; const SortPlan *sort = nullptr; };

static int32_t prepare_begin(wh_ctx *c, const uint8_t verify_key[16], uint64_t n, uint64_t max_range, const int32_t *d_status,
                             PrepPlan &pl) {
    // (any number of aggregators: the servers run two, Prio3::prepare_batched is generic, prio3.rs:2843-2855)
    if (n > 0xfffffff0ull) return WH_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    const HostSizes &z = c->z;
    int32_t rc;
    c->P.mode = 0; c->P.ver_stride = 0;
    c->P.half = f_inv(f_small(c->params.num_aggregators));
    // allocations first (none inside the launch sequence once the context is warm)
    if ((rc = alloc_workspace(c, pick_cap(c, max_range)))) return rc;
    if ((rc = ensure(c, c->d_out_shares, c->out_cap, (n ? n : 1) * (uint64_t)z.output_len))) return rc;
    if (d_status != c->d_status_own && (rc = ensure(c, c->d_status_keep, c->status_keep_cap, n ? n : 1))) return rc;
    c->producer = wh_ctx::NONE; c->n_last = 0;           // until this batch is enqueued completely
    if (!c->aes_valid || memcmp(c->d_aes_key, verify_key, 16)) {
        uint8_t t[432];
        aes_tables(verify_key, t);
        HIPCHK(c, hipMemcpyAsync(c->d_aes, t, 432, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        memcpy(c->d_aes_key, verify_key, 16);
        c->aes_valid = true;
    }
    memcpy(&pl.vk0, verify_key, 8); memcpy(&pl.vk1, verify_key + 8, 8);
    pl.q = query_launch(c, false);
    if (!pl.q.fn) { c->err = "circuit has too many gadget calls for the query kernel's LDS rows (48 B per call in 160 KiB)"; return WH_ERR_UNSUPPORTED; }
    return WH_OK;
}

// d_tags / d_ids / d_status are the whole batch's arrays; the range writes its slice [base, base + nb)
static int32_t prepare_range(wh_ctx *c, const PrepPlan &pl, const uint8_t *d_blob, const uint64_t *d_off, uint64_t nr, uint64_t base0,
                             uint8_t *d_tags, uint8_t *d_ids, int32_t *d_status) {
    const HostSizes &z = c->z;
    const QueryLaunch &q = pl.q;
    uint8_t *d_out = reinterpret_cast<uint8_t *>(c->d_out_shares);
    for (uint64_t r = 0; r < nr; r += c->cap) {
        const uint64_t base = base0 + r;
        const uint32_t nb = (uint32_t)((nr - r < c->cap) ? nr - r : c->cap);
        const uint32_t npad = (nb + 255) / 256 * 256;
        uint8_t *tags_b = d_tags + 16 * base, *out_b = d_out + 16 * base * z.output_len;
        HIPCHK(c, hipMemsetAsync(c->W.counters, 0, 64, c->stream));
        { KTimer t_(c, WH_K_PARSE);
        hipLaunchKernelGGL(k_parse, dim3(npad / 256), dim3(256), 0, c->stream, c->P, c->W, d_blob, d_off + r, nb,
                           d_status + base, tags_b, d_ids + 16 * base, out_b, c->d_aes); }
        // wh_prio3_prepare_sorted_batch_dev: every testing id of the batch exists once the last sub-batch is parsed, and the
        // sort of the permutation (dozens of short launches) goes to a stream of its own.  Short shares: beside the chain
        // kernel.  Long shares: beside the hash and query kernels, BEHIND k_meas -- launches that take wave slots while k_meas
        // places its three long wavefronts per SIMD cost it 2 ms (31.9 against 29.4 ms per 125 000 reports).
        const bool sort_here = pl.sort && r + c->cap >= nr && base0 + nr == pl.sort->n;
        auto start_sort = [&]() -> int32_t {
            const SortPlan &sp = *pl.sort;
            HIPCHK(c, hipEventRecord(c->ev_ids, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream_sort, c->ev_ids, 0));
            int e;
            { KTimer t_(c, WH_K_SORT, c->stream_sort);
              e = launch_sort_by_id(d_ids, (uint32_t)sp.n, sp.d_order, c->d_sort_tmp, c->d_sort_hist, c->stream_sort);
              if (!e && sp.d_sorted_ids) e = launch_sort_gather(d_ids, 16, sp.d_order, (uint32_t)sp.n, sp.d_sorted_ids, c->stream_sort); }
            if (e) { c->err = std::string("sort launch: ") + hipGetErrorString((hipError_t)e); return WH_ERR_HIP; }
            HIPCHK(c, hipEventRecord(c->ev_sorted, c->stream_sort));
            return WH_OK;
        };
        const bool chain = z.input_len <= c->knobs.v[WH_OPT_CHAIN_MAX_INPUT_LEN];
        if (sort_here && chain) { const int32_t rc_ = start_sort(); if (rc_) return rc_; }
        if (chain) {
            // short measurement shares: every sponge of the batch in one launch (chain.hip), then the query
            { KTimer t_(c, WH_K_CHAIN);
            launch_chain(c->P, c->W, npad, c->knobs.v[WH_OPT_CHAIN_PREFETCH] != 0, pl.vk0, pl.vk1, d_status + base, c->stream); }
            { KTimer t_(c, WH_K_QUERY);
            hipLaunchKernelGGL(q.fn, query_grid(q, nb), dim3(64 * q.waves), q.lds, c->stream, c->P, c->W,
                               tags_b, out_b, d_status + base); }
            HIPCHK(c, hipGetLastError());
            continue;
        }
        // side stream: the Helpers' proof shares and the hash kernel's job B (hash of the public parts, RLC coefficients,
        // testing ids).  Both are independent of k_meas and run in the wave slots it leaves; what then remains between
        // k_meas and the query kernel is job A alone (3.9 -> 3.2 ms on the critical path, +0.3 ms in k_meas).
        // Neither k_expand nor k_meas writes a status, so job B sees the statuses it would see after them.
        const bool side = c->knobs.v[WH_OPT_SIDE_STREAM] != 0;
        const bool hash_b_side = side && c->knobs.v[WH_OPT_HASH_B_SIDE];
        hipStream_t s2 = side ? c->stream2 : c->stream;
        if (side) {
            HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        }
        { KTimer t_(c, WH_K_EXPAND, s2);
        launch_expand(c->P, c->W, npad, 1, s2); }
        // Very short measurement shares (Histogram of 128, Average): the step is a chain of kernel latencies, and job B gets a
        // stream of its own beside the proof-share expansion (Histogram 3.12 -> 2.92 ms per 100 k reports, Average unchanged).
        // From cfg-1's 2 048 elements on it stays behind the expansion: beside it, it takes wave slots before k_meas has placed
        // its wavefronts (cfg-1: k_meas 16.0 -> 18.3 ms, step 29.5 -> 30.1 ms; sumvec_large loses 0.4-1.5 ms).
        const bool b_third = hash_b_side && c->z.input_len <= 256 && c->knobs.v[WH_OPT_THIRD_STREAM];
        if (b_third) HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
        if (hash_b_side)
            hipLaunchKernelGGL(k_hash, dim3(npad / 256), dim3(256), 0, b_third ? c->stream3 : s2, c->P, c->W, npad, 1u, pl.vk0, pl.vk1,
                               d_status + base);
        if (b_third) HIPCHK(c, hipEventRecord(c->ev_join3, c->stream3));
        if (side) HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
        { KTimer t_(c, WH_K_MEAS);
        launch_meas(c->P, c->W, 0, npad, d_status + base, c->stream); }     // measurement shares and their sponges
        // reports whose share stream had a rejected chunk (none, in practice): the general path
        launch_expand(c->P, c->W, npad, 2, c->stream);
        launch_meas(c->P, c->W, 1, npad, d_status + base, c->stream);
        if (sort_here) { const int32_t rc_ = start_sort(); if (rc_) return rc_; }
        if (side) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        if (b_third) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join3, 0));
        { KTimer t_(c, WH_K_HASH);
        hipLaunchKernelGGL(k_hash, dim3((hash_b_side ? 1 : 2) * npad / 256), dim3(256), 0, c->stream, c->P, c->W, npad,
                           hash_b_side ? 0u : 2u, pl.vk0, pl.vk1, d_status + base); }
        { KTimer t_(c, WH_K_QUERY);
        hipLaunchKernelGGL(q.fn, query_grid(q, nb), dim3(64 * q.waves), q.lds, c->stream, c->P, c->W,
                           tags_b, out_b, d_status + base); }
        HIPCHK(c, hipGetLastError());
    }
    return WH_OK;
}

static int32_t prepare_end(wh_ctx *c, const int32_t *d_status, uint64_t n) {
    int32_t rc;
    if ((rc = remember_status(c, d_status, n))) return rc;
    c->n_last = n; c->producer = wh_ctx::PREPARE_BATCH;
    return WH_OK;
}

int32_t wh_prio3_prepare_batch_dev(wh_ctx *c, const uint8_t verify_key[16], const uint8_t *d_blob,
                                   const uint64_t *d_offsets, uint64_t n, uint8_t *d_tags, uint8_t *d_ids,
                                   int32_t *d_status) {
    if (!c || !verify_key || (n && (!d_blob || !d_offsets || !d_tags || !d_ids || !d_status))) return WH_ERR_PARAM;
    WH_LOCK(c);
    PrepPlan pl;
    int32_t rc;
    if ((rc = prepare_begin(c, verify_key, n, n, d_status, pl))) return rc;
    if ((rc = prepare_range(c, pl, d_blob, d_offsets, n, 0, d_tags, d_ids, d_status))) return rc;
    return prepare_end(c, d_status, n);
}

// run_vdaf_prepare up to the group test (main.rs:75-94): wh_prio3_prepare_batch_dev + wh_sort_by_testing_id_dev as one call.  Same
// outputs as the two calls; the difference is where the sort runs.  The testing ids of the batch exist after the parse kernel,
// so the sort of the permutation and the gather of the sorted ids -- a few dozen launches, each far too small for the chip --
// run on a stream of their own beside the sponge kernels instead of behind the query kernel; only the gather of the sorted
// tags waits for the tags.
int32_t wh_prio3_prepare_sorted_batch_dev(wh_ctx *c, const uint8_t verify_key[16], const uint8_t *d_blob, const uint64_t *d_offsets,
                                          uint64_t n, uint8_t *d_tags, uint8_t *d_ids, int32_t *d_status, uint32_t *d_order,
                                          uint8_t *d_sorted_ids, uint8_t *d_sorted_tags) {
    if (!c || !verify_key || (n && (!d_blob || !d_offsets || !d_tags || !d_ids || !d_status || !d_order))) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (n > (1ull << 28)) return WH_ERR_PARAM;
    PrepPlan pl;
    int32_t rc;
    if ((rc = prepare_begin(c, verify_key, n, n, d_status, pl))) return rc;
    if (n) {
        if ((rc = ensure(c, c->d_sort_tmp, c->sort_tmp_cap, n))) return rc;
        if ((rc = ensure(c, c->d_sort_hist, c->sort_hist_cap, 256 * sort_blocks(n) + 256))) return rc;
    }
    const SortPlan sp{n, d_order, d_sorted_ids};
    pl.sort = &sp;
    if ((rc = prepare_range(c, pl, d_blob, d_offsets, n, 0, d_tags, d_ids, d_status))) return rc;
    if (n) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_sorted, 0));
        if (d_sorted_tags) {
            const int e = launch_sort_gather(d_tags, 16, d_order, (uint32_t)n, d_sorted_tags, c->stream);
            if (e) { c->err = std::string("sort gather launch: ") + hipGetErrorString((hipError_t)e); return WH_ERR_HIP; }
        }
    }
    return prepare_end(c, d_status, n);
}

int32_t wh_prio3_prepare_init_batch_dev(wh_ctx *c, const uint8_t verify_key[16], const uint8_t *d_blob,
                                        const uint64_t *d_offsets, uint64_t n, uint8_t *d_prep, int32_t *d_status) {
    if (!c || !verify_key || (n && (!d_blob || !d_offsets || !d_prep || !d_status))) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (n > 0xfffffff0ull) return WH_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    const HostSizes &z = c->z;
    int32_t rc;
    c->P.mode = 1; c->P.ver_stride = z.verifier_len + 1;
    c->P.half = f_inv(f_small(c->params.num_aggregators));
    if ((rc = alloc_workspace(c, pick_cap(c, n)))) return rc;
    if ((rc = ensure(c, c->d_out_shares, c->out_cap, (n ? n : 1) * (uint64_t)z.output_len))) return rc;
    // the prepare state prepare_finish needs (Prio3PrepareState, prio3.rs:1976-2010): the out shares above and the joint-rand
    // seed of every report -- kept outside the workspace, so a batch may take any number of sub-batches, as the reference's
    // loop takes any Vec (server-base-prio3/src/main.rs:41-61)
    if ((rc = ensure(c, c->d_jr_seeds, c->jr_seeds_cap, n ? n : 1))) return rc;
    if (d_status != c->d_status_own && (rc = ensure(c, c->d_status_keep, c->status_keep_cap, n ? n : 1))) return rc;
    uint64_t vk0, vk1;
    memcpy(&vk0, verify_key, 8); memcpy(&vk1, verify_key + 8, 8);
    c->producer = wh_ctx::NONE; c->n_last = 0; c->init_finished = false;
    const QueryLaunch q = query_launch(c, true);
    if (!q.fn) { c->err = "circuit has too many gadget calls for the query kernel's LDS rows (48 B per call in 160 KiB)"; return WH_ERR_UNSUPPORTED; }
    if (!n) { c->d_status_last = nullptr; c->producer = wh_ctx::PREPARE_INIT; c->init_finished = true; return WH_OK; }   // nothing to decide
    uint8_t *d_out = reinterpret_cast<uint8_t *>(c->d_out_shares);
    const uint64_t prep_row = 16ull * (z.verifier_len + 1);
    const bool paired = c->knobs.v[WH_OPT_BASE_PAIRED] != 0;
    for (uint64_t base = 0; base < n; base += c->cap) {
        const uint32_t nb = (uint32_t)(n - base < c->cap ? n - base : c->cap), npad = (nb + 255) / 256 * 256;
        uint8_t *prep_b = d_prep + prep_row * base, *out_b = d_out + 16 * base * z.output_len;
        int32_t *status_b = d_status + base;
        HIPCHK(c, hipMemsetAsync(c->W.counters, 0, 64, c->stream));
        { KTimer t_(c, WH_K_PARSE);
        hipLaunchKernelGGL(k_parse_base, dim3(npad / 256), dim3(256), 0, c->stream, c->P, c->W, d_blob, d_offsets + base, nb, status_b,
                           prep_b, out_b); }
        // as in the batched path: the Helpers' proof shares on the side stream, the measurement shares and their joint-rand
        // sponges in k_meas (Leader lanes absorb from the key, Helper lane pairs squeeze and absorb in step); keys whose share
        // stream had a rejected chunk go through the general path.  WH_OPT_BASE_PAIRED = 0 keeps the two-pass form.
        if (paired) {
            HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            { KTimer t_(c, WH_K_EXPAND, c->stream2);
            launch_expand(c->P, c->W, npad, 1, c->stream2); }
            HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
            { KTimer t_(c, WH_K_MEAS);
            launch_meas(c->P, c->W, 0, npad, status_b, c->stream); }
            launch_expand(c->P, c->W, npad, 2, c->stream);
            launch_meas(c->P, c->W, 1, npad, status_b, c->stream);
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        } else {
            KTimer t_(c, WH_K_EXPAND);
            launch_expand(c->P, c->W, npad, 0, c->stream);
        }
        { KTimer t_(c, WH_K_HASH);
        hipLaunchKernelGGL(k_hash_base, dim3(npad / 256), dim3(256), 0, c->stream, c->P, c->W, vk0, vk1, status_b, paired ? 1u : 0u); }
        { KTimer t_(c, WH_K_QUERY);
        hipLaunchKernelGGL(q.fn, query_grid(q, nb), dim3(64 * q.waves), q.lds, c->stream, c->P, c->W, prep_b, out_b, status_b); }
        hipLaunchKernelGGL(k_keep_seeds, dim3(npad / 256), dim3(256), 0, c->stream, c->W, nb, c->d_jr_seeds + base);
        HIPCHK(c, hipGetLastError());
    }
    if ((rc = remember_status(c, d_status, n))) return rc;
    c->n_last = n; c->producer = wh_ctx::PREPARE_INIT;
    return WH_OK;
}

int32_t wh_prio3_prepare_init_batch(wh_ctx *c, const uint8_t verify_key[16], const uint8_t *blob, const uint64_t *offsets,
                                    uint64_t n, uint8_t *prep_shares, int32_t *status) {
    if (!c || !verify_key || (n && (!blob || !offsets || !prep_shares || !status))) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (n == 0) { c->n_last = 0; c->d_status_last = nullptr; c->producer = wh_ctx::PREPARE_INIT; c->init_finished = true; return WH_OK; }
    for (uint64_t i = 0; i < n; i++) if (offsets[i + 1] < offsets[i]) return WH_ERR_PARAM;
    int32_t rc;
    const uint64_t base0 = offsets[0], bytes = offsets[n] - base0, prep_bytes = 16ull * (c->z.verifier_len + 1) * n;
    if ((rc = ensure(c, c->d_blob, c->blob_cap, bytes + 64))) return rc;
    if ((rc = ensure(c, c->d_offsets, c->off_cap, n + 1))) return rc;
    if ((rc = ensure(c, c->d_prep, c->prep_cap, prep_bytes))) return rc;
    if ((rc = ensure(c, c->d_status_own, c->status_cap, n))) return rc;
    std::vector<uint64_t> rel(n + 1);
    for (uint64_t i = 0; i <= n; i++) rel[i] = offsets[i] - base0;
    HIPCHK(c, hipMemcpyAsync(c->d_blob, blob + base0, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_offsets, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((rc = wh_prio3_prepare_init_batch_dev(c, verify_key, c->d_blob, c->d_offsets, n, c->d_prep, c->d_status_own))) return rc;
    HIPCHK(c, hipMemcpyAsync(prep_shares, c->d_prep, prep_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(status, c->d_status_own, 4 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

int32_t wh_prio3_prepare_finish_batch(wh_ctx *c, const uint8_t *const *prep_shares, uint64_t n, uint8_t *prep_msgs,
                                      int32_t *status) {
    if (!c || (n && (!prep_shares || !prep_msgs || !status))) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->producer != wh_ctx::PREPARE_INIT || n != c->n_last || (n && c->d_status_last != c->d_status_own)) return WH_ERR_STATE;
    if (n == 0) { c->init_finished = true; return WH_OK; }
    const uint32_t na = c->params.num_aggregators;
    const uint64_t prep_bytes = 16ull * (c->z.verifier_len + 1) * n;
    for (uint32_t a = 0; a < na; a++) if (!prep_shares[a]) return WH_ERR_PARAM;
    int32_t rc;
    // scratch from the context (grown on demand): all prepare shares, the messages, the row pointers
    if ((rc = ensure(c, c->d_tmp_a, c->tmp_a_cap, prep_bytes * na))) return rc;
    if ((rc = ensure(c, c->d_tmp_b, c->tmp_b_cap, 16 * n))) return rc;
    if ((rc = ensure(c, c->d_tmp_c, c->tmp_c_cap, sizeof(uint8_t *) * na))) return rc;
    uint8_t *d_all = c->d_tmp_a, *d_msgs = c->d_tmp_b;
    const uint8_t **d_ptrs = reinterpret_cast<const uint8_t **>(c->d_tmp_c);
    std::vector<const uint8_t *> ptrs(na);
    for (uint32_t a = 0; a < na; a++) {
        ptrs[a] = d_all + prep_bytes * a;
        HIPCHK(c, hipMemcpyAsync(d_all + prep_bytes * a, prep_shares[a], prep_bytes, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(d_ptrs, ptrs.data(), sizeof(uint8_t *) * na, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // `ptrs` is a local
    HIPCHK(c, hipMemsetAsync(d_msgs, 0, 16 * n, c->stream));
    hipLaunchKernelGGL(k_finish_base, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, c->stream, c->P, c->d_jr_seeds, d_ptrs, (uint32_t)n,
                       d_msgs, c->d_status_own);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(prep_msgs, d_msgs, 16 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(status, c->d_status_own, 4 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->init_finished = true;
    return WH_OK;
}

// Type::query on rows already in HBM: d_in[n][input_len], d_pf[n][proof_len], d_qr[n][1], d_jr[n][jr_len] -> d_ver[n][verifier_len]
static int32_t flp_query_dev(wh_ctx *c, const uint8_t *d_in, const uint8_t *d_pf, const uint8_t *d_qr, const uint8_t *d_jr,
                             uint32_t num_shares, uint64_t n, uint8_t *d_ver, int32_t *d_status) {
    const HostSizes &z = c->z;
    int32_t rc;
    c->P.mode = 2; c->P.ver_stride = z.verifier_len;
    c->P.half = f_inv(f_small(num_shares));
    if ((rc = alloc_workspace(c, pick_cap(c, n)))) return rc;
    if (n > c->cap) { c->err = "flp query: one sub-batch only"; return WH_ERR_UNSUPPORTED; }
    if ((rc = ensure(c, c->d_out_shares, c->out_cap, (n ? n : 1) * (uint64_t)z.output_len))) return rc;
    // Type::query leaves Type::truncate of its inputs for wh_prio3_out_shares; there is nothing to aggregate
    c->n_last = n; c->d_status_last = nullptr; c->producer = wh_ctx::FLP_QUERY;
    if (!n) return WH_OK;
    const QueryLaunch q = query_launch(c, true);
    if (!q.fn) { c->err = "circuit has too many gadget calls for the query kernel's LDS rows (48 B per call in 160 KiB)"; return WH_ERR_UNSUPPORTED; }
    HIPCHK(c, hipMemsetAsync(c->W.counters, 0, 64, c->stream));
    const uint32_t nb = (uint32_t)n, npad = (nb + 255) / 256 * 256;
    hipLaunchKernelGGL(k_flp_setup, dim3(npad / 256), dim3(256), 0, c->stream, c->P, c->W, d_in, d_pf, d_qr, d_jr, nb, d_status);
    hipLaunchKernelGGL(q.fn, query_grid(q, nb), dim3(64 * q.waves), q.lds, c->stream, c->P, c->W, d_ver,
                       reinterpret_cast<uint8_t *>(c->d_out_shares), d_status);
    HIPCHK(c, hipGetLastError());
    return WH_OK;
}

int32_t wh_flp_query_batch(wh_ctx *c, const uint8_t *input, const uint8_t *proof, const uint8_t *query_rand,
                           const uint8_t *joint_rand, uint32_t num_shares, uint64_t n, uint8_t *verifier, int32_t *status) {
    if (!c || !num_shares || (n && (!input || !proof || !query_rand || !joint_rand || !verifier || !status))) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (n > 0xfffffff0ull) return WH_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    const HostSizes &z = c->z;
    int32_t rc;
    const uint64_t in_b = 16ull * z.input_len * n, pf_b = 16ull * z.proof_len * n, jr_b = 16ull * z.jr_len * n, ver_b = 16ull * z.verifier_len * n;
    if ((rc = ensure(c, c->d_blob, c->blob_cap, in_b + pf_b + 16 * n + jr_b + 64))) return rc;
    if ((rc = ensure(c, c->d_prep, c->prep_cap, ver_b ? ver_b : 16))) return rc;
    if ((rc = ensure(c, c->d_status_own, c->status_cap, n ? n : 1))) return rc;
    uint8_t *d_in = c->d_blob, *d_pf = d_in + in_b, *d_qr = d_pf + pf_b, *d_jr = d_qr + 16 * n;
    if (n) {
        HIPCHK(c, hipMemcpyAsync(d_in, input, in_b, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_pf, proof, pf_b, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_qr, query_rand, 16 * n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_jr, joint_rand, jr_b, hipMemcpyHostToDevice, c->stream));
    }
    if ((rc = flp_query_dev(c, d_in, d_pf, d_qr, d_jr, num_shares, n, c->d_prep, c->d_status_own))) return rc;
    if (!n) return WH_OK;
    HIPCHK(c, hipMemcpyAsync(verifier, c->d_prep, ver_b, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(status, c->d_status_own, 4 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

// transform tables of the prover, built on first use: w_2m^k, and the twist w_2m^bitrev(pos) / m that turns coefficients into
// the odd half of the 2m evaluations
static int32_t prove_tables(wh_ctx *c, ProveDev &D) {
    const HostSizes &z = c->z;
    if (!c->d_omega2m) {
        const uint32_t m = z.m, two_m = 2 * z.m;
        std::vector<F128> om(two_m), tw(m);
        const F128 w = root_of_unity(z.log2m + 1), minv = f_inv(f_small(m));
        om[0] = f_one();
        for (uint32_t k = 1; k < two_m; k++) om[k] = f_mul(om[k - 1], w);
        for (uint32_t pos = 0; pos < m; pos++) {
            uint32_t r = 0;
            for (uint32_t b = 0; b < z.log2m; b++) r |= ((pos >> b) & 1) << (z.log2m - 1 - b);
            tw[pos] = f_mul(om[r], minv);
        }
        HIPCHK(c, hipMalloc((void **)&c->d_omega2m, two_m * sizeof(F128)));
        HIPCHK(c, hipMalloc((void **)&c->d_twist, m * sizeof(F128)));
        HIPCHK(c, hipMemcpy(c->d_omega2m, om.data(), two_m * sizeof(F128), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_twist, tw.data(), m * sizeof(F128), hipMemcpyHostToDevice));
    }
    D = ProveDev{z.input_len, c->params.chunk, z.calls, z.m, z.log2m, z.arity, z.gpoly_len, z.proof_len, z.jr_len,
                 c->d_omega2m, c->d_twist, f_inv(f_small(2ull * z.m))};
    return WH_OK;
}

int32_t wh_flp_prove_batch(wh_ctx *c, const uint8_t *input, const uint8_t *prove_rand, const uint8_t *joint_rand, uint64_t n,
                           uint8_t *proof, int32_t *status) {
    if (!c || (n && (!input || !prove_rand || !proof || !status)) || (n && c->z.jr_len && !joint_rand)) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (n > 0x7fffffffull) return WH_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    const HostSizes &z = c->z;
    if (z.m > PROVE_MAX_M) { c->err = "flp prove: more than 511 gadget calls do not fit the kernel's LDS"; return WH_ERR_UNSUPPORTED; }
    if (!n) return WH_OK;
    int32_t rc;
    ProveDev D;
    if ((rc = prove_tables(c, D))) return rc;
    const uint64_t in_b = 16ull * z.input_len * n, pr_b = 16ull * z.arity * n, jr_b = 16ull * z.jr_len * n, pf_b = 16ull * z.proof_len * n;
    if ((rc = ensure(c, c->d_blob, c->blob_cap, in_b + pr_b + jr_b + 64))) return rc;
    if ((rc = ensure(c, c->d_prep, c->prep_cap, pf_b))) return rc;
    if ((rc = ensure(c, c->d_status_own, c->status_cap, n))) return rc;
    uint8_t *d_in = c->d_blob, *d_pr = d_in + in_b, *d_jr = d_pr + pr_b;
    HIPCHK(c, hipMemcpyAsync(d_in, input, in_b, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_pr, prove_rand, pr_b, hipMemcpyHostToDevice, c->stream));
    if (jr_b) HIPCHK(c, hipMemcpyAsync(d_jr, joint_rand, jr_b, hipMemcpyHostToDevice, c->stream));
    int e;
    { KTimer t_(c, WH_K_PROVE);
      e = launch_flp_prove(D, c->params.circuit != WH_AVERAGE, d_in, d_pr, d_jr, (uint32_t)n, c->d_prep, c->d_status_own, c->stream); }
    if (e) { c->err = std::string("flp prove launch: ") + hipGetErrorString((hipError_t)e); return WH_ERR_HIP; }
    HIPCHK(c, hipMemcpyAsync(proof, c->d_prep, pf_b, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(status, c->d_status_own, 4 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

// Prio3::shard_batched for n clients on the GPU (shard.hip + the prover + the query kernels), byte for byte what
// wh_client_shard_many writes.  The two report buffers may be host or device memory (hipMemcpyDefault).  Uses the context's query workspace: the state of an earlier prepare call is gone afterwards.
int32_t wh_client_shard_many_gpu(wh_ctx *c, const uint64_t *measurements, uint32_t meas_words, const uint8_t *nonces,
                                 const uint8_t *rands, uint64_t first_client_id, uint64_t n, uint8_t *leader_reports,
                                 uint8_t *helper_reports) {
    if (!c || (n && (!measurements || !nonces || !rands || !leader_reports || !helper_reports))) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (n > 0x7fffffffull) return WH_ERR_PARAM;
    const HostSizes &z = c->z;
    const wh_prio3_params &p = c->params;
    if (p.num_aggregators != 2) { c->err = "shard on the GPU: two aggregators only (wh_client_shard_many takes any number)"; return WH_ERR_UNSUPPORTED; }
    if (meas_words < (p.circuit == WH_SUMVEC ? p.len : 1u)) return WH_ERR_PARAM;
    if (z.m > PROVE_MAX_M) { c->err = "shard: more than 511 gadget calls do not fit the prover's LDS"; return WH_ERR_UNSUPPORTED; }
    if (2 * z.jr_len > 4) { c->err = "shard: joint randomness longer than two elements per proof"; return WH_ERR_UNSUPPORTED; }
    for (uint64_t i = 0; i < n; i++) {               // Type::encode_measurement's range checks (types.rs:246-253, 388-399, 582-606)
        const uint64_t *m = measurements + i * meas_words;
        if (p.circuit == WH_HISTOGRAM) { if (m[0] >= p.len) return WH_ERR_PARAM; }
        else if (p.bits < 64)
            for (uint32_t e = 0; e < (p.circuit == WH_SUMVEC ? p.len : 1u); e++) if (m[e] >> p.bits) return WH_ERR_PARAM;
    }
    if (!n) return WH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int32_t rc;
    ProveDev PD;
    if ((rc = prove_tables(c, PD))) return rc;
    ShardDev D{p.circuit, p.bits, p.len, z.input_len, z.proof_len, z.verifier_len, z.arity, z.jr_len, z.alg_id,
               z.leader_len, z.helper_len, meas_words, c->P.sampler_bound};
    // one arena per call: sizes per client, each region 256-byte aligned
    const uint64_t vin = 16ull * z.input_len, vpf = 16ull * z.proof_len, var = 16ull * z.arity, vjr = 16ull * z.jr_len,
                   vver = 16ull * z.verifier_len;
    const uint64_t per_client = 8ull * meas_words + 16 + 144 + 2 * vin + 6 * vpf + 2 * var + 4 * vjr + 32 + 2 * vver + 64 +
                                z.leader_len + z.helper_len + 4;
    uint64_t sub = (24ull << 30) / per_client;         // one-lane-per-client kernels want many clients per launch; HBM is 288 GB
    const uint64_t qcap = pick_cap(c, n);
    if (sub > qcap) sub = qcap;
    if (sub > n) sub = n;
    if (sub < 1) sub = 1;
    uint8_t *arena = nullptr;
    HIPCHK(c, hipMalloc((void **)&arena, sub * per_client + 32 * 256));
    uint8_t *cur = arena;
    auto take = [&](uint64_t bytes) { uint8_t *r = cur; cur += (bytes + 255) / 256 * 256; return r; };
    ShardBuf B{};
    uint64_t *d_meas = reinterpret_cast<uint64_t *>(take(8ull * meas_words * sub));
    uint8_t *d_nonces = take(16 * sub), *d_rands = take(144 * sub);
    B.meas = d_meas; B.nonces = d_nonces; B.rands = d_rands;
    B.x = take(vin * sub); B.hm = take(vin * sub);
    for (int t = 0; t < 2; t++) { B.hp[t] = take(vpf * sub); B.pr[t] = take(var * sub); B.proof[t] = take(vpf * sub); }
    for (int t = 0; t < 2; t++) { B.jr[t] = take(vjr * sub); B.qr[t] = take(16 * sub); B.pub[t] = take(vver * sub); }
    B.jrf = take(2 * vjr * sub);
    B.parts = take(64 * sub);
    B.lrep = take((uint64_t)z.leader_len * sub); B.hrep = take((uint64_t)z.helper_len * sub);
    int32_t *d_st = reinterpret_cast<int32_t *>(take(4 * sub));
    std::vector<int32_t> st(sub);
    hipError_t e = hipSuccess;
    rc = WH_OK;
    for (uint64_t base = 0; base < n && !rc && e == hipSuccess; base += sub) {
        const uint64_t k = n - base < sub ? n - base : sub;
        const uint32_t kn = (uint32_t)k;
        B.first_id = first_client_id + base;
        if ((e = hipMemcpyAsync(d_meas, measurements + base * meas_words, 8ull * meas_words * k, hipMemcpyHostToDevice, c->stream))) break;
        if ((e = hipMemcpyAsync(d_nonces, nonces + 16 * base, 16 * k, hipMemcpyHostToDevice, c->stream))) break;
        if ((e = hipMemcpyAsync(d_rands, rands + 144 * base, 144 * k, hipMemcpyHostToDevice, c->stream))) break;
        launch_shard_stage1(D, B, kn, c->stream);
        for (int t = 0; t < 2 && e == hipSuccess; t++)
            e = (hipError_t)launch_flp_prove(PD, p.circuit != WH_AVERAGE, B.x, B.pr[t], B.jr[t], kn, B.proof[t], d_st, c->stream);
        if (e) break;
        launch_shard_stage2(D, B, kn, c->stream);
        for (int t = 0; t < 2 && !rc; t++) {
            if ((rc = flp_query_dev(c, B.x, B.proof[t], B.qr[t], B.jr[t], 1, k, B.pub[t], d_st))) break;
            // a query randomness that is an m-th root of unity fails the whole call (flp.rs:413-419), as on the CPU
            if ((e = hipMemcpyAsync(st.data(), d_st, 4 * k, hipMemcpyDeviceToHost, c->stream))) break;
            if ((e = hipStreamSynchronize(c->stream))) break;
            for (uint64_t i = 0; i < k && !rc; i++) if (st[i]) rc = st[i];
        }
        if (rc || e) break;
        launch_shard_stage3(D, B, kn, c->stream);
        if ((e = hipGetLastError())) break;
        if ((e = hipMemcpyAsync(leader_reports + base * z.leader_len, B.lrep, (uint64_t)z.leader_len * k, hipMemcpyDefault, c->stream))) break;
        if ((e = hipMemcpyAsync(helper_reports + base * z.helper_len, B.hrep, (uint64_t)z.helper_len * k, hipMemcpyDefault, c->stream))) break;
        e = hipStreamSynchronize(c->stream);
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(arena);
    c->n_last = 0; c->d_status_last = nullptr; c->producer = wh_ctx::NONE;   // the query kernels ran on scratch of this call
    if (!rc && e != hipSuccess) { c->err = std::string("shard: ") + hipGetErrorString(e); rc = WH_ERR_HIP; }
    return rc;
}

// the servers' stable sort of a batch by testing id, on ids (and tags) already in HBM
int32_t wh_sort_by_testing_id_dev(wh_ctx *c, const uint8_t *d_ids, uint64_t n, uint32_t *d_order, const uint8_t *d_tags,
                                  uint32_t tag_bytes, uint8_t *d_sorted_ids, uint8_t *d_sorted_tags) {
    if (!c || (n && (!d_ids || !d_order))) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (n > (1ull << 28)) return WH_ERR_PARAM;
    if (d_sorted_tags && (!d_tags || !tag_bytes || tag_bytes % 16)) return WH_ERR_PARAM;
    if (!n) return WH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int32_t rc;
    if ((rc = ensure(c, c->d_sort_tmp, c->sort_tmp_cap, n))) return rc;
    if ((rc = ensure(c, c->d_sort_hist, c->sort_hist_cap, 256 * sort_blocks(n) + 256))) return rc;
    int e;
    { KTimer t_(c, WH_K_SORT);
      e = launch_sort_by_id(d_ids, (uint32_t)n, d_order, c->d_sort_tmp, c->d_sort_hist, c->stream);
      if (!e && d_sorted_ids) e = launch_sort_gather(d_ids, 16, d_order, (uint32_t)n, d_sorted_ids, c->stream);
      if (!e && d_sorted_tags) e = launch_sort_gather(d_tags, tag_bytes, d_order, (uint32_t)n, d_sorted_tags, c->stream); }
    if (e) { c->err = std::string("sort launch: ") + hipGetErrorString((hipError_t)e); return WH_ERR_HIP; }
    return WH_OK;
}

// The same sort for callers that cannot allocate (HIP graph capture, serving loops): the second permutation buffer and the
// count table are this device's copy of the file-scope `scratch` array instead of context buffers that grow on demand.
// No hipMalloc / hipFree / event creation on any path; nothing in the scratch has to start at zero (every word the kernels
// read in a pass was written earlier in that pass), so calls of different sizes may follow each other.  All calls on one
// GPU share the array: calls through this entry point are ordered among themselves by a per-device event (each waits for
// the one before on its own stream), whatever context or stream they come from.
int32_t wh_sort_by_testing_id_noalloc_dev(wh_ctx *c, const uint8_t *d_ids, uint64_t n, uint32_t *d_order, const uint8_t *d_tags,
                                          uint32_t tag_bytes, uint8_t *d_sorted_ids, uint8_t *d_sorted_tags) {
    if (!c || (n && (!d_ids || !d_order))) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (n > SORT_NOALLOC_MAX_N) { c->err = "allocation-free sort: more than 2^20 ids (use wh_sort_by_testing_id_dev)"; return WH_ERR_PARAM; }
    if (d_sorted_tags && (!d_tags || !tag_bytes || tag_bytes % 16)) return WH_ERR_PARAM;
    if (!c->d_scratch) { c->err = "allocation-free sort: no scratch on this device: " + c->scratch_err; return WH_ERR_STATE; }
    if (!n) return WH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t *tmp = c->d_scratch, *hist = c->d_scratch + SORT_NOALLOC_MAX_N;
    // one user of the device's array at a time, in the order the calls arrive: behind the last user's event, ahead of the next
    std::lock_guard<std::mutex> g(g_scratch_mu);
    hipEvent_t ev = g_scratch_ev[c->device];
    HIPCHK(c, hipStreamWaitEvent(c->stream, ev, 0));
    int e = launch_sort_by_id(d_ids, (uint32_t)n, d_order, tmp, hist, c->stream);
    if (!e && d_sorted_ids) e = launch_sort_gather(d_ids, 16, d_order, (uint32_t)n, d_sorted_ids, c->stream);
    if (!e && d_sorted_tags) e = launch_sort_gather(d_tags, tag_bytes, d_order, (uint32_t)n, d_sorted_tags, c->stream);
    HIPCHK(c, hipEventRecord(ev, c->stream));
    if (e) { c->err = "allocation-free sort: kernel launch failed"; return WH_ERR_HIP; }
    return WH_OK;
}

// the filter of the servers' merge loops: keep[i] = !bad_set.contains(testing_id[i])
int32_t wh_keep_mask_dev(wh_ctx *c, const uint8_t *d_ids, uint64_t n, const uint8_t *bad_ids, uint64_t n_bad, uint8_t *d_keep) {
    if (!c || (n && (!d_ids || !d_keep)) || (n_bad && !bad_ids)) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (n > 0xfffffff0ull || n_bad > 0xfffffff0ull) return WH_ERR_PARAM;
    if (!n) return WH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int32_t rc;
    if ((rc = ensure(c, c->d_bad_ids, c->bad_ids_cap, 2 * (n_bad ? n_bad : 1)))) return rc;
    if (n_bad) {
        // the sorted ids stay in the context until the next call, so the copy can stay asynchronous; an earlier call's copy
        // out of the same vector must have left the host first
        if (c->bad_sorted_busy) HIPCHK(c, hipEventSynchronize(c->ev_bad));
        if (n_bad > c->h_bad_cap) {
            if (c->h_bad) HIPCHK(c, hipHostFree(c->h_bad));
            c->h_bad = nullptr; c->h_bad_cap = 0;
            HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_bad), 16 * n_bad, hipHostMallocDefault));
            c->h_bad_cap = n_bad;
        }
        memcpy(c->h_bad, bad_ids, 16 * n_bad);
        std::sort(c->h_bad, c->h_bad + n_bad);
        HIPCHK(c, hipMemcpyAsync(c->d_bad_ids, c->h_bad, 16 * n_bad, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_bad, c->stream));
        c->bad_sorted_busy = true;
    }
    const int e = launch_keep_mask(d_ids, (uint32_t)n, c->d_bad_ids, (uint32_t)n_bad, d_keep, c->stream);
    if (e) { c->err = std::string("keep mask launch: ") + hipGetErrorString((hipError_t)e); return WH_ERR_HIP; }
    return WH_OK;
}

int32_t wh_sort_by_testing_id(wh_ctx *c, const uint8_t *ids, uint64_t n, uint32_t *order) {
    if (!c || (n && (!ids || !order))) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (n > (1ull << 28)) return WH_ERR_PARAM;
    if (!n) return WH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int32_t rc;
    if ((rc = ensure(c, c->d_ids, c->ids_cap, 16 * n))) return rc;
    if ((rc = ensure(c, c->d_tmp_a, c->tmp_a_cap, 4 * n))) return rc;
    uint32_t *d_order = reinterpret_cast<uint32_t *>(c->d_tmp_a);
    HIPCHK(c, hipMemcpyAsync(c->d_ids, ids, 16 * n, hipMemcpyHostToDevice, c->stream));
    if ((rc = wh_sort_by_testing_id_dev(c, c->d_ids, n, d_order, nullptr, 0, nullptr, nullptr))) return rc;
    HIPCHK(c, hipMemcpyAsync(order, d_order, 4 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

int32_t wh_flp_decide_batch(wh_ctx *c, const uint8_t *verifier, uint64_t n, int32_t *decision) {
    if (!c || (n && (!verifier || !decision))) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (n > 0xfffffff0ull) return WH_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    if (!n) return WH_OK;
    const uint64_t ver_b = 16ull * c->z.verifier_len * n;
    int32_t rc;
    if ((rc = ensure(c, c->d_tmp_a, c->tmp_a_cap, ver_b))) return rc;
    if ((rc = ensure(c, c->d_tmp_b, c->tmp_b_cap, 4 * n))) return rc;
    uint8_t *d_v = c->d_tmp_a; int32_t *d_ok = reinterpret_cast<int32_t *>(c->d_tmp_b);
    HIPCHK(c, hipMemcpyAsync(d_v, verifier, ver_b, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_flp_decide, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, c->stream, c->P, d_v, (uint32_t)n, d_ok);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(decision, d_ok, 4 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

// pinned host memory for callers that receive reports straight into it (sockets -> pinned buffer -> DMA): the host-buffer
// entry points then skip their staging copy
int32_t wh_host_alloc(uint64_t bytes, void **out) {
    if (!out || !bytes) return WH_ERR_PARAM;
    *out = nullptr;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? WH_OK : WH_ERR_HIP;
}
int32_t wh_host_free(void *p) { return !p || hipHostFree(p) == hipSuccess ? WH_OK : WH_ERR_HIP; }

static bool is_pinned_host(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// copy with several host threads: one core moves ~10 GB/s, PCIe Gen5 x16 takes ~55
static void parallel_copy(uint8_t *dst, const uint8_t *src, uint64_t bytes, unsigned threads) {
    if (bytes < (8u << 20) || threads < 2) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const uint64_t per = (bytes / threads + 4095) & ~4095ull;
    for (unsigned t = 0; t < threads; t++) {
        const uint64_t lo = (uint64_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
        if (lo >= hi) break;
        th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto &x : th) x.join();
}

// Host buffers in, host buffers out: the batch crosses PCIe in chunks (default 2 GiB, WH_HOST_CHUNK_MB) through two device
// buffers, the copy of chunk k + 1 (copy stream) under the verification of chunk k (context stream).  Pageable input is
// first moved into two pinned staging buffers by several host threads (that copy of chunk k + 1 runs under the DMA of chunk
// k); input that already is pinned (wh_host_alloc, hipHostRegister) goes straight to the DMA engine.
int32_t wh_prio3_prepare_batch(wh_ctx *c, const uint8_t verify_key[16], const uint8_t *blob, const uint64_t *offsets,
                               uint64_t n, uint8_t *tags, uint8_t *ids, int32_t *status) {
    if (!c || !verify_key || (n && (!blob || !offsets || !tags || !ids || !status))) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (n == 0) { c->n_last = 0; c->d_status_last = nullptr; c->producer = wh_ctx::PREPARE_BATCH; return WH_OK; }
    for (uint64_t i = 0; i < n; i++) if (offsets[i + 1] < offsets[i]) return WH_ERR_PARAM;
    int32_t rc;
    uint64_t chunk_bytes = 2048ull << 20;
    if (const uint64_t v = c->knobs.v[WH_OPT_HOST_CHUNK_MB]) chunk_bytes = v << 20;
    // chunk boundaries: whole reports, at most chunk_bytes each (a single larger report is a chunk of its own)
    std::vector<uint64_t> first{0};
    uint64_t max_bytes = 0, max_reports = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i + 1;
        while (j < n && offsets[j + 1] - offsets[i] <= chunk_bytes) j++;
        max_bytes = std::max(max_bytes, offsets[j] - offsets[i]);
        max_reports = std::max(max_reports, j - i);
        first.push_back(j);
        i = j;
    }
    const uint64_t nchunks = first.size() - 1;
    const bool pinned_in = is_pinned_host(blob);
    const uint64_t nbuf = nchunks > 1 ? 2 : 1;
    const uint64_t dev_stride = (max_bytes + 64 + 255) & ~255ull, off_stride = max_reports + 1;
    if ((rc = ensure(c, c->d_blob, c->blob_cap, nbuf * dev_stride))) return rc;
    if ((rc = ensure(c, c->d_offsets, c->off_cap, nbuf * off_stride))) return rc;
    if ((rc = ensure(c, c->d_tags, c->tags_cap, 16 * n))) return rc;
    if ((rc = ensure(c, c->d_ids, c->ids_cap, 16 * n))) return rc;
    if ((rc = ensure(c, c->d_status_own, c->status_cap, n))) return rc;
    // pinned staging: the relative offsets of both chunks in flight, and (pageable input only) their bytes
    const uint64_t stage_bytes = nbuf * off_stride * 8 + (pinned_in ? 0 : nbuf * dev_stride);
    if (stage_bytes > c->h_stage_cap) {
        if (c->h_stage) HIPCHK(c, hipHostFree(c->h_stage));
        c->h_stage = nullptr; c->h_stage_cap = 0;
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_stage), stage_bytes, hipHostMallocDefault));
        c->h_stage_cap = stage_bytes;
    }
    uint64_t *h_off = reinterpret_cast<uint64_t *>(c->h_stage);
    uint8_t *h_bytes = c->h_stage + nbuf * off_stride * 8;
    PrepPlan pl;
    if ((rc = prepare_begin(c, verify_key, n, max_reports, c->d_status_own, pl))) return rc;
    unsigned threads = std::thread::hardware_concurrency();
    if (const uint64_t v = c->knobs.v[WH_OPT_HOST_COPY_THREADS]) threads = (unsigned)v;
    threads = threads < 1 ? 1 : (threads > 8 ? 8 : threads);
    bool used[2] = {false, false};
    // after the first enqueue an error must not return while the copy stream may still read the caller's blob (pinned input)
    // or the staging buffers, nor leave work queued behind a context that reports "no batch": drain, then return
    auto drain = [&](int32_t code) {
        (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->stream2); (void)hipStreamSynchronize(c->stream3);
        return code;
    };
#define HIPCHK_DRAIN(call)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) { c->err = std::string(#call) + ": " + hipGetErrorString(e_); return drain(WH_ERR_HIP); } \
    } while (0)
    for (uint64_t k = 0; k < nchunks; k++) {
        const uint64_t b = k % nbuf, lo = first[k], hi = first[k + 1], bytes = offsets[hi] - offsets[lo];
        // the pinned staging of this buffer was last read by the copy of chunk k - 2
        if (used[b]) HIPCHK_DRAIN(hipEventSynchronize(c->ev_h2d[b]));
        uint64_t *rel = h_off + b * off_stride;
        for (uint64_t i = lo; i <= hi; i++) rel[i - lo] = offsets[i] - offsets[lo];
        const uint8_t *src = blob + offsets[lo];
        if (!pinned_in) { parallel_copy(h_bytes + b * dev_stride, src, bytes, threads); src = h_bytes + b * dev_stride; }
        // the device buffer was last read by the verification of chunk k - 2
        if (used[b]) HIPCHK_DRAIN(hipStreamWaitEvent(c->copy_stream, c->ev_done[b], 0));
        HIPCHK_DRAIN(hipMemcpyAsync(c->d_blob + b * dev_stride, src, bytes, hipMemcpyHostToDevice, c->copy_stream));
        HIPCHK_DRAIN(hipMemcpyAsync(c->d_offsets + b * off_stride, rel, (hi - lo + 1) * 8, hipMemcpyHostToDevice, c->copy_stream));
        HIPCHK_DRAIN(hipEventRecord(c->ev_h2d[b], c->copy_stream));
        HIPCHK_DRAIN(hipStreamWaitEvent(c->stream, c->ev_h2d[b], 0));
        if ((rc = prepare_range(c, pl, c->d_blob + b * dev_stride, c->d_offsets + b * off_stride, hi - lo, lo, c->d_tags, c->d_ids,
                                c->d_status_own))) return drain(rc);
        HIPCHK_DRAIN(hipEventRecord(c->ev_done[b], c->stream));
        used[b] = true;
    }
#undef HIPCHK_DRAIN
    if ((rc = prepare_end(c, c->d_status_own, n))) return rc;
    HIPCHK(c, hipMemcpyAsync(tags, c->d_tags, 16 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(ids, c->d_ids, 16 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(status, c->d_status_own, 4 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

int32_t wh_prio3_out_shares(wh_ctx *c, uint8_t *out) {
    if (!c || !out) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (c->producer == wh_ctx::NONE) return WH_ERR_STATE;
    if (!c->n_last) return WH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->d_out_shares, c->n_last * c->z.output_len * sizeof(F128), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

// keep_eff[i] = keep[i] && status[i] == 0
__global__ void k_keep_eff(const uint8_t *keep, const int32_t *status, uint64_t n, uint8_t *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (status[i] == 0) && (!keep || keep[i]);
}

// d_out[len] = (extra ? extra : 0) + sum of the kept rows, mod p.  extra may be d_out itself (accumulate in place).
static int32_t sum_rows(wh_ctx *c, const F128 *rows, const uint8_t *d_keep, uint64_t n, uint32_t len, F128 *d_out, const F128 *extra = nullptr) {
    int32_t rc;
    if (n == 0) {
        if (!extra) HIPCHK(c, hipMemsetAsync(d_out, 0, len * sizeof(F128), c->stream));
        else if (extra != d_out) HIPCHK(c, hipMemcpyAsync(d_out, extra, len * sizeof(F128), hipMemcpyDeviceToDevice, c->stream));
        return WH_OK;
    }
    KTimer t_(c, WH_K_AGG);
    if (len >= 4096) {
        // long rows (sketches): the columns alone fill the chip; a few slabs of rows on top
        const uint32_t gx = (len + 255) / 256;
        uint64_t gy = 4096 / gx; if (gy < 1) gy = 1; if (gy > n) gy = n; if (gy > 64) gy = 64;
        if (gy == 1) {
            hipLaunchKernelGGL(k_agg_cols_direct, dim3(gx), dim3(256), 0, c->stream, rows, d_keep, n, len, extra, d_out);
            HIPCHK(c, hipGetLastError());
            return WH_OK;
        }
        const uint64_t rps = (n + gy - 1) / gy;
        gy = (n + rps - 1) / rps;
        if ((rc = ensure(c, c->d_partials, c->partials_cap, gy * len))) return rc;
        hipLaunchKernelGGL(k_agg_partial_cols, dim3(gx, (uint32_t)gy), dim3(256), 0, c->stream, rows, d_keep, n, len, (uint32_t)rps, c->d_partials);
        hipLaunchKernelGGL(k_agg_final, dim3((len + 15) / 16), dim3(256), 0, c->stream, c->d_partials, (uint32_t)gy, len, extra, d_out);
        HIPCHK(c, hipGetLastError());
        return WH_OK;
    }
    const uint32_t tpr = len < 256 ? len : 256, rstep = 256 / tpr;
    // ~2048 blocks, each a slab of rows
    uint64_t rpb = (n + 2047) / 2048;
    if (rpb < rstep) rpb = rstep;
    const uint32_t nblocks = (uint32_t)((n + rpb - 1) / rpb);
    if ((rc = ensure(c, c->d_partials, c->partials_cap, (uint64_t)nblocks * len))) return rc;
    hipLaunchKernelGGL(k_agg_partial, dim3(nblocks), dim3(256), 0, c->stream, rows, d_keep, n, len, (uint32_t)rpb, c->d_partials);
    hipLaunchKernelGGL(k_agg_final, dim3((len + 15) / 16), dim3(256), 0, c->stream, c->d_partials, nblocks, len, extra, d_out);
    HIPCHK(c, hipGetLastError());
    return WH_OK;
}

int32_t wh_prio3_aggregate_dev(wh_ctx *c, const uint8_t *d_keep, uint8_t *d_agg) {
    if (!c || !d_agg) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    // only a prepare call leaves something to aggregate (not Type::query, not the sharder, which reuse the workspace)
    if (c->producer != wh_ctx::PREPARE_BATCH && c->producer != wh_ctx::PREPARE_INIT) return WH_ERR_STATE;
    // draft-07: out shares exist once prepare_next has accepted the prepare message; before wh_prio3_prepare_finish_batch the
    // statuses are prepare_init's and nothing has been decided
    if (c->producer == wh_ctx::PREPARE_INIT && !c->init_finished) { c->err = "aggregate: the prepare_init batch has not been through prepare_finish"; return WH_ERR_STATE; }
    const uint64_t n = c->n_last;
    int32_t rc;
    if (n) {
        if (!c->d_status_last) return WH_ERR_STATE;
        if ((rc = ensure(c, c->d_keep_eff, c->keep_cap, n))) return rc;
        hipLaunchKernelGGL(k_keep_eff, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, d_keep, c->d_status_last, n, c->d_keep_eff);
    }
    return sum_rows(c, c->d_out_shares, c->d_keep_eff, n, c->z.output_len, reinterpret_cast<F128 *>(d_agg));
}

int32_t wh_prio3_aggregate(wh_ctx *c, const uint8_t *keep, uint8_t *agg) {
    if (!c || !agg) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->producer != wh_ctx::PREPARE_BATCH && c->producer != wh_ctx::PREPARE_INIT) return WH_ERR_STATE;
    if (c->producer == wh_ctx::PREPARE_INIT && !c->init_finished) { c->err = "aggregate: the prepare_init batch has not been through prepare_finish"; return WH_ERR_STATE; }
    const uint64_t n = c->n_last, len = c->z.output_len;
    int32_t rc;
    if ((rc = ensure(c, c->d_tmp_a, c->tmp_a_cap, len * 16))) return rc;
    uint8_t *d_keep = nullptr, *d_agg = c->d_tmp_a;
    if (keep && n) {
        if ((rc = ensure(c, c->d_tmp_b, c->tmp_b_cap, n))) return rc;
        d_keep = c->d_tmp_b;
        HIPCHK(c, hipMemcpyAsync(d_keep, keep, n, hipMemcpyHostToDevice, c->stream));
    }
    if ((rc = wh_prio3_aggregate_dev(c, d_keep, d_agg))) return rc;
    HIPCHK(c, hipMemcpyAsync(agg, d_agg, len * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

int32_t wh_f128_sum_rows_dev(wh_ctx *c, const uint8_t *d_rows, uint64_t n_rows, uint32_t len, const uint8_t *d_keep,
                             uint8_t *d_out) {
    if (!c || !d_out || (n_rows && !d_rows) || !len) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    return sum_rows(c, reinterpret_cast<const F128 *>(d_rows), d_keep, n_rows, len, reinterpret_cast<F128 *>(d_out));
}

// acc[words] += sum of the kept rows of buckets[n][words] (no zeroing), on stream st
// `beside`: the launch shares the chip with the heavy-hitter proof kernels on the main stream, which are chains of latencies
// with few wavefronts (k_hh_final: 219 for 14 000 key chains).  Left alone, the sums' 4 000 workgroups per launch fill every
// wave slot and those kernels' workgroups wait behind them (k_hh_final 1.0 -> 2.2 ms).  Here every workgroup asks for
// 120 KiB of LDS it does not use: one per CU, four wavefronts with eight vectors in flight per lane; the sums then take
// 3.0 instead of 2.2 ms, still under the proof kernels, and those run at their own pace (profiles/r2d_bench_hh_*.json).
static int32_t hh_sum_on(wh_ctx *c, hipStream_t st, const uint32_t *d_buckets, const uint8_t *d_keep, uint64_t n, uint64_t words, uint32_t *d_acc,
                         bool beside = false) {
    if (!n) return WH_OK;
    const uint32_t beside_lds = (uint32_t)c->knobs.v[WH_OPT_HH_SUM_LDS];
    const uint32_t gx = (uint32_t)((words + 4 * 256 - 1) / (4 * 256));
    // enough (x, y) blocks to fill the chip: target ~4096 blocks
    uint32_t gy = 4096 / gx; if (gy < 1) gy = 1; if (gy > n) gy = (uint32_t)n; if (gy > 65535) gy = 65535;
    const uint32_t cpb = (uint32_t)((n + gy - 1) / gy);
    gy = (uint32_t)((n + cpb - 1) / cpb);
    if (beside && beside_lds)
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_hh_sum_u32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)beside_lds));
    hipLaunchKernelGGL(k_hh_sum_u32, dim3(gx, gy), dim3(256), beside ? beside_lds : 0, st, d_buckets, d_keep, n, words, cpb, d_acc);
    HIPCHK(c, hipGetLastError());
    return WH_OK;
}
static int32_t hh_sum(wh_ctx *c, const uint32_t *d_buckets, const uint8_t *d_keep, uint64_t n, uint64_t words,
                      uint32_t *d_acc, bool zero_first) {
    if (!c || !d_acc || !words || (n && !d_buckets)) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (zero_first) HIPCHK(c, hipMemsetAsync(d_acc, 0, words * 4, c->stream));
    KTimer t_(c, WH_K_HH);
    return hh_sum_on(c, c->stream, d_buckets, d_keep, n, words, d_acc);
}
// the same sums over a Field128 ring (BASELINE config 5 names it; the reference's AggRing is u32, hhcore/src/lib.rs:13)
static int32_t hh_sum_f128(wh_ctx *c, const uint8_t *d_buckets, const uint8_t *d_keep, uint64_t n, uint64_t words, uint8_t *d_acc, bool accumulate) {
    if (!c || !d_acc || !words || words > 0xffffffffull || (n && !d_buckets)) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    F128 *acc = reinterpret_cast<F128 *>(d_acc);
    return sum_rows(c, reinterpret_cast<const F128 *>(d_buckets), d_keep, n, (uint32_t)words, acc, accumulate ? acc : nullptr);
}
int32_t wh_hh_bucket_sum_f128_dev(wh_ctx *c, const uint8_t *d_buckets, const uint8_t *d_keep, uint64_t n, uint64_t words, uint8_t *d_acc) {
    return hh_sum_f128(c, d_buckets, d_keep, n, words, d_acc, false);
}
int32_t wh_hh_bucket_accumulate_f128_dev(wh_ctx *c, const uint8_t *d_buckets, const uint8_t *d_keep, uint64_t n, uint64_t words, uint8_t *d_acc) {
    return hh_sum_f128(c, d_buckets, d_keep, n, words, d_acc, true);
}
int32_t wh_hh_bucket_sum_u32_dev(wh_ctx *c, const uint32_t *d_buckets, const uint8_t *d_keep, uint64_t n, uint64_t words,
                                 uint32_t *d_acc) {
    return hh_sum(c, d_buckets, d_keep, n, words, d_acc, true);
}
int32_t wh_hh_bucket_accumulate_u32_dev(wh_ctx *c, const uint32_t *d_buckets, const uint8_t *d_keep, uint64_t n, uint64_t words,
                                        uint32_t *d_acc) {
    return hh_sum(c, d_buckets, d_keep, n, words, d_acc, false);
}

// ---- heavy-hitter key chains ----
int32_t wh_hh_ctx_create(int32_t device_id, wh_ctx **out) {
    // the heavy-hitter entry points use the context's device, stream and buffers only; the Prio3 tables are those of
    // the smallest circuit
    const wh_prio3_params p = {WH_HISTOGRAM, 0, 2, 1, 2, 0};
    return wh_ctx_create(device_id, &p, out);
}
uint64_t wh_hh_key_chain_len(uint32_t depth) { return depth >= 1 && depth <= HH_MAX_DEPTH ? HhLayout(depth).total : 0; }

static int32_t hh_tables(wh_ctx *c, const uint8_t verify_key[16]) {
    if (c->hh_tab_valid && !memcmp(c->hh_vk, verify_key, 16)) return WH_OK;
    if (!c->d_hh_tab) HIPCHK(c, hipMalloc((void **)&c->d_hh_tab, sizeof(HhTables)));
    HhTables t;
    memcpy(t.te0, aes_tables_host().te0, sizeof t.te0);
    uint8_t k[16];
    memset(k, 0, 16); const AesKey k0(k);
    memset(k, 1, 16); const AesKey k1(k);
    const AesKey kv(verify_key);
    memcpy(t.rk[0], k0.rk, sizeof k0.rk); memcpy(t.rk[1], k1.rk, sizeof k1.rk); memcpy(t.rk[2], kv.rk, sizeof kv.rk);
    for (int x = 0; x < 2; x++) {
        uint8_t b[16] = {0}, s0[16], s2[16];
        b[0] = (uint8_t)x;
        k0.mmo(b, s0); k0.mmo(s0, s2);
        memcpy(&t.cx[x][0], s0, 16); memcpy(&t.cx[x][4], s2, 16);
    }
    HIPCHK(c, hipMemcpyAsync(c->d_hh_tab, &t, sizeof t, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(c->hh_vk, verify_key, 16);
    c->hh_tab_valid = true;
    return WH_OK;
}

static int32_t hh_workspace(wh_ctx *c, uint64_t cap, uint32_t depth) {
    if (cap <= c->hh_cap && depth <= c->hh_cap_depth) return WH_OK;
    if (cap < c->hh_cap) cap = c->hh_cap;
    if (depth < c->hh_cap_depth) depth = c->hh_cap_depth;
    void *ptrs[] = {c->hh_w.tpc, c->hh_w.proofblk, c->hh_w.supd, c->hh_w.buckg};
    for (void *p : ptrs) if (p) HIPCHK(c, hipFree(p));
    c->hh_w = HhWork{}; c->hh_cap = 0; c->hh_cap_depth = 0;
    HIPCHK(c, hipMalloc((void **)&c->hh_w.tpc, cap * (64ull << depth)));
    HIPCHK(c, hipMalloc((void **)&c->hh_w.proofblk, cap * (HH_SUPPORT + 1) * 64ull));
    HIPCHK(c, hipMalloc((void **)&c->hh_w.supd, cap * HH_SUPPORT * 8ull));
    HIPCHK(c, hipMalloc((void **)&c->hh_w.buckg, cap * HH_BKT * 4ull));
    c->hh_cap = cap; c->hh_cap_depth = depth;
    return WH_OK;
}

// eval_and_check of n key chains, sub-batch by sub-batch.  d_dat != NULL: the bucket vectors are materialised there.
// d_sketch != NULL: they are summed into sketch (index / group_size) -- from a sub-batch of materialised vectors in context
// scratch, by k_hh_sum_u32 on the side stream UNDER the latency-bound support / chain / final kernels of the same sub-batch
// (summing inside the expansion with atomics is three times slower: vdpf.hip, HhSketchOut).
static int32_t hh_eval(wh_ctx *c, const uint8_t verify_key[16], const uint8_t *d_keys, uint64_t stride, uint64_t n, uint32_t depth,
                       uint32_t *d_dat, uint32_t *d_sketch, uint64_t group_size, int32_t accumulate, uint8_t *d_tags, uint8_t *d_ids,
                       int32_t *d_status) {
    HIPCHK(c, hipSetDevice(c->device));
    int32_t rc;
    if ((rc = hh_tables(c, verify_key))) return rc;
    const uint64_t words = (uint64_t)HH_BKT << depth;
    // sub-batches bound the scratch (64 B per leaf for the proof chain): at most 16384 key chains or 2 GiB at a time
    uint64_t cap = (2ull << 30) / (64ull << depth);
    if (cap > 16384) cap = 16384;
    if (const uint64_t v = c->knobs.v[WH_OPT_HH_SUBBATCH]) {       // tuning / test knob: key chains per sub-batch
        if (v < cap) cap = v;
    }
    if (cap > n) cap = n ? n : 1;
    if ((rc = hh_workspace(c, cap, depth))) return rc;
    const bool sketch = d_sketch && n;
    if (sketch) {
        if ((rc = ensure(c, c->d_hh_part, c->hh_part_cap, cap * words))) return rc;       // one sub-batch of bucket vectors
        const uint64_t n_sk = (n + group_size - 1) / group_size;
        if (!accumulate) HIPCHK(c, hipMemsetAsync(d_sketch, 0, n_sk * words * 4, c->stream));
    }
    for (uint64_t base = 0; base < n; base += cap) {
        const uint32_t m = (uint32_t)(n - base < cap ? n - base : cap);
        const uint8_t *kp = d_keys + base * stride;
        uint32_t *dat_b = sketch ? c->d_hh_part : (d_dat ? d_dat + base * words : nullptr);
        // the scratch vectors of the sub-batch before are still being summed on the side stream
        if (sketch && base) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        int e;
        { KTimer t_(c, WH_K_HH_MAIN);
          e = launch_hh_main(kp, stride, m, depth, c->d_hh_tab, c->hh_w, dat_b, d_ids + 16 * base, d_status + base, HhSketch{}, c->stream); }
        if (e) { c->err = std::string("heavy-hitter eval launch: ") + hipGetErrorString((hipError_t)e); return WH_ERR_HIP; }
        if (sketch) {
            HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            KTimer t_(c, WH_K_HH, c->stream2);
            for (uint64_t lo = base; lo < base + m;) {           // the stretch of each sketch inside this sub-batch
                const uint64_t g = lo / group_size, hi = std::min<uint64_t>(base + m, (g + 1) * group_size);
                if ((rc = hh_sum_on(c, c->stream2, c->d_hh_part + (lo - base) * words, nullptr, hi - lo, words, d_sketch + g * words, true))) return rc;
                lo = hi;
            }
            HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
        }
        { KTimer t_(c, WH_K_HH_SUPPORT); e = launch_hh_support(kp, stride, m, depth, c->d_hh_tab, c->hh_w, d_status + base, c->stream); }
        if (!e) { KTimer t_(c, WH_K_HH_CHAIN); e = launch_hh_chain(kp, stride, m, depth, c->d_hh_tab, c->hh_w, d_status + base, c->stream); }
        if (!e) { KTimer t_(c, WH_K_HH_FINAL); e = launch_hh_final(kp, stride, m, depth, c->hh_w, d_tags + 64 * base, d_status + base, c->stream); }
        if (e) { c->err = std::string("heavy-hitter eval launch: ") + hipGetErrorString((hipError_t)e); return WH_ERR_HIP; }
    }
    if (sketch) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    return WH_OK;
}

int32_t wh_hh_eval_and_check_batch_dev(wh_ctx *c, const uint8_t verify_key[16], const uint8_t *d_keys, uint64_t stride, uint64_t n,
                                       uint32_t depth, uint32_t *d_dat, uint8_t *d_tags, uint8_t *d_ids, int32_t *d_status) {
    if (!c || !verify_key || depth < 1 || depth > HH_MAX_DEPTH || (n && (!d_keys || !d_tags || !d_ids || !d_status))) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (stride < HhLayout(depth).total || n > 0x7fffffffull) return WH_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    int32_t rc;
    const uint64_t words = (uint64_t)HH_BKT << depth;
    if (!d_dat) {
        if ((rc = ensure(c, c->d_hh_dat, c->hh_dat_cap, (n ? n : 1) * words))) return rc;
        d_dat = c->d_hh_dat;
    }
    c->d_hh_dat_last = d_dat; c->hh_n_last = n; c->hh_depth_last = depth;
    return hh_eval(c, verify_key, d_keys, stride, n, depth, d_dat, nullptr, 0, 0, d_tags, d_ids, d_status);
}

int32_t wh_hh_eval_and_sketch_batch_dev(wh_ctx *c, const uint8_t verify_key[16], const uint8_t *d_keys, uint64_t stride, uint64_t n,
                                        uint32_t depth, uint64_t group_size, uint32_t *d_sketch, int32_t accumulate, uint8_t *d_tags,
                                        uint8_t *d_ids, int32_t *d_status) {
    if (!c || !verify_key || depth < 1 || depth > HH_MAX_DEPTH || !group_size || (n && (!d_keys || !d_sketch || !d_tags || !d_ids || !d_status)))
        return WH_ERR_PARAM;
    WH_LOCK(c);
    if (stride < HhLayout(depth).total || n > 0x7fffffffull) return WH_ERR_PARAM;
    c->d_hh_dat_last = nullptr; c->hh_n_last = 0;          // nothing materialised: wh_hh_buckets / wh_hh_aggregate have no batch
    return hh_eval(c, verify_key, d_keys, stride, n, depth, nullptr, d_sketch, group_size, accumulate, d_tags, d_ids, d_status);
}

// "sum all, then subtract the bad ones": re-expand the key chains the group test rejected and take their bucket vectors out
// of the sketches they went into (exact in Z/2^32).  Only the buckets are recomputed (no proof chain, no tags).
int32_t wh_hh_sketch_subtract_dev(wh_ctx *c, const uint8_t verify_key[16], const uint8_t *d_keys, uint64_t stride, uint32_t depth,
                                  const uint64_t *bad_index, uint64_t n_bad, uint64_t group_size, uint32_t *d_sketch) {
    if (!c || !verify_key || depth < 1 || depth > HH_MAX_DEPTH || !group_size || (n_bad && (!d_keys || !bad_index || !d_sketch))) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (stride < HhLayout(depth).total || n_bad > 0x7fffffffull) return WH_ERR_PARAM;
    if (!n_bad) return WH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int32_t rc;
    if ((rc = hh_tables(c, verify_key))) return rc;
    if ((rc = ensure(c, c->d_tmp_a, c->tmp_a_cap, 4 * n_bad))) return rc;
    std::vector<uint32_t> idx(n_bad);
    for (uint64_t i = 0; i < n_bad; i++) { if (bad_index[i] > 0x7fffffffull) return WH_ERR_PARAM; idx[i] = (uint32_t)bad_index[i]; }
    HIPCHK(c, hipMemcpyAsync(c->d_tmp_a, idx.data(), 4 * n_bad, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));            // `idx` is a local
    HhSketch sk;
    sk.sketch = d_sketch; sk.group_size = group_size; sk.kidx = reinterpret_cast<const uint32_t *>(c->d_tmp_a); sk.flip = 1; sk.buckets_only = true;
    int e;
    { KTimer t_(c, WH_K_HH_MAIN);
      e = launch_hh_main(d_keys, stride, (uint32_t)n_bad, depth, c->d_hh_tab, c->hh_w, nullptr, nullptr, nullptr, sk, c->stream); }
    if (e) { c->err = std::string("sketch subtract launch: ") + hipGetErrorString((hipError_t)e); return WH_ERR_HIP; }
    return WH_OK;
}

int32_t wh_hh_eval_and_check_batch(wh_ctx *c, const uint8_t verify_key[16], const uint8_t *keys, uint64_t stride, uint64_t n,
                                   uint32_t depth, uint8_t *tags, uint8_t *testing_ids, int32_t *status) {
    if (!c || !verify_key || (n && (!keys || !tags || !testing_ids || !status))) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    int32_t rc;
    const uint64_t m = n ? n : 1;
    if ((rc = ensure(c, c->d_hh_keys, c->hh_keys_cap, m * stride))) return rc;
    if ((rc = ensure(c, c->d_hh_tags, c->hh_tags_cap, m * 64))) return rc;
    if ((rc = ensure(c, c->d_hh_ids, c->hh_ids_cap, m * 16))) return rc;
    if ((rc = ensure(c, c->d_hh_status, c->hh_status_cap, m))) return rc;
    if (n) HIPCHK(c, hipMemcpyAsync(c->d_hh_keys, keys, n * stride, hipMemcpyHostToDevice, c->stream));
    if ((rc = wh_hh_eval_and_check_batch_dev(c, verify_key, c->d_hh_keys, stride, n, depth, nullptr, c->d_hh_tags, c->d_hh_ids,
                                             c->d_hh_status))) return rc;
    if (n) {
        HIPCHK(c, hipMemcpyAsync(tags, c->d_hh_tags, n * 64, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(testing_ids, c->d_hh_ids, n * 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(status, c->d_hh_status, n * 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

int32_t wh_hh_buckets(wh_ctx *c, uint32_t *dat) {
    if (!c || !dat) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (!c->d_hh_dat_last) return WH_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dat, c->d_hh_dat_last, c->hh_n_last * ((uint64_t)HH_BKT << c->hh_depth_last) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

int32_t wh_hh_aggregate_dev(wh_ctx *c, const uint8_t *d_keep, uint32_t *d_acc, int32_t accumulate) {
    if (!c || !d_acc) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (!c->d_hh_dat_last) return WH_ERR_STATE;
    return hh_sum(c, c->d_hh_dat_last, d_keep, c->hh_n_last, (uint64_t)HH_BKT << c->hh_depth_last, d_acc, !accumulate);
}

int32_t wh_hh_aggregate(wh_ctx *c, const uint8_t *keep, uint32_t *acc) {
    if (!c || !acc) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (!c->d_hh_dat_last) return WH_ERR_STATE;
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t n = c->hh_n_last, words = (uint64_t)HH_BKT << c->hh_depth_last;
    int32_t rc;
    if ((rc = ensure(c, c->d_hh_acc, c->hh_acc_cap, words))) return rc;
    const uint8_t *d_keep = nullptr;
    if (keep && n) {
        if ((rc = ensure(c, c->d_hh_keep, c->hh_keep_cap, n))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->d_hh_keep, keep, n, hipMemcpyHostToDevice, c->stream));
        d_keep = c->d_hh_keep;
    }
    if ((rc = wh_hh_aggregate_dev(c, d_keep, c->d_hh_acc, 0))) return rc;
    HIPCHK(c, hipMemcpyAsync(acc, c->d_hh_acc, words * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

int32_t wh_ctx_set_option(wh_ctx *c, int32_t option, uint64_t value) {
    if (!c || option < 0 || option >= WH_NUM_OPTS) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (option == WH_OPT_HH_SUM_LDS && value > (160u << 10)) return WH_ERR_PARAM;
    c->knobs.v[option] = value;
    return WH_OK;
}
int32_t wh_ctx_get_option(wh_ctx *c, int32_t option, uint64_t *value) {
    if (!c || !value || option < 0 || option >= WH_NUM_OPTS) return WH_ERR_PARAM;
    WH_LOCK(c);
    *value = c->knobs.v[option];
    return WH_OK;
}

int32_t wh_ctx_set_profiling(wh_ctx *c, int32_t on) {
    if (!c) return WH_ERR_PARAM;
    WH_LOCK(c);
    c->profiling = on != 0;
    return WH_OK;
}

int32_t wh_ctx_kernel_times(wh_ctx *c, double *ms, uint64_t *launches, int32_t reset) {
    if (!c || !ms || !launches) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto &e : c->evs) {
        float t = 0;
        if (hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) { c->k_ms[e.id] += t; c->k_launches[e.id]++; }
        c->ev_pool.push_back(e.a); c->ev_pool.push_back(e.b);
    }
    c->evs.clear();
    for (int i = 0; i < WH_NUM_KERNELS; i++) {
        ms[i] = c->k_ms[i]; launches[i] = c->k_launches[i];
        if (reset) { c->k_ms[i] = 0; c->k_launches[i] = 0; }
    }
    return WH_OK;
}

int32_t wh_ctx_set_test_sampler_bound(wh_ctx *c, const uint8_t bound[16]) {
    if (!c) return WH_ERR_PARAM;
    WH_LOCK(c);
    F128 b = F128{P_LO, P_HI};
    if (bound) {
        F128 v;
        memcpy(&v, bound, 16);
        if (f_in_range(v.lo, v.hi)) b = v;
    }
    c->P.sampler_bound = b;
    return WH_OK;
}

int32_t wh_test_f128_op(wh_ctx *c, int32_t op, const uint8_t *a, const uint8_t *b, uint8_t *out, uint64_t n) {
    if (!c || !a || !out || !n) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    int32_t rc;
    if ((rc = ensure(c, c->d_tmp_a, c->tmp_a_cap, 16 * n))) return rc;
    if ((rc = ensure(c, c->d_tmp_b, c->tmp_b_cap, 16 * n))) return rc;
    if ((rc = ensure(c, c->d_tmp_c, c->tmp_c_cap, 16 * n))) return rc;
    F128 *da = reinterpret_cast<F128 *>(c->d_tmp_a), *db = reinterpret_cast<F128 *>(c->d_tmp_b), *dout = reinterpret_cast<F128 *>(c->d_tmp_c);
    HIPCHK(c, hipMemcpyAsync(da, a, 16 * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(db, b ? b : a, 16 * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_test_f128, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, da, db, dout, (uint32_t)n, op);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, dout, 16 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

int32_t wh_ntt_f128_batch(wh_ctx *c, const uint8_t *in, uint64_t n_in, uint64_t size, int32_t inverse, uint64_t n, uint8_t *out) {
    if (!c || !out || !size || (size & (size - 1)) || n_in > size || (n && n_in && !in)) return WH_ERR_PARAM;
    WH_LOCK(c);
    if (n > 0x7fffffffull) return WH_ERR_PARAM;
    if (size > (1ull << 20)) return WH_ERR_PARAM;             // fft.rs:42-44: FftError::SizeTooLarge (MAX_ROOTS)
    if (size > (1ull << NTT_MAX_LOG2)) { c->err = "transforms above 4096 points do not fit the kernel's LDS"; return WH_ERR_UNSUPPORTED; }
    const uint32_t lg = log2_u32((uint32_t)size);
    if (!n) return WH_OK;
    // decode: every element must be a canonical residue (field.rs:405-423), as for every other entry point
    for (uint64_t k = 0; k < n * n_in; k++) {
        uint64_t lo, hi;
        memcpy(&lo, in + 16 * k, 8); memcpy(&hi, in + 16 * k + 8, 8);
        if (!f_in_range(lo, hi)) return WH_ERR_MODULUS;
    }
    HIPCHK(c, hipSetDevice(c->device));
    int32_t rc;
    const uint64_t in_b = 16 * n * (n_in ? n_in : 1), out_b = 16 * n * size;
    if ((rc = ensure(c, c->d_tmp_a, c->tmp_a_cap, in_b))) return rc;
    if ((rc = ensure(c, c->d_tmp_c, c->tmp_c_cap, out_b))) return rc;
    if (n_in) HIPCHK(c, hipMemcpyAsync(c->d_tmp_a, in, 16 * n * n_in, hipMemcpyHostToDevice, c->stream));
    const int e = launch_ntt_f128(reinterpret_cast<const F128 *>(c->d_tmp_a), (uint32_t)n_in, lg, inverse != 0, root_of_unity(lg),
                                  f_inv(f_small(size)), (uint32_t)n, reinterpret_cast<F128 *>(c->d_tmp_c), c->stream);
    if (e) { c->err = std::string("ntt launch: ") + hipGetErrorString((hipError_t)e); return WH_ERR_HIP; }
    HIPCHK(c, hipMemcpyAsync(out, c->d_tmp_c, out_b, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

int32_t wh_test_f64_op(wh_ctx *c, int32_t op, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n) {
    if (!c || !a || !out || !n) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    int32_t rc;
    if ((rc = ensure(c, c->d_tmp_a, c->tmp_a_cap, 8 * n))) return rc;
    if ((rc = ensure(c, c->d_tmp_b, c->tmp_b_cap, 8 * n))) return rc;
    if ((rc = ensure(c, c->d_tmp_c, c->tmp_c_cap, 8 * n))) return rc;
    uint64_t *da = reinterpret_cast<uint64_t *>(c->d_tmp_a), *db = reinterpret_cast<uint64_t *>(c->d_tmp_b), *dout = reinterpret_cast<uint64_t *>(c->d_tmp_c);
    HIPCHK(c, hipMemcpyAsync(da, a, 8 * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(db, b ? b : a, 8 * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_test_f64, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, da, db, dout, (uint32_t)n, op);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, dout, 8 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

int32_t wh_test_xof_expand(wh_ctx *c, const uint8_t *seeds, uint64_t n, uint32_t alg_id, uint32_t usage, const uint8_t *binder,
                           uint32_t binder_len, uint32_t n_elems, uint8_t *out) {
    if (!c || !seeds || !out || !n || !n_elems) return WH_ERR_PARAM;
    WH_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    int32_t rc;
    if ((rc = ensure(c, c->d_tmp_a, c->tmp_a_cap, 16 * n))) return rc;
    if ((rc = ensure(c, c->d_tmp_b, c->tmp_b_cap, (uint64_t)binder_len + 16))) return rc;
    if ((rc = ensure(c, c->d_tmp_c, c->tmp_c_cap, 16ull * n_elems * n))) return rc;
    uint8_t *ds = c->d_tmp_a, *db = c->d_tmp_b, *dout = c->d_tmp_c;
    HIPCHK(c, hipMemcpyAsync(ds, seeds, 16 * n, hipMemcpyHostToDevice, c->stream));
    if (binder_len) HIPCHK(c, hipMemcpyAsync(db, binder, binder_len, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_test_xof, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, c->stream, ds, alg_id, usage, db, binder_len, n_elems, dout, (uint32_t)n, c->P.sampler_bound);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, dout, 16ull * n_elems * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return WH_OK;
}

}  // extern "C"
