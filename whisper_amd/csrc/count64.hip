// Prio3Count = Prio3<Count<Field64>, XofShake128, 16> (batched-flpcp-rs/src/vdaf/prio3.rs:81): the reference's Field64
// instantiation, un-batched draft-07 preparation as server-base-prio3 runs it (src/main.rs:26-103) for a whole batch.
//
//   Count (flp/types.rs:39-105): one Mul gadget called once, valid = g(x, x) - x; input 1, proof 5, verifier 4, output 1
//   element; no joint randomness, so there is no public share, no blind, and the prepare message is empty.
//   Field64 (fp.rs:359-389): p = 2^64 - 2^32 + 1, 8-byte little-endian canonical residues (field.rs:598-607); the sampler
//   (prng.rs:73-99) walks the XOF output in 8-byte chunks and skips chunks >= p (probability 2^-32 each).
//   VdafKey (vdaf.rs:209-233): agg_id | input share | nonce[16]; Leader share = meas | proof (48 B), Helper share = two seeds.
//
// With m = 2 the wire polynomials are lines through (1, seed_w) and (-1, x), the gadget polynomial has three coefficients,
// and Type::query (flp.rs:359-479) is a handful of Field64 operations: one lane per report does all of it -- parse, the
// Helper's two expansions, the query randomness XOF(verify_key, 5) . nonce, the verifier share.  The work is three to five
// Keccak-f per report; the kernel exists for coverage of the Field64 path, not because it needs a GPU.
#include <hip/hip_runtime.h>
#include <cstring>
#include <mutex>
#include <string>
#include "../../include/whisper_gfx950.h"
#include "keccak.hpp"
#include "f64.hpp"

namespace wh {
int ctx_device(wh_ctx *c);
hipStream_t ctx_stream(wh_ctx *c);
void **ctx_count_slot(wh_ctx *c);
std::recursive_mutex &ctx_mutex(wh_ctx *c);
void ctx_set_error(wh_ctx *c, const std::string &what);
void count64_free(void *state);
}
using namespace wh;

namespace {

constexpr uint32_t C_DST_MEAS = 1, C_DST_PROOF = 2, C_DST_QUERY_RAND = 5;
constexpr uint32_t COUNT_ALG_ID = 0;      // Prio3Count's algorithm id in the XOF domain separation tag

struct Count64 {
    uint8_t *d_blob = nullptr; uint64_t blob_cap = 0;
    uint64_t *d_off = nullptr; uint64_t off_cap = 0;
    uint8_t *d_prep = nullptr; uint64_t prep_cap = 0;        // n * 32: own prepare shares
    uint8_t *d_peer = nullptr; uint64_t peer_cap = 0;        // staging of all aggregators' prepare shares
    uint64_t *d_out = nullptr; uint64_t out_cap = 0;         // n out shares
    int32_t *d_status = nullptr; uint64_t status_cap = 0;
    uint8_t *d_keep = nullptr; uint64_t keep_cap = 0;
    uint64_t *d_part = nullptr;                              // [2 * AGG_BLOCKS + 1]
    uint64_t n_last = 0;
    bool valid = false;
    bool finished = false;      // prepare_finish has decided the batch: only then are there out shares to hand out or sum
    uint64_t bound = P64;                                    // sampler bound (test hook)
};
constexpr uint32_t AGG_BLOCKS = 256;

// Prng::get for Field64 over a sponge that has absorbed its input: n <= 5 elements into out[] (static indices only)
template <int N>
__device__ __forceinline__ void sample64(SpongeG &s, uint64_t (&out)[N], uint32_t n, uint64_t bound) {
    s.finalize();
    uint32_t got = 0;
    for (;;) {
#pragma unroll
        for (int w = 0; w < (int)RATE_WORDS; w++) {
            const uint64_t v = s.st[w];
            const bool take = got < n && v < bound;
#pragma unroll
            for (int k = 0; k < N; k++) if (take && got == (uint32_t)k) out[k] = v;
            got += take;
        }
        if (got >= n) break;
        keccak_f1600(s.st);
    }
}

__device__ __forceinline__ uint64_t poly3(const uint64_t *c, uint64_t x) { return g_add(g_mul(g_add(g_mul(c[2], x), c[1]), x), c[0]); }

// replaces: VdafKey::decode_with_param + Prio3::prepare_init + the out share of prepare_next, per report
__global__ void __launch_bounds__(256) k_count_init(const uint8_t *blob, const uint64_t *offsets, uint64_t total, uint32_t n, uint32_t num_aggregators,
                                                    uint64_t vk0, uint64_t vk1, uint64_t bound, uint8_t *prep, uint64_t *out_shares,
                                                    int32_t *status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *rep = blob + offsets[i];
    // a descending pair, or one that points past the blob, is an empty key (WH_ERR_SHORT), never a read outside the blob
    const uint64_t len = offsets[i + 1] >= offsets[i] && offsets[i + 1] <= total ? offsets[i + 1] - offsets[i] : 0;
    int32_t st = WH_OK;
    uint32_t agg = 0;
    if (len < 1) st = WH_ERR_SHORT;
    if (!st) { agg = rep[0]; if (agg >= num_aggregators) st = WH_ERR_AGG_ID; }
    const uint64_t share_len = agg == 0 ? 48 : 32;
    if (!st && len < 1 + share_len + 16) st = WH_ERR_SHORT;
    uint64_t x = 0, proof[5] = {0, 0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
    if (!st) {
        const uint8_t *share = rep + 1, *nonce = rep + 1 + share_len;
        if (agg == 0) {
            x = ld64(share);
            bool bad = x >= P64;
#pragma unroll
            for (int k = 0; k < 5; k++) { proof[k] = ld64(share + 8 + 8 * k); bad |= proof[k] >= P64; }
            if (bad) st = WH_ERR_MODULUS;                                          // FieldError::ModulusOverflow
        } else {
            SpongeG s;
            uint64_t t1[1];
            s.init_xof(COUNT_ALG_ID, C_DST_MEAS, ld64(share), ld64(share + 8));    // prio3.rs:1758-1775: XOF(seed, 1) . [agg_id]
            s.absorb_byte((uint8_t)agg);
            sample64(s, t1, 1, bound);
            x = t1[0];
            s.init_xof(COUNT_ALG_ID, C_DST_PROOF, ld64(share + 16), ld64(share + 24));
            s.absorb_byte((uint8_t)agg);
            sample64(s, proof, 5, bound);
        }
        if (!st) {
            SpongeG s;
            uint64_t r1[1];
            s.init_xof(COUNT_ALG_ID, C_DST_QUERY_RAND, vk0, vk1);                   // prio3.rs:2232-2239
#pragma unroll 1
            for (int b = 0; b < 16; b++) s.absorb_byte(nonce[b]);
            sample64(s, r1, 1, bound);
            const uint64_t r = r1[0];
            if (g_mul(r, r) == 1) st = WH_ERR_QUERY_RAND;                          // r^m == 1, m = 2 (flp.rs:413-419)
            else {
                // inv(2) = (p + 1) / 2
                const uint64_t half = 0x7fffffff80000001ULL;
                v[0] = g_sub(poly3(proof + 2, P64 - 1), x);                        // QueryShimGadget: G at w^1 = -1, minus x
#pragma unroll
                for (int w = 0; w < 2; w++) {
                    const uint64_t c0 = g_mul(g_add(proof[w], x), half), c1 = g_mul(g_sub(proof[w], x), half);
                    v[1 + w] = g_add(c0, g_mul(c1, r));
                }
                v[3] = poly3(proof + 2, r);
            }
        }
    }
    if (st) { x = 0; v[0] = v[1] = v[2] = v[3] = 0; }
    uint64_t *pr = reinterpret_cast<uint64_t *>(prep) + 4ull * i;
#pragma unroll
    for (int k = 0; k < 4; k++) pr[k] = v[k];
    out_shares[i] = x;
    status[i] = st;
}

// replaces: prepare_shares_to_prepare_message + decide (prio3.rs:2347-2405, flp.rs:483-512), prepare_next's out share
__global__ void __launch_bounds__(256) k_count_finish(const uint8_t *shares, uint32_t n, uint32_t num_aggregators, int32_t *status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || status[i] != WH_OK) return;
    uint64_t v[4] = {0, 0, 0, 0};
    bool bad = false;
    for (uint32_t a = 0; a < num_aggregators; a++) {
        const uint64_t *p = reinterpret_cast<const uint64_t *>(shares) + ((uint64_t)a * n + i) * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint64_t t = p[k]; bad |= t >= P64; v[k] = g_add(v[k], t >= P64 ? 0 : t); }
    }
    if (bad) status[i] = WH_ERR_MODULUS;
    else if (v[0] != 0 || g_mul(v[1], v[2]) != v[3]) status[i] = WH_ERR_DECIDE;
}

// replaces: Aggregator::aggregate over the accepted, kept out shares (prio3.rs:2445-2456).  Sums of canonical residues in
// 128 bits (n < 2^32 terms cannot overflow), one partial per workgroup, then one wavefront folds them mod p.
__global__ void __launch_bounds__(256) k_count_agg_partial(const uint64_t *out_shares, const int32_t *status, const uint8_t *keep, uint64_t n,
                                                           uint64_t *partials) {
    uint64_t lo = 0, hi = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (status[i] == WH_OK && (!keep || keep[i])) { const uint64_t t = out_shares[i]; lo += t; hi += lo < t; }
    __shared__ uint64_t sl[256], sh[256];
    sl[threadIdx.x] = lo; sh[threadIdx.x] = hi;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t a = 0, b = 0;
        for (int k = 0; k < 256; k++) { a += sl[k]; b += sh[k] + (a < sl[k]); }
        partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = b;
    }
}
__global__ void k_count_agg_final(const uint64_t *partials, uint32_t nparts, uint64_t *agg) {
    if (threadIdx.x || blockIdx.x) return;
    uint64_t acc = 0;
    for (uint32_t k = 0; k < nparts; k++) acc = g_add(acc, g_reduce128(partials[2 * k], partials[2 * k + 1]));
    agg[0] = acc;
}

__global__ void k_test_f64_xof(const uint8_t *seeds, uint32_t usage, const uint8_t *binder, uint32_t binder_len, uint32_t n_elems, uint64_t bound,
                               uint32_t n, uint64_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SpongeG s;
    s.init_xof(COUNT_ALG_ID, usage, ld64(seeds + 16ull * i), ld64(seeds + 16ull * i + 8));
    for (uint32_t b = 0; b < binder_len; b++) s.absorb_byte(binder[b]);
    uint64_t o[5] = {0, 0, 0, 0, 0};
    sample64(s, o, n_elems, bound);
    for (uint32_t k = 0; k < n_elems; k++) out[5ull * i + k] = o[k];
}

#define C_HIPCHK(c, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) { ctx_set_error(c, std::string(#call) + ": " + hipGetErrorString(e_)); return WH_ERR_HIP; } \
    } while (0)

template <typename T>
int32_t c_ensure(wh_ctx *c, T *&ptr, uint64_t &cap, uint64_t want) {
    if (want <= cap && ptr) return WH_OK;
    if (ptr) C_HIPCHK(c, hipFree(ptr));
    ptr = nullptr; cap = 0;
    C_HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&ptr), want * sizeof(T)));
    cap = want;
    return WH_OK;
}

Count64 *state_of(wh_ctx *c) {
    void **slot = ctx_count_slot(c);
    if (!*slot) *slot = new Count64();
    return static_cast<Count64 *>(*slot);
}

int32_t count_init_dev(wh_ctx *c, Count64 *S, uint32_t na, const uint8_t vk[16], const uint8_t *d_blob, const uint64_t *d_off, uint64_t total, uint64_t n,
                       uint8_t *d_prep, int32_t *d_status) {
    int32_t rc;
    if ((rc = c_ensure(c, S->d_out, S->out_cap, n ? n : 1))) return rc;
    S->valid = false; S->finished = false; S->n_last = 0;
    uint64_t k0, k1;
    memcpy(&k0, vk, 8); memcpy(&k1, vk + 8, 8);
    if (n) {
        hipLaunchKernelGGL(k_count_init, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx_stream(c), d_blob, d_off, total, (uint32_t)n, na, k0, k1, S->bound,
                           d_prep, S->d_out, d_status);
        C_HIPCHK(c, hipGetLastError());
    }
    return WH_OK;
}

}  // namespace

namespace wh {
void count64_free(void *state) {
    Count64 *S = static_cast<Count64 *>(state);
    if (!S) return;
    void *ptrs[] = {S->d_blob, S->d_off, S->d_prep, S->d_peer, S->d_out, S->d_status, S->d_keep, S->d_part};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete S;
}
}

extern "C" {

int32_t wh_prio3_count_prepare_init_batch(wh_ctx *c, uint32_t num_aggregators, const uint8_t verify_key[16], const uint8_t *blob,
                                          const uint64_t *offsets, uint64_t n, uint8_t *prep_shares, int32_t *status) {
    if (!c || !verify_key || (n && (!blob || !offsets || !prep_shares || !status)) || num_aggregators < 2 || num_aggregators > 254 ||
        n > 0xfffffff0ull) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    C_HIPCHK(c, hipSetDevice(ctx_device(c)));
    Count64 *S = state_of(c);
    const uint64_t nbytes = n ? offsets[n] : 0;
    int32_t rc;
    if ((rc = c_ensure(c, S->d_blob, S->blob_cap, nbytes + 16))) return rc;
    if ((rc = c_ensure(c, S->d_off, S->off_cap, n + 1))) return rc;
    if ((rc = c_ensure(c, S->d_prep, S->prep_cap, 32 * (n ? n : 1)))) return rc;
    if ((rc = c_ensure(c, S->d_status, S->status_cap, n ? n : 1))) return rc;
    hipStream_t st = ctx_stream(c);
    if (n) {
        C_HIPCHK(c, hipMemcpyAsync(S->d_blob, blob, nbytes, hipMemcpyHostToDevice, st));
        C_HIPCHK(c, hipMemcpyAsync(S->d_off, offsets, 8 * (n + 1), hipMemcpyHostToDevice, st));
    }
    if ((rc = count_init_dev(c, S, num_aggregators, verify_key, S->d_blob, S->d_off, nbytes, n, S->d_prep, S->d_status))) return rc;
    if (n) {
        C_HIPCHK(c, hipMemcpyAsync(prep_shares, S->d_prep, 32 * n, hipMemcpyDeviceToHost, st));
        C_HIPCHK(c, hipMemcpyAsync(status, S->d_status, 4 * n, hipMemcpyDeviceToHost, st));
    }
    C_HIPCHK(c, hipStreamSynchronize(st));
    S->n_last = n; S->valid = true; S->finished = n == 0;      // an empty batch has nothing left to decide
    return WH_OK;
}

int32_t wh_prio3_count_prepare_finish_batch(wh_ctx *c, uint32_t num_aggregators, const uint8_t *const *prep_shares, uint64_t n, int32_t *status) {
    if (!c || num_aggregators < 2 || num_aggregators > 254 || (n && (!prep_shares || !status))) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    Count64 *S = state_of(c);
    if (!S->valid || S->n_last != n) { ctx_set_error(c, "wh_prio3_count_prepare_finish_batch: no prepare_init batch of this size in the context"); return WH_ERR_STATE; }
    if (!n) { S->finished = true; return WH_OK; }
    C_HIPCHK(c, hipSetDevice(ctx_device(c)));
    int32_t rc;
    if ((rc = c_ensure(c, S->d_peer, S->peer_cap, 32ull * n * num_aggregators))) return rc;
    hipStream_t st = ctx_stream(c);
    for (uint32_t a = 0; a < num_aggregators; a++) {
        if (!prep_shares[a]) return WH_ERR_PARAM;
        C_HIPCHK(c, hipMemcpyAsync(S->d_peer + 32ull * n * a, prep_shares[a], 32 * n, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(k_count_finish, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, S->d_peer, (uint32_t)n, num_aggregators, S->d_status);
    C_HIPCHK(c, hipGetLastError());
    C_HIPCHK(c, hipMemcpyAsync(status, S->d_status, 4 * n, hipMemcpyDeviceToHost, st));
    C_HIPCHK(c, hipStreamSynchronize(st));
    S->finished = true;
    return WH_OK;
}

int32_t wh_prio3_count_out_shares(wh_ctx *c, uint8_t *out_shares) {
    if (!c) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    Count64 *S = state_of(c);
    if (!S->valid || !S->finished) { ctx_set_error(c, "wh_prio3_count_out_shares: no batch that has been through prepare_init and prepare_finish"); return WH_ERR_STATE; }
    if (!S->n_last) return WH_OK;
    if (!out_shares) return WH_ERR_PARAM;
    C_HIPCHK(c, hipSetDevice(ctx_device(c)));
    C_HIPCHK(c, hipMemcpyAsync(out_shares, S->d_out, 8 * S->n_last, hipMemcpyDeviceToHost, ctx_stream(c)));
    C_HIPCHK(c, hipStreamSynchronize(ctx_stream(c)));
    return WH_OK;
}

int32_t wh_prio3_count_aggregate(wh_ctx *c, const uint8_t *keep, uint8_t agg[8]) {
    if (!c || !agg) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    Count64 *S = state_of(c);
    if (!S->valid || !S->finished) { ctx_set_error(c, "wh_prio3_count_aggregate: no batch that has been through prepare_init and prepare_finish"); return WH_ERR_STATE; }
    C_HIPCHK(c, hipSetDevice(ctx_device(c)));
    hipStream_t st = ctx_stream(c);
    const uint64_t n = S->n_last;
    int32_t rc;
    if (!S->d_part) C_HIPCHK(c, hipMalloc((void **)&S->d_part, 8 * (2 * AGG_BLOCKS + 1)));
    if (keep && n) {
        if ((rc = c_ensure(c, S->d_keep, S->keep_cap, n))) return rc;
        C_HIPCHK(c, hipMemcpyAsync(S->d_keep, keep, n, hipMemcpyHostToDevice, st));
    }
    const uint32_t blocks = n ? (uint32_t)((n + 255) / 256 < AGG_BLOCKS ? (n + 255) / 256 : AGG_BLOCKS) : 0;
    if (blocks) hipLaunchKernelGGL(k_count_agg_partial, dim3(blocks), dim3(256), 0, st, S->d_out, S->d_status, keep ? S->d_keep : nullptr, n, S->d_part);
    hipLaunchKernelGGL(k_count_agg_final, dim3(1), dim3(64), 0, st, S->d_part, blocks, S->d_part + 2 * AGG_BLOCKS);
    C_HIPCHK(c, hipGetLastError());
    C_HIPCHK(c, hipMemcpyAsync(agg, S->d_part + 2 * AGG_BLOCKS, 8, hipMemcpyDeviceToHost, st));
    C_HIPCHK(c, hipStreamSynchronize(st));
    return WH_OK;
}

// test hooks: the Field64 sampler on its own, with any bound (the rejection branch has probability 2^-32 with the real p)
int32_t wh_test_f64_xof_expand(wh_ctx *c, const uint8_t *seeds, uint32_t n, uint32_t usage, const uint8_t *binder, uint32_t binder_len,
                               uint32_t n_elems, uint64_t bound, uint64_t *out) {
    if (!c || !seeds || !out || n_elems < 1 || n_elems > 5 || !n || (binder_len && !binder)) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    C_HIPCHK(c, hipSetDevice(ctx_device(c)));
    uint8_t *d_seeds = nullptr, *d_binder = nullptr; uint64_t *d_out = nullptr;
    hipStream_t st = ctx_stream(c);
    int32_t rc = WH_OK;
    auto fail = [&](hipError_t e, const char *what) { if (e != hipSuccess && rc == WH_OK) { ctx_set_error(c, std::string(what) + ": " + hipGetErrorString(e)); rc = WH_ERR_HIP; } };
    fail(hipMalloc((void **)&d_seeds, 16ull * n), "hipMalloc");
    fail(hipMalloc((void **)&d_binder, binder_len ? binder_len : 1), "hipMalloc");
    fail(hipMalloc((void **)&d_out, 40ull * n), "hipMalloc");
    if (rc == WH_OK) {
        fail(hipMemcpyAsync(d_seeds, seeds, 16ull * n, hipMemcpyHostToDevice, st), "hipMemcpyAsync");
        if (binder_len) fail(hipMemcpyAsync(d_binder, binder, binder_len, hipMemcpyHostToDevice, st), "hipMemcpyAsync");
        fail(hipMemsetAsync(d_out, 0, 40ull * n, st), "hipMemsetAsync");
    }
    if (rc == WH_OK) {
        hipLaunchKernelGGL(k_test_f64_xof, dim3((n + 63) / 64), dim3(64), 0, st, d_seeds, usage, d_binder, binder_len, n_elems, bound, n, d_out);
        fail(hipGetLastError(), "k_test_f64_xof");
        fail(hipMemcpyAsync(out, d_out, 40ull * n, hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
        fail(hipStreamSynchronize(st), "hipStreamSynchronize");
    }
    if (d_seeds) (void)hipFree(d_seeds);
    if (d_binder) (void)hipFree(d_binder);
    if (d_out) (void)hipFree(d_out);
    return rc;
}
int32_t wh_prio3_count_set_test_sampler_bound(wh_ctx *c, uint64_t bound) {
    if (!c) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    state_of(c)->bound = bound ? bound : P64;
    return WH_OK;
}

}  // extern "C"
