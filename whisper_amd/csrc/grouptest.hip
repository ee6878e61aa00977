// The group test's batch proofs (common/src/grouptest.rs:43-99: batch_proof, batch_proof_shake128, batch_proof_sha256) for
// MANY id ranges of one sorted tag list at a time -- the d proofs of the first split (grouptest.rs:182-208) and the two per
// mismatching range of every later round (grouptest.rs:306-318).
//
// One hash over a slice is one sequential sponge / Merkle-Damgard stream; nothing inside it runs in parallel, and a lone
// GPU lane walks it far slower than a host core.  What is parallel is the number of slices: one LANE per slice.  Lanes of a
// wavefront walk slices of different lengths, so nothing here is wave-uniform: whole blocks go into the state at static
// indices under the lane's own loop count, the tail block under per-word predicates.  The complete batch proof (one
// slice, grouptest.rs:143) stays a host job; it is accepted here, it is just slow.
#include <hip/hip_runtime.h>
#include <cstring>
#include <mutex>
#include <string>
#include "../../include/whisper_gfx950.h"
#include "keccak.hpp"
#include "sha256_dev.hpp"

namespace wh {
int ctx_device(wh_ctx *c);
hipStream_t ctx_stream(wh_ctx *c);
void **ctx_grouptest_slot(wh_ctx *c);
std::recursive_mutex &ctx_mutex(wh_ctx *c);
void ctx_set_error(wh_ctx *c, const std::string &what);

namespace {

// a slice of the sorted list and the bound that ends it: items [start, end), of which the proof covers those with
// lower < id <= upper when has_lower is set (the leading ids <= lower are skipped: they belong to the slice before),
// id <= upper otherwise
struct GtJob {
    uint64_t start, end;
    uint64_t upper_lo, upper_hi;
    uint64_t lower_lo, lower_hi;
    uint64_t has_lower, pad;
};
struct GtOut {
    uint8_t proof[32];
    uint64_t first, count;      // index of the first item hashed, number of items hashed (batch_proof's split index)
    uint64_t pad[2];
};

__device__ __forceinline__ bool id_gt(const uint8_t *ids, uint64_t i, uint64_t lo, uint64_t hi) {
    const uint64_t a = ld64g(ids + 16 * i), b = ld64g(ids + 16 * i + 8);
    return b > hi || (b == hi && a > lo);
}
// first index in [s, e) whose id is greater than (lo, hi); e if none (the list is sorted)
__device__ __forceinline__ uint64_t first_greater(const uint8_t *ids, uint64_t s, uint64_t e, uint64_t lo, uint64_t hi) {
    while (s < e) {
        const uint64_t m = s + ((e - s) >> 1);
        if (id_gt(ids, m, lo, hi)) e = m; else s = m + 1;
    }
    return s;
}

// SHAKE128(verify_key || tags)[0..16] (grouptest.rs:80-99).  The message is a run of 8-byte words: two of the key, then
// the tags (tag_bytes is a multiple of 8); a rate block is 21 of them.  `p + 8 j` is message word j for j >= 2.
__device__ __forceinline__ void proof_shake128(uint64_t vk0, uint64_t vk1, const uint8_t *p, uint64_t words, uint8_t *out) {
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    const uint64_t blocks = words / RATE_WORDS;
    const uint32_t rest = (uint32_t)(words - blocks * RATE_WORDS);
    uint64_t j0 = 0;
    if (blocks) {       // the block that starts with the key
        st[0] = vk0; st[1] = vk1;
#pragma unroll
        for (int i = 2; i < (int)RATE_WORDS; i++) st[i] = ld64g(p + 8 * i);
        keccak_f1600(st);
        j0 = RATE_WORDS;
    }
#pragma unroll 1
    for (uint64_t b = 1; b < blocks; b++, j0 += RATE_WORDS) {
#pragma unroll
        for (int i = 0; i < (int)RATE_WORDS; i++) st[i] ^= ld64g(p + 8 * (j0 + i));
        keccak_f1600(st);
    }
    // what is left of the message (fewer than 21 words), the padding byte behind it, the last bit of the rate
#pragma unroll
    for (int i = 0; i < (int)RATE_WORDS; i++) {
        uint64_t w = 0;
        if ((uint32_t)i < rest) {
            const uint64_t j = j0 + i;
            w = j == 0 ? vk0 : (j == 1 ? vk1 : ld64g(p + 8 * j));
        } else if ((uint32_t)i == rest) {
            w = 0x1f;
        }
        st[i] ^= w;
    }
    st[20] ^= 0x8000000000000000ULL;
    keccak_f1600(st);
    __builtin_memcpy(out, &st[0], 8);
    __builtin_memcpy(out + 8, &st[1], 8);
}

__device__ __forceinline__ uint32_t ld32g(const uint8_t *p) {
    typedef uint32_t u32_u1 __attribute__((aligned(1)));
    return *(const __attribute__((address_space(1))) u32_u1 *)p;
}
// SHA-256(verify_key || tags) (grouptest.rs:62-78).  Message words are 4 bytes, big endian; `p + 4 j` is word j for j >= 4.
__device__ __forceinline__ void proof_sha256(uint64_t vk0, uint64_t vk1, const uint8_t *p, uint64_t words, uint8_t *out) {
    uint32_t h[8], w[16];
    sha_init(h);
    const uint64_t bits = 32 * words;
    const uint64_t blocks = (words + 3 + 15) / 16;      // message, the 0x80 word, the 64-bit length
    const uint32_t vk[4] = {(uint32_t)vk0, (uint32_t)(vk0 >> 32), (uint32_t)vk1, (uint32_t)(vk1 >> 32)};
#pragma unroll 1
    for (uint64_t b = 0; b < blocks; b++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint64_t j = 16 * b + i;
            uint32_t v = 0;
            if (j < words) v = __builtin_bswap32(j < 4 ? vk[j & 3] : ld32g(p + 4 * j));
            else if (j == words) v = 0x80000000u;
            else if (j == 16 * blocks - 2) v = (uint32_t)(bits >> 32);
            else if (j == 16 * blocks - 1) v = (uint32_t)bits;
            w[i] = v;
        }
        sha_compress(h, w);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t v = __builtin_bswap32(h[i]);
        __builtin_memcpy(out + 4 * i, &v, 4);
    }
}

__global__ void __launch_bounds__(64) k_batch_proofs(const uint8_t *ids, const uint8_t *tags, uint32_t tag_bytes, const GtJob *jobs,
                                                     uint32_t k, uint64_t vk0, uint64_t vk1, uint32_t proof_size, GtOut *out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k) return;
    const GtJob jb = jobs[t];
    uint64_t s = jb.start;
    if (jb.has_lower) s = first_greater(ids, s, jb.end, jb.lower_lo, jb.lower_hi);
    const uint64_t e = first_greater(ids, s, jb.end, jb.upper_lo, jb.upper_hi);
    const uint64_t cnt = e - s;
    GtOut *o = out + t;
    o->first = s;
    o->count = cnt;
    const uint8_t *p = tags + (uint64_t)tag_bytes * s - 16;      // message byte m >= 16 is p[m]
    if (proof_size == 16) {
        proof_shake128(vk0, vk1, p, 2 + cnt * (tag_bytes >> 3), o->proof);
        for (int i = 16; i < 32; i++) o->proof[i] = 0;
    } else {
        proof_sha256(vk0, vk1, p, 4 + cnt * (tag_bytes >> 2), o->proof);
    }
}

constexpr uint32_t GT_MAX_SLICES = 1u << 24;     // per call; the staging is 160 bytes per slice
struct GtState {
    GtJob *d_jobs = nullptr, *h_jobs = nullptr;
    GtOut *d_out = nullptr, *h_out = nullptr;
    uint32_t cap = 0;
};

#define G_HIPCHK(c, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) { ctx_set_error(c, std::string(#call) + ": " + hipGetErrorString(e_)); return WH_ERR_HIP; } \
    } while (0)

int32_t gt_state(wh_ctx *c, uint32_t k, GtState *&S) {
    void **slot = ctx_grouptest_slot(c);
    if (!*slot) *slot = new GtState();
    S = static_cast<GtState *>(*slot);
    if (k <= S->cap) return WH_OK;
    if (S->d_jobs) G_HIPCHK(c, hipFree(S->d_jobs));
    if (S->d_out) G_HIPCHK(c, hipFree(S->d_out));
    if (S->h_jobs) G_HIPCHK(c, hipHostFree(S->h_jobs));
    if (S->h_out) G_HIPCHK(c, hipHostFree(S->h_out));
    S->d_jobs = S->h_jobs = nullptr; S->d_out = S->h_out = nullptr; S->cap = 0;
    const uint32_t cap = k < 1024 ? 1024 : k + k / 2;
    G_HIPCHK(c, hipMalloc((void **)&S->d_jobs, cap * sizeof(GtJob)));
    G_HIPCHK(c, hipMalloc((void **)&S->d_out, cap * sizeof(GtOut)));
    G_HIPCHK(c, hipHostMalloc((void **)&S->h_jobs, cap * sizeof(GtJob), hipHostMallocDefault));
    G_HIPCHK(c, hipHostMalloc((void **)&S->h_out, cap * sizeof(GtOut), hipHostMallocDefault));
    S->cap = cap;
    return WH_OK;
}

// the jobs are in S->h_jobs: copy, hash, copy back, hand out
int32_t gt_run(wh_ctx *c, GtState *S, const uint8_t *d_ids, const uint8_t *d_tags, uint32_t tag_bytes, const uint8_t *verify_key,
               uint32_t k, uint32_t proof_size, uint8_t *proofs, uint64_t *firsts, uint64_t *counts) {
    hipStream_t st = ctx_stream(c);
    uint64_t vk0, vk1;
    memcpy(&vk0, verify_key, 8);
    memcpy(&vk1, verify_key + 8, 8);
    G_HIPCHK(c, hipMemcpyAsync(S->d_jobs, S->h_jobs, k * sizeof(GtJob), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_batch_proofs, dim3((k + 63) / 64), dim3(64), 0, st, d_ids, d_tags, tag_bytes, S->d_jobs, k, vk0, vk1, proof_size,
                       S->d_out);
    G_HIPCHK(c, hipGetLastError());
    G_HIPCHK(c, hipMemcpyAsync(S->h_out, S->d_out, k * sizeof(GtOut), hipMemcpyDeviceToHost, st));
    G_HIPCHK(c, hipStreamSynchronize(st));
    for (uint32_t i = 0; i < k; i++) {
        memcpy(proofs + (size_t)proof_size * i, S->h_out[i].proof, proof_size);
        if (firsts) firsts[i] = S->h_out[i].first;
        counts[i] = S->h_out[i].count;
    }
    return WH_OK;
}

bool gt_params_ok(wh_ctx *c, const void *d_ids, const void *d_tags, uint64_t n, uint32_t tag_bytes, const void *vk, uint32_t proof_size) {
    if (!c || !vk || (n && (!d_ids || !d_tags))) return false;
    if (tag_bytes == 0 || tag_bytes % 8 || tag_bytes > 4096) return false;
    return proof_size == 16 || proof_size == 32;
}

}  // namespace

void grouptest_free(void *state) {
    GtState *S = static_cast<GtState *>(state);
    if (!S) return;
    if (S->d_jobs) (void)hipFree(S->d_jobs);
    if (S->d_out) (void)hipFree(S->d_out);
    if (S->h_jobs) (void)hipHostFree(S->h_jobs);
    if (S->h_out) (void)hipHostFree(S->h_out);
    delete S;
}

}  // namespace wh

using namespace wh;

extern "C" {

int32_t wh_grouptest_batch_proofs_dev(wh_ctx *c, const uint8_t *d_ids, const uint8_t *d_tags, uint64_t n, uint32_t tag_bytes,
                                      const uint8_t *verify_key, const uint64_t *starts, const uint64_t *ends, const uint8_t *uppers,
                                      uint32_t k, uint32_t proof_size, uint8_t *proofs, uint64_t *counts) {
    if (!gt_params_ok(c, d_ids, d_tags, n, tag_bytes, verify_key, proof_size)) return WH_ERR_PARAM;
    if (k == 0) return WH_OK;
    if (k > GT_MAX_SLICES || !starts || !ends || !uppers || !proofs || !counts) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    for (uint32_t i = 0; i < k; i++)
        if (starts[i] > ends[i] || ends[i] > n) { ctx_set_error(c, "wh_grouptest_batch_proofs_dev: slice outside the list"); return WH_ERR_PARAM; }
    G_HIPCHK(c, hipSetDevice(ctx_device(c)));
    GtState *S;
    if (int32_t rc = gt_state(c, k, S)) return rc;
    for (uint32_t i = 0; i < k; i++) {
        GtJob &j = S->h_jobs[i];
        memset(&j, 0, sizeof j);
        j.start = starts[i]; j.end = ends[i];
        memcpy(&j.upper_lo, uppers + 16 * (size_t)i, 8);
        memcpy(&j.upper_hi, uppers + 16 * (size_t)i + 8, 8);
    }
    return gt_run(c, S, d_ids, d_tags, tag_bytes, verify_key, k, proof_size, proofs, nullptr, counts);
}

int32_t wh_grouptest_split_proofs_dev(wh_ctx *c, const uint8_t *d_ids, const uint8_t *d_tags, uint64_t n, uint32_t tag_bytes,
                                      const uint8_t *verify_key, const uint8_t *uppers, uint32_t d, uint32_t proof_size, uint8_t *proofs,
                                      uint64_t *firsts, uint64_t *counts) {
    if (!gt_params_ok(c, d_ids, d_tags, n, tag_bytes, verify_key, proof_size)) return WH_ERR_PARAM;
    if (d == 0) return WH_OK;
    if (d > GT_MAX_SLICES || !uppers || !proofs || !firsts || !counts) return WH_ERR_PARAM;
    std::lock_guard<std::recursive_mutex> lk(ctx_mutex(c));
    // slice i starts where slice i - 1 stopped: behind the ids <= the largest bound so far
    unsigned __int128 prev = 0, run_max = 0;
    G_HIPCHK(c, hipSetDevice(ctx_device(c)));
    GtState *S;
    if (int32_t rc = gt_state(c, d, S)) return rc;
    for (uint32_t i = 0; i < d; i++) {
        GtJob &j = S->h_jobs[i];
        memset(&j, 0, sizeof j);
        j.start = 0; j.end = n;
        unsigned __int128 up;
        memcpy(&up, uppers + 16 * (size_t)i, 16);
        if (i) {
            if (prev > run_max) run_max = prev;
            j.has_lower = 1;
            j.lower_lo = (uint64_t)run_max; j.lower_hi = (uint64_t)(run_max >> 64);
        }
        j.upper_lo = (uint64_t)up; j.upper_hi = (uint64_t)(up >> 64);
        prev = up;
    }
    return gt_run(c, S, d_ids, d_tags, tag_bytes, verify_key, d, proof_size, proofs, firsts, counts);
}

}  // extern "C"
