// The measurement-share sponges, in their own translation unit (hipcc 7.2's inliner does not
// survive all sponge kernels in one).
//
// Joint-randomness part of every report (prio3.rs:2543-2555): XOF(blind, 7) . [agg_id] . nonce .
// enc(meas_share) -> seed, 1561 Keccak-f for the 16384-element share of the bench workload.  For a
// Helper the share itself is the output of another sponge, XOF(meas_seed, 1) . [agg_id]
// (prio3.rs:2480-2496), another 1561 Keccak-f.  Both are sequential chains, so their
// parallelism is the number of reports, and a batch has too few to fill the chip when they run
// one after the other (k_expand, then the absorb).  Here a Helper gets TWO lanes: the even lane
// squeezes the share (and stores it for the query kernel), the odd lane absorbs the same words
// one permutation behind, handed over through DPP; Leader-role reports absorb their share from
// the report bytes in the same launch.  That is three long lanes per report pair instead of two
// after one, i.e. ~3 waves per SIMD in one launch.
//
// The hand-over assumes that no squeezed 16-byte chunk is rejected by the sampler (probability
// 2^-59 each): then the absorbed stream is the squeezed stream, 42 bytes (the absorber's prefix)
// out of phase, and every word index is static.  A pair that sees a rejected chunk puts its report
// on the dirty list; those reports are redone by the general path (k_expand + mode 1 here).
//
// The two lanes of a pair are two THREADS to the compiler.  The DPP exchange (a convergent operation) is what keeps their
// regions in program order; a variant that handed the words over through memory -- `if (even) { stores } fence;
// if (!even) { loads }` -- was compiled to `if (!even) { fence; loads } else { stores; fence }`, legal per thread, and read the
// row before it was written until a __builtin_amdgcn_wave_barrier() sat between the regions (profiles/r3_meas_readback.txt).
#include <hip/hip_runtime.h>
#include "keccak.hpp"
#include "prio3_dev.hpp"

namespace wh {

__device__ __forceinline__ uint32_t dpp_prev(uint32_t x) {      // x of lane - 1 (rows of 16 lanes)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   // the first lane of a row reads 0: it is a squeezer
}
struct W2 { uint32_t lo, hi; };
__device__ __forceinline__ W2 fetch_prev(uint64_t w) { return W2{dpp_prev((uint32_t)w), dpp_prev((uint32_t)(w >> 32))}; }
// b ^= ((cur << 16) | (prev >> 48)) & mask: the squeezed stream enters the absorber's block at byte 42 = 5 words + 2 bytes
__device__ __forceinline__ void xor_shifted(uint64_t &b, const W2 &cur, uint32_t prev_hi, uint32_t m) {
    const uint32_t vlo = __builtin_amdgcn_alignbit(cur.lo, prev_hi, 16);
    const uint32_t vhi = __builtin_amdgcn_alignbit(cur.hi, cur.lo, 16);
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)b, vlo, m, 0x78);           // a ^ (b & c)
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(b >> 32), vhi, m, 0x78);
    b = lo | ((uint64_t)hi << 32);
}

enum : uint32_t { LANE_FULL = 0, LANE_SQUEEZE = 1, LANE_ABSORB = 2 };
#ifdef WH_MEAS_NT_STORES
#define WH_MEAS_ST st128g_nt
#else
#define WH_MEAS_ST st128g
#endif

// (wave_slot / rotate_prio: keccak.hpp)
// one rate block as ten 16-byte loads and one 8-byte load (any alignment)
__device__ __forceinline__ void load_block(uint64_t (&w)[RATE_WORDS], const uint8_t *p) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
#ifdef WH_MEAS_NT_LOADS
        const F128 t = ld128g_nt(p + 16 * i);
#else
        const F128 t = ld128g(p + 16 * i);
#endif
        w[2 * i] = t.lo; w[2 * i + 1] = t.hi;
    }
    w[20] = ld64g(p + 160);
}
// nb whole rate blocks from p into a sponge that sits on a block boundary (pos == 0); the next block's loads are in
// flight under the permutation
__device__ __forceinline__ void absorb_whole_blocks(SpongeG &s, const uint8_t *p, uint32_t nb, uint32_t slot) {
    if (!nb) return;
    uint64_t w[RATE_WORDS];
    load_block(w, p);
#pragma unroll 1
    for (uint32_t b = 0; b < nb; b++) {
#pragma unroll
        for (int i = 0; i < (int)RATE_WORDS; i++) s.st[i] ^= w[i];
        p += SHAKE128_RATE;
        if (b + 1 < nb) load_block(w, p);
        keccak_f1600_ranked(s.st, rotate_prio(slot));
    }
}

// Diagnostic build only (tools/meas_bench.hip): lane 0 of every wavefront stamps the shader clock and the 100 MHz
// real-time counter around its sponge loop into a buffer nothing else reads; the product build has no stamp.
#ifdef WH_MEAS_CLOCKPROBE
#define WH_STAMP(slot)                                                                            \
    do {                                                                                          \
        if ((threadIdx.x & 63) == 0) {                                                            \
            g_probe[6ull * (gid >> 6) + 2 * (slot)] = __builtin_amdgcn_s_memtime();               \
            g_probe[6ull * (gid >> 6) + 2 * (slot) + 1] = __builtin_amdgcn_s_memrealtime();       \
            g_probe[6ull * (gid >> 6) + 4] = __builtin_amdgcn_s_getreg(63492);  /* HW_ID */       \
            g_probe[6ull * (gid >> 6) + 5] = __builtin_amdgcn_s_getreg(63508);  /* XCC_ID */      \
        }                                                                                         \
    } while (0)
#else
#define WH_STAMP(slot) do { } while (0)
#endif

// mode 0: Leader-role reports (whole absorb) and Helper pairs; mode 1: the dirty list (whole absorb from the
// rows k_expand wrote).
template <uint32_t mode>
__device__ __forceinline__ void meas_body(const Prio3Dev &P, const Workspace &W, int32_t *status) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t slot = wave_slot();
    uint32_t i, kind;
    if (mode == 1) {
        if (gid >= W.counters[2]) return;
        i = W.dirty[gid];
        kind = LANE_FULL;
    } else {
        const uint32_t nl = W.counters[1], nh = W.counters[0], nlpad = (nl + 63) & ~63u;
        if (gid < nlpad) {
            if (gid >= nl) return;
            i = W.llist[gid];
            kind = LANE_FULL;
        } else {
            const uint32_t t = gid - nlpad;
            if ((t >> 1) >= nh) return;
            i = W.hlist[t >> 1];
            kind = (t & 1) ? LANE_ABSORB : LANE_SQUEEZE;
        }
    }
    if (status[i] != ST_OK) return;
    const RepDesc d = W.desc[i];
    const uint8_t *agg_id = P.mode ? d.base : d.base + 16 + 16 * P.num_aggregators;
    const uint8_t *nonce = P.mode ? d.tail : d.tail + 16 * P.num_aggregators + 8 + 32ull * P.verifier_len + 16;
    const F128 bound = P.sampler_bound;

    SpongeG s;
    const uint8_t *key = kind == LANE_SQUEEZE ? d.seeds : d.jr_blind;
    s.init_xof(P.alg_id, kind == LANE_SQUEEZE ? DST_MEAS : DST_JR_PART, ld64(key), ld64(key + 8));
    // the short prefixes never fill a block: [agg_id] for the share stream, [agg_id] . nonce for the part
    {
        const uint32_t plen = kind == LANE_SQUEEZE ? 1 : 17;
        for (uint32_t b = 0; b < plen; b++) {
            const uint8_t v = b ? nonce[b - 1] : agg_id[0];
            s.xor_word(s.pos >> 3, (uint64_t)v << (8 * (s.pos & 7)));     // active lanes share pos
            s.pos++;
        }
    }
    WH_STAMP(0);
    const uint8_t *rest = d.meas;                    // what phase 2 still has to absorb
    uint32_t rest_len = 16u * P.input_len;
    uint32_t dirty = 0;

    if (kind != LANE_FULL) {
        const bool even = kind == LANE_SQUEEZE;
        const uint32_t m = even ? 0u : ~0u;
        if (even) s.finalize();                      // first output block
        uint8_t *out = const_cast<uint8_t *>(d.meas);
        const uint32_t iters = P.input_len / 21;     // 21 chunks = two rate blocks
        // Words 15..20 of the squeezed block reach the absorber only after ITS permutation: they wait in LDS (48 B per
        // lane, written and read by the same wavefront, so no barrier), not in registers across the permutation.
        __shared__ uint64_t park[256 * 6];
        uint64_t *mine = park + threadIdx.x, *prev = park + (threadIdx.x ? threadIdx.x - 1 : 0);
        // Rejection check (prng.rs:73-99), cheap form: the running maximum of the chunks' top 32 bits; only a half-iteration
        // that saw a top word >= the bound's (probability 2^-32 per chunk with the real p) looks at the chunks exactly.
        const uint32_t bound_top = (uint32_t)(bound.hi >> 32);
#define WH_EXACT(lo_, hi_) dirty |= !((hi_) < bound.hi || ((hi_) == bound.hi && (lo_) < bound.lo))
#pragma unroll 1
        for (uint32_t it = 0; it < iters; it++) {
            uint32_t prev_hi = 0, top = 0;
            const uint32_t rank = rotate_prio(slot);
            // even block e[0..20] is in the squeezer's state: absorber words 5..20 take e[0..15]
#pragma unroll
            for (int w = 0; w < 16; w++) {
                const W2 c = fetch_prev(s.st[w]);
                xor_shifted(s.st[5 + w], c, prev_hi, m);
                prev_hi = c.hi;
            }
#pragma unroll
            for (int w = 0; w < 6; w++) mine[256 * w] = s.st[15 + w];
#pragma unroll
            for (int c = 0; c < 10; c++) top = max(top, (uint32_t)(s.st[2 * c + 1] >> 32));
            if (__builtin_expect(__any(top >= bound_top), 0)) {
#pragma unroll
                for (int c = 0; c < 10; c++) WH_EXACT(s.st[2 * c], s.st[2 * c + 1]);
            }
            if (even) {
#pragma unroll
                for (int c = 0; c < 10; c++) WH_MEAS_ST(out + 16 * c, F128{s.st[2 * c], s.st[2 * c + 1]});
            }
            const uint64_t carry = s.st[20];
            keccak_f1600_ranked(s.st, rank);
            // absorber: words 0..4 of its next block from e[15..20]; then o[0..15] into words 5..20
            {
                uint64_t t[6];
#pragma unroll
                for (int w = 0; w < 6; w++) t[w] = prev[256 * w];
#pragma unroll
                for (int w = 0; w < 5; w++) xor_shifted(s.st[w], W2{(uint32_t)t[w + 1], (uint32_t)(t[w + 1] >> 32)}, (uint32_t)(t[w] >> 32), m);
                prev_hi = (uint32_t)(t[5] >> 32);
            }
#pragma unroll
            for (int w = 0; w < 16; w++) {
                const W2 c = fetch_prev(s.st[w]);
                xor_shifted(s.st[5 + w], c, prev_hi, m);
                prev_hi = c.hi;
            }
#pragma unroll
            for (int w = 0; w < 6; w++) mine[256 * w] = s.st[15 + w];
            top = (uint32_t)(s.st[0] >> 32);
#pragma unroll
            for (int c = 0; c < 10; c++) top = max(top, (uint32_t)(s.st[2 * c + 2] >> 32));
            if (__builtin_expect(__any(top >= bound_top), 0)) {
                WH_EXACT(carry, s.st[0]);
#pragma unroll
                for (int c = 0; c < 10; c++) WH_EXACT(s.st[2 * c + 1], s.st[2 * c + 2]);
            }
            if (even) {
                WH_MEAS_ST(out + 160, F128{carry, s.st[0]});
#pragma unroll
                for (int c = 0; c < 10; c++) WH_MEAS_ST(out + 176 + 16 * c, F128{s.st[2 * c + 1], s.st[2 * c + 2]});
            }
            out += 336;
            keccak_f1600_ranked(s.st, rank);
            {
                uint64_t t[6];
#pragma unroll
                for (int w = 0; w < 6; w++) t[w] = prev[256 * w];
#pragma unroll
                for (int w = 0; w < 5; w++) xor_shifted(s.st[w], W2{(uint32_t)t[w + 1], (uint32_t)(t[w + 1] >> 32)}, (uint32_t)(t[w] >> 32), m);
                xor_shifted(s.st[5], W2{0, 0}, (uint32_t)(t[5] >> 32), m);
            }
        }
        dirty &= even;
        // the last r < 21 chunks: squeezed word by word, absorbed from memory in phase 2
        const uint32_t r = P.input_len - 21 * iters;
        if (even) {
            uint32_t wpos = 0;
            for (uint32_t c = 0; c < r; c++) {
                uint64_t v[2];
                for (int h = 0; h < 2; h++) {
                    if (wpos == RATE_WORDS) { keccak_f1600(s.st); wpos = 0; }
                    v[h] = s.get_word(wpos++);
                }
                WH_EXACT(v[0], v[1]);
                st128g(out, F128{v[0], v[1]});
                out += 16;
            }
        }
#undef WH_EXACT
        // a rejected chunk anywhere makes the whole hand-over invalid: the absorber reports it
        dirty = __builtin_amdgcn_update_dpp(0, (int)dirty, 0x111, 0xf, 0xf, false) | dirty;
        WH_STAMP(1);
        if (even) return;
        if (dirty) {
            W.dirty[atomicAdd(&W.counters[2], 1u)] = i;
            return;
        }
        // the squeezer's last stores must be visible to this lane's loads
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        rest = d.meas + 16ull * 21 * iters;
        rest_len = 16u * r;
    }
    // phase 2: Leader (and mode 1): the whole share; absorber lanes: the tail
    {
        // up to the next block boundary the generic way, whole blocks with the explicit loop, the tail the generic way
        const uint32_t head = min(rest_len, s.pos ? SHAKE128_RATE - s.pos : 0u);     // pos is wave-uniform
        s.absorb(rest, head);
        const uint32_t nb = (rest_len - head) / SHAKE128_RATE;
        absorb_whole_blocks(s, rest + head, nb, slot);
        s.absorb(rest + head + nb * SHAKE128_RATE, rest_len - head - nb * SHAKE128_RATE);
    }
    s.finalize();
    if (kind == LANE_FULL) WH_STAMP(1);
    const F128 sd = F128{s.st[0], s.st[1]};
    // the locally computed part vs the one in the public share (prio3.rs:2790-2805); the
    // descriptor is read again here rather than held in registers across the long loops
    asm volatile("" ::: "memory");
    const RepDesc d2 = W.desc[i];
    RepScal *sc = &W.scal[i];
    const uint8_t *their_part = (P.mode ? d2.base + 1 : d2.base + 16) + 16 * d2.role;
    sc->rand_diff[0] = !f_eq(sd, ld128(their_part));
    sc->own = sd;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_meas(Prio3Dev P, Workspace W, int32_t *status) {
    meas_body<0>(P, W, status);
}
// the general path for the dirty list (empty unless a share stream had a rejected chunk)
__global__ void __launch_bounds__(256) k_meas_redo(Prio3Dev P, Workspace W, int32_t *status) { meas_body<1>(P, W, status); }

void launch_meas(const Prio3Dev &P, const Workspace &W, uint32_t mode, uint32_t npad, int32_t *status, hipStream_t stream) {
    // mode 0: Leaders (padded to a wavefront) + two lanes per Helper; mode 1: at most npad dirty reports
    if (mode) hipLaunchKernelGGL(k_meas_redo, dim3(npad / 256), dim3(256), 0, stream, P, W, status);
    else hipLaunchKernelGGL(k_meas, dim3((3 * npad + 64 + 255) / 256), dim3(256), 0, stream, P, W, status);
}

}  // namespace wh
