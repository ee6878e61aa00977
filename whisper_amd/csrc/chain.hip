// All SHAKE128 work of a batched report in ONE launch, for circuits whose measurement share is short (Histogram, Average,
// the servers' SumVec of 128): Prio3::prepare_batched up to the query (batched-flpcp-rs/src/vdaf/prio3.rs:2480-2683,
// 2807-2862); the testing id (server-batch-prio3/src/main.rs:51) comes from the parse kernel.
//
// The long-share pipeline (k_expand beside k_meas, then k_hash, meas.hip / kernels.hip) is a chain of launches, each with as
// many lanes as there are reports: at 100 000 reports that is under two wavefronts per SIMD per launch, every wavefront
// issuing at the rate of a lone one, and the side-stream kernels finish when they finish.  Here a report is three or four
// independent chains of sponges, each chain one lane, all in one grid:
//   blocks [0, B)      Helper chain: measurement share (prio3.rs:2480-2496), both proof shares (:2498-2536) -- squeezed
//                      into the expansion rows, where the query kernel reads them -- then the Leader chain's stages from there
//   blocks [B, 2B)     Leader chain: own joint-rand part (:2543-2555), joint-rand seed and randomness (:2564-2585), own
//                      query-rand part over both proof shares (:2610-2642), query-rand seed and randomness (:2651-2672),
//                      r^m != 1 (flp.rs:413-419), barycentric weights and power tables of the query kernel
//   blocks [2B, 3B)    hash of the public parts and its field element (:2808-2840)
//   blocks [3B, 4B)    RLC coefficients (:2859-2862)
// Workgroups are dispatched in index order, so the longest chains start first and the short ones fill the wave slots that
// are left: one launch holds 4+ wavefronts per SIMD from 100 000 reports, and nothing waits for another stream.
// The Helper's shares go through the sampler proper (xof_emit: rejected chunks are skipped where they occur), so there is
// no dirty list and no second pass.  Every sponge of a lane runs through one loop body (one copy of the permutation code).
#include <hip/hip_runtime.h>
#include "tables_dev.hpp"

namespace wh {

enum : uint32_t {
    CS_EXP_MEAS = 0, CS_EXP_PROOF1, CS_EXP_PROOF2, CS_JR_PART, CS_JR_SEED, CS_JR, CS_QR_PART, CS_QR_SEED, CS_QR,
    CS_HASH, CS_HASH_ELEM, CS_RLC, CS_END
};

// PRIO: SpongeT's priority mode (keccak.hpp)
template <bool PREFETCH, int PRIO>
__device__ __forceinline__ void chain_body(const Prio3Dev &P, const Workspace &W, uint32_t npad, uint64_t vk0, uint64_t vk1,
                                           int32_t *status) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t seg = __builtin_amdgcn_readfirstlane(gid / npad);     // npad is a multiple of the block size
    const uint32_t t = gid - seg * npad;
#ifdef WH_CHAIN_SEG_MASK        // timing experiment (results invalid): only the segments of the mask run
    if (!((WH_CHAIN_SEG_MASK >> seg) & 1)) return;
#endif
    const uint32_t nl = W.counters[1], nh = W.counters[0];
    uint32_t i, first, last;
    if (seg == 0) {
        if (t >= nh) return;
        i = W.hlist[t]; first = CS_EXP_MEAS; last = CS_QR;
    } else if (seg == 1) {
        if (t >= nl) return;
        i = W.llist[t]; first = CS_JR_PART; last = CS_QR;
    } else {
        if (!report_of_slot(W, t, i)) return;
        first = seg == 2 ? CS_HASH : CS_RLC; last = seg == 2 ? CS_HASH_ELEM : CS_RLC;
    }
    if (status[i] != ST_OK) return;
    const RepDesc d = W.desc[i];
    RepScal *sc = &W.scal[i];
    const uint32_t psb = 16 * P.num_aggregators;    // bytes of a public share: one part per aggregator
    const uint8_t *agg_id = d.base + 16 + psb;
    const uint8_t *ps_parts = d.base + 16;
    const uint8_t *ps2_parts = d.tail;
    const uint8_t *pp = d.tail + psb + 8;
    const uint8_t *qr_blind = pp + 32ull * P.verifier_len;
    const uint8_t *nonce = qr_blind + 16;
    const uint32_t njr = 2 * P.jr_len;
    uint8_t *own = reinterpret_cast<uint8_t *>(&sc->own);
    SpongeT<true, PRIO> s;
    s.slot = PRIO < 0 ? wave_slot() : 0;
    F128 sd = f_zero();
    uint32_t sub = 0;                   // seed stages: first part of this pass (three parts a pass, wave-uniform)
#pragma unroll 1
    for (uint32_t stg = first; stg <= last; stg++) {
        bool last_sub = true;
        uint32_t usage = 0, l1 = 0, l2 = 0, l3 = 0, emit_n = 0;
        uint64_t k0 = 0, k1 = 0;
        const uint8_t *p1 = nullptr, *p2 = nullptr, *p3 = nullptr;
        uint8_t *emit_p = nullptr;
        switch (__builtin_amdgcn_readfirstlane(stg)) {
        case CS_EXP_MEAS:       // P::seed_stream(seed, dst(usage), [agg_id]).into_field_vec(n)
            usage = DST_MEAS; k0 = ld64(d.seeds); k1 = ld64(d.seeds + 8); p1 = agg_id; l1 = 1;
            emit_n = P.input_len; emit_p = const_cast<uint8_t *>(d.meas);
            break;
        case CS_EXP_PROOF1:     // batched report: meas_seed | proof1_seed | jr_blind | proof2_seed
            usage = DST_PROOF; k0 = ld64(d.seeds + 16); k1 = ld64(d.seeds + 24); p1 = agg_id; l1 = 1;
            emit_n = P.proof_len; emit_p = const_cast<uint8_t *>(d.proof1);
            break;
        case CS_EXP_PROOF2:
            usage = DST_PROOF; k0 = ld64(d.seeds + 48); k1 = ld64(d.seeds + 56); p1 = agg_id; l1 = 1;
            emit_n = P.proof_len; emit_p = const_cast<uint8_t *>(d.proof2);
            break;
        case CS_JR_PART:        // XOF(blind, 7) . [agg_id] . nonce . enc(meas share)
            usage = DST_JR_PART; k0 = ld64(d.jr_blind); k1 = ld64(d.jr_blind + 8);
            p1 = agg_id; l1 = 1; p2 = nonce; l2 = 16; p3 = d.meas; l3 = 16u * P.input_len;
            break;
        case CS_JR_SEED:        // derive_joint_rand_seed over the parts, own part at index role (prio3.rs:482-493)
        case CS_QR_SEED: {      // the same function, DST_JR_SEED included, for the query randomness (:2651-2665)
            const uint8_t *parts = stg == CS_JR_SEED ? ps_parts : ps2_parts;
            usage = DST_JR_SEED;
            last_sub = seed_parts(parts, own, d.role, P.num_aggregators, sub, p1, l1, p2, l2, p3, l3);
            break;
        }
        case CS_JR: usage = DST_JOINT_RAND; k0 = sd.lo; k1 = sd.hi; emit_n = njr; emit_p = reinterpret_cast<uint8_t *>(sc->jr); break;
        case CS_QR_PART:        // XOF(blind, 8) . enc(jr_fused) . enc(proof1) . enc(proof2)
            usage = DST_QR_PART; k0 = ld64(qr_blind); k1 = ld64(qr_blind + 8);
            p1 = reinterpret_cast<const uint8_t *>(sc->jr); l1 = 16 * njr;
            p2 = d.proof1; l2 = 16u * P.proof_len; p3 = d.proof2; l3 = 16u * P.proof_len;
            break;
        case CS_QR: usage = DST_QUERY_RAND; k0 = sd.lo; k1 = sd.hi; emit_n = 2; emit_p = reinterpret_cast<uint8_t *>(sc->qr); break;
        case CS_HASH:           // XOF(0, 10) . parts . parts2 . enc(pp0) . enc(pp1)
            usage = DST_HASH; p1 = ps_parts; l1 = psb; p2 = ps2_parts; l2 = psb; p3 = pp; l3 = 32u * P.verifier_len;
            break;
        case CS_HASH_ELEM: usage = DST_HASH; k0 = sd.lo; k1 = sd.hi; emit_n = 1; emit_p = reinterpret_cast<uint8_t *>(&sc->hash); break;
        default:                // CS_RLC: only the first verifier_len + 4 + (num_aggregators - 1) coefficients are consumed (:2865-2871)
            usage = DST_RLC; k0 = vk0; k1 = vk1; p1 = nonce; l1 = 16;
            emit_n = rlc_len(P); emit_p = reinterpret_cast<uint8_t *>(W.rlc + (uint64_t)i * rlc_len(P));
            break;
        }
        if (!sub) s.init_xof(P.alg_id, usage, k0, k1);
        s.template absorb3<PREFETCH>(p1, l1, p2, l2, p3, l3);
        if (!last_sub) { sub += 3; stg--; continue; }       // more than three aggregators: the same sponge, the next parts
        sub = 0;
        s.finalize();
        if (emit_n) {
            xof_emit(s, emit_p, emit_n, P.sampler_bound);
            if (stg == CS_RLC) fold_hash_coefficient(P, W.rlc + (uint64_t)i * rlc_len(P), d.role);
        } else {
            sd = F128{s.st[0], s.st[1]};          // Xof::into_seed
            // the locally computed parts vs the ones in the public shares (prio3.rs:2790-2805)
            if (stg == CS_JR_PART) { sc->rand_diff[0] = !f_eq(sd, ld128(ps_parts + 16 * d.role)); st128(own, sd); }
            if (stg == CS_QR_PART) { sc->rand_diff[1] = !f_eq(sd, ld128(ps2_parts + 16 * d.role)); st128(own, sd); }
        }
    }
    if (seg <= 1) {
        // flp.rs:413-419 for both runs: r^m == 1 is FlpError::Query
        const F128 r1 = sc->qr[0], r = sc->qr[1];
        const F128 r1m = f_sqr_n(r1, P.log2m), rm = f_sqr_n(r, P.log2m);
        if (f_eq(r1m, f_one()) || f_eq(rm, f_one())) { status[i] = ST_ERR_QUERY_RAND; return; }
        weights_and_tables(P, W, i, r, rm, sc->jr[P.jr_len]);
    }
}

#ifndef WH_CHAIN_WAVES
#define WH_CHAIN_WAVES 4
#endif
#ifndef WH_CHAIN_PRIO_MODE      // 0: no priorities outside the rotation runs; 1: rotating ranks; 2: fixed by chain length
#define WH_CHAIN_PRIO_MODE 0
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WH_CHAIN_WAVES, WH_CHAIN_WAVES)))
k_chain(Prio3Dev P, Workspace W, uint32_t npad, uint64_t vk0, uint64_t vk1, int32_t *status) {
#if WH_CHAIN_PRIO_MODE == 2     // fixed by chain length: Helper chains 2, Leader chains 1, the short ones 0
    const uint32_t seg = __builtin_amdgcn_readfirstlane(blockIdx.x * blockDim.x / npad);
    if (seg == 0) chain_body<false, 2>(P, W, npad, vk0, vk1, status);
    else if (seg == 1) chain_body<false, 1>(P, W, npad, vk0, vk1, status);
    else chain_body<false, 0>(P, W, npad, vk0, vk1, status);
#else
    chain_body<false, WH_CHAIN_PRIO_MODE == 1 ? -1 : 0>(P, W, npad, vk0, vk1, status);
#endif
}
// the same with the next rate block's loads in flight under the permutation (42 more VGPRs: three wavefronts a SIMD)
#ifndef WH_CHAIN_PF_WAVES
#define WH_CHAIN_PF_WAVES 3
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WH_CHAIN_PF_WAVES, WH_CHAIN_PF_WAVES)))
k_chain_prefetch(Prio3Dev P, Workspace W, uint32_t npad, uint64_t vk0, uint64_t vk1, int32_t *status) {
    chain_body<true, WH_CHAIN_PRIO_MODE == 1 ? -1 : 0>(P, W, npad, vk0, vk1, status);
}

void launch_chain(const Prio3Dev &P, const Workspace &W, uint32_t npad, bool prefetch, uint64_t vk0, uint64_t vk1, int32_t *status,
                  hipStream_t stream) {
    if (prefetch) hipLaunchKernelGGL(k_chain_prefetch, dim3(4 * npad / 256), dim3(256), 0, stream, P, W, npad, vk0, vk1, status);
    else hipLaunchKernelGGL(k_chain, dim3(4 * npad / 256), dim3(256), 0, stream, P, W, npad, vk0, vk1, status);
}

}  // namespace wh
