// gfx950 kernels for whisper's server hot path: batched FLPCP verification of Prio3 reports
// (reference: Prio3::prepare_batched, batched-flpcp-rs/src/vdaf/prio3.rs:2466-2879, called from
// server-batch-prio3/src/main.rs:33-60) and share aggregation (vdaf.rs:658-676,
// hhcore/src/bucket.rs:48-55).
//
// Pipeline over one sub-batch of encoded reports resident in HBM:
//   parse   (lane / report)   header checks, pointers, per-role report lists
//   expand  (lane / Helper)   SHAKE128 expansion of the Helper's seeds into share rows
//   hash    (lane / report)   joint-rand + query-rand Fiat-Shamir sponges, barycentric weights;
//                             public-parts hash, RLC stream, AES testing id
//   query   (wave / report)   range checks, truncate, wire evaluation at r, decide, RLC tag
// then, once the host has the group-test verdict:
//   aggregate                 masked column sums of the out shares (HBM streaming)
//
// The SHAKE work is one sponge per lane with the state in VGPRs (keccak.hpp); the field work
// is one wavefront per report with lanes across the gadget chunk (f128.hpp).  No MFMA: this is
// prime-field integer arithmetic.
#include <hip/hip_runtime.h>
#include "keccak.hpp"
#include "prio3_dev.hpp"
#include "tables_dev.hpp"
#include "f64.hpp"

namespace wh {

// ------------------------------------------------------------------------------- helpers ----
__device__ __forceinline__ uint64_t be64(const uint8_t *p) { return __builtin_bswap64(ld64(p)); }

// ld128g (keccak.hpp): 16-byte load from a pointer known to be in global memory -- global_load instead of flat_load, so that
// loads issued ahead of their use are counted by vmcnt alone.

__device__ __forceinline__ F128 shfl_xor_f(const F128 &v, int mask) {
    F128 r;
    r.lo = __shfl_xor((unsigned long long)v.lo, mask, 64);
    r.hi = __shfl_xor((unsigned long long)v.hi, mask, 64);
    return r;
}
// sum over the 64 lanes of a wavefront, result in every lane
__device__ __forceinline__ F128 wave_sum(F128 v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = f_add(v, shfl_xor_f(v, m));
    return v;
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v |= __shfl_xor(v, m, 64);
    return v;
}
// --------------------------------------------------------------------------------- parse ----
// VdafBatchedKey::decode_with_param (vdaf.rs:164-192) as far as structure goes: lengths,
// agg_id, num_queries.  Field-element range checks happen where the elements are loaded
// (query kernel).  Trailing bytes are not looked at (server-batch-prio3/src/main.rs:46-47).
// The testing id of a well-formed report is computed here too (AES-128 of the client id, main.rs:51): the ids of a batch exist
// before its first sponge runs, so the sort by testing id can run beside the sponge kernels (wh_prio3_prepare_sorted_batch_dev).
__global__ void k_parse(Prio3Dev P, Workspace W, const uint8_t *blob, const uint64_t *offsets, uint32_t n,
                        int32_t *status, uint8_t *tags, uint8_t *testing_ids, uint8_t *out_shares, const uint8_t *aes_rk) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *rep = blob + offsets[i];
    // a descending offset pair is an empty report (ST_ERR_SHORT), not a length of ~2^64
    uint64_t rep_len = offsets[i + 1] >= offsets[i] ? offsets[i + 1] - offsets[i] : 0;
    RepDesc d;
    d.base = rep; d.meas = d.proof1 = d.proof2 = d.tail = d.jr_blind = d.seeds = nullptr; d.role = 0; d.hslot = 0;
    int32_t st = ST_OK;
    // client id | num_aggregators joint-rand parts | agg id | share | ... (vdaf.rs:148-162); psb = bytes of a public share
    const uint64_t psb = 16ull * P.num_aggregators, head = 17 + psb;
    if (rep_len < head) st = ST_ERR_SHORT;
    if (!st) {
        d.role = rep[16 + psb];
        if (d.role >= P.num_aggregators) st = ST_ERR_AGG_ID;
    }
    if (!st) {
        uint64_t share_len = d.role == 0 ? 16ull * (P.input_len + 2ull * P.proof_len) + 16 : 64;
        uint64_t off = head + share_len;
        if (rep_len < off + psb + 8) st = ST_ERR_SHORT;
        if (!st) {
            d.tail = rep + off;
            uint64_t nq = be64(rep + off + psb);
            uint64_t rest = rep_len - (off + psb + 8);
            if (nq != P.verifier_len) {
                st = (nq > rest / 32 || rest - 32 * nq < 32) ? ST_ERR_SHORT : ST_ERR_NUM_QUERIES;
            } else if (rest < 32ull * P.verifier_len + 32) {
                st = ST_ERR_SHORT;
            }
        }
    }
    if (!st) {
        if (d.role == 0) {
            d.meas = rep + head;
            d.proof1 = d.meas + 16ull * P.input_len;
            d.jr_blind = d.proof1 + 16ull * P.proof_len;
            d.proof2 = d.jr_blind + 16;
        } else {
            d.hslot = atomicAdd(&W.counters[0], 1u);
            W.hlist[d.hslot] = i;
            uint8_t *row = W.hrows + (uint64_t)d.hslot * W.hrow_bytes;
            d.meas = row;
            d.proof1 = row + 16ull * P.input_len;
            d.proof2 = d.proof1 + 16ull * P.proof_len;
            d.jr_blind = rep + head + 32;
            d.seeds = rep + head;
        }
    }
    // the later kernels run per role over the two lists; a malformed report is in neither and
    // gets its (zero) outputs here
    if (!st && d.role == 0) W.llist[atomicAdd(&W.counters[1], 1u)] = i;
    if (st) {
        F128 *out_row = reinterpret_cast<F128 *>(out_shares) + (uint64_t)i * P.output_len;
        for (uint32_t e = 0; e < P.output_len; e++) out_row[e] = f_zero();
        st128(tags + 16ull * i, f_zero());
        st128(testing_ids + 16ull * i, f_zero());      // an undecodable report has no id: zero, whatever the buffer held
    } else {
        aes_testing_id(aes_rk, rep, testing_ids + 16ull * i);
    }
    W.desc[i] = d;
    status[i] = st;
}

// (the Helper expansion kernel lives in expand.hip: hipcc 7.2 crashes in its inliner when this
// translation unit holds both sponge kernels)

// ---------------------------------------------------------------------------------- hash ----
// Three wavefronts a SIMD (168 VGPRs).  Register prefetch of the next rate block: it lost while jobs A and B shared one launch
// (3 906 wavefronts: the third wavefront per SIMD was worth more than the loads in flight, 4.45 -> 3.65 ms at 125 000 reports);
// since job B runs beside k_meas, job A alone is under two wavefronts per SIMD, every block's loads are exposed, and the
// prefetch wins (3.17 -> 2.93 ms).
#ifndef WH_HASH_PREFETCH
#define WH_HASH_PREFETCH true
#endif
#ifndef WH_HASH_ATTR
#ifndef WH_HASH_WAVES
#define WH_HASH_WAVES 3
#endif
#define WH_HASH_ATTR __attribute__((amdgpu_waves_per_eu(WH_HASH_WAVES, WH_HASH_WAVES)))
#endif
// Job A (lanes [0, npad)): five sponges (the own joint-rand part, prio3.rs:2543-2555, comes from
//   k_meas) -- joint-rand
//   seed (derive_joint_rand_seed :482-493 with the own part substituted, :2564-2578), joint
//   randomness (:2580-2585), own query-rand part (:2610-2642), query-rand seed (:2651-2665, which
//   reuses DST_JR_SEED), query randomness (:2667-2672) -- then the barycentric weights.
// Job B (lanes [npad, 2 npad)): hash of the public parts (:2808-2840), RLC coefficients
//   (:2859-2862).  (The testing id, main.rs:51, comes from the parse kernel.)
// All sponges of a lane go through one loop body; a wave-uniform switch picks the inputs.
__global__ void __launch_bounds__(256) WH_HASH_ATTR k_hash(Prio3Dev P, Workspace W, uint32_t npad, uint32_t jobs, uint64_t vk0,
                                              uint64_t vk1, int32_t *status) {
    uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    // jobs: 0 = A only, 1 = B only, 2 = A in lanes [0, npad) and B in [npad, 2 npad)
    uint32_t which = jobs == 2 ? job >= npad : jobs;
    uint32_t slot = (jobs == 2 && which) ? job - npad : job;
    uint32_t i;
    if (!report_of_slot(W, slot, i) || status[i] != ST_OK) return;
    const RepDesc d = W.desc[i];
    RepScal *sc = &W.scal[i];
    const uint32_t psb = 16 * P.num_aggregators;    // bytes of a public share: one part per aggregator
    const uint8_t *ps_parts = d.base + 16;
    const uint8_t *ps2_parts = d.tail;
    const uint8_t *pp = d.tail + psb + 8;
    const uint8_t *qr_blind = pp + 32ull * P.verifier_len;
    const uint8_t *nonce = qr_blind + 16;
    const uint32_t njr = 2 * P.jr_len;
    uint8_t *own = reinterpret_cast<uint8_t *>(&sc->own);
    SpongeG s;
    F128 sd = f_zero();
    const uint32_t nst = which ? 3 : 6;
    uint32_t sub = 0;                   // seed stages: first part of this pass (three parts a pass, wave-uniform)
#pragma unroll 1
    for (uint32_t stg = which ? 0 : 1; stg < nst; stg++) {     // job A's stage 0 (the own joint-rand part) is k_meas
        bool last_sub = true;
        uint32_t usage = 0, l1 = 0, l2 = 0, l3 = 0, emit_n = 0;
        uint64_t k0 = 0, k1 = 0;
        const uint8_t *p1 = nullptr, *p2 = nullptr, *p3 = nullptr;
        uint8_t *emit_p = nullptr;
        switch (__builtin_amdgcn_readfirstlane(which * 8 + stg)) {
        case 1:  // joint-rand seed over the parts, own part at index role
        case 4: {
            const uint8_t *parts = stg == 1 ? ps_parts : ps2_parts;
            usage = DST_JR_SEED;
            last_sub = seed_parts(parts, own, d.role, P.num_aggregators, sub, p1, l1, p2, l2, p3, l3);
            break;
        }
        case 2: usage = DST_JOINT_RAND; k0 = sd.lo; k1 = sd.hi; emit_n = njr; emit_p = reinterpret_cast<uint8_t *>(sc->jr); break;
        case 3:  // own query-rand part: XOF(blind, 8) . enc(jr_fused) . enc(proof1) . enc(proof2)
            usage = DST_QR_PART; k0 = ld64(qr_blind); k1 = ld64(qr_blind + 8);
            p1 = reinterpret_cast<const uint8_t *>(sc->jr); l1 = 16 * njr;
            p2 = d.proof1; l2 = 16u * P.proof_len; p3 = d.proof2; l3 = 16u * P.proof_len;
#if defined(WH_HASH_EXP) && WH_HASH_EXP == 2        // timing experiment (results invalid): the proof shares are not absorbed
            l2 = l3 = 0;
#endif
            break;
        case 5: usage = DST_QUERY_RAND; k0 = sd.lo; k1 = sd.hi; emit_n = 2; emit_p = reinterpret_cast<uint8_t *>(sc->qr); break;
        case 8:  // hash of the public parts: XOF(0, 10) . parts . parts2 . enc(pp0) . enc(pp1)
            usage = DST_HASH; p1 = ps_parts; l1 = psb; p2 = ps2_parts; l2 = psb; p3 = pp; l3 = 32u * P.verifier_len;
            break;
        case 9: usage = DST_HASH; k0 = sd.lo; k1 = sd.hi; emit_n = 1; emit_p = reinterpret_cast<uint8_t *>(&sc->hash); break;
        default:  // 10: RLC coefficients; only the first verifier_len + 4 + (num_aggregators - 1) are consumed (:2865-2871)
            usage = DST_RLC; k0 = vk0; k1 = vk1; p1 = nonce; l1 = 16;
            emit_n = rlc_len(P); emit_p = reinterpret_cast<uint8_t *>(W.rlc + (uint64_t)i * rlc_len(P));
            break;
        }
        if (!sub) s.init_xof(P.alg_id, usage, k0, k1);
        s.absorb3<WH_HASH_PREFETCH>(p1, l1, p2, l2, p3, l3);
        if (!last_sub) { sub += 3; stg--; continue; }       // more than three aggregators: the same sponge, the next parts
        sub = 0;
        s.finalize();
        if (emit_n) {
            xof_emit(s, emit_p, emit_n, P.sampler_bound);
            if (which && stg == 2) fold_hash_coefficient(P, W.rlc + (uint64_t)i * rlc_len(P), d.role);
        } else {
            sd = F128{s.st[0], s.st[1]};          // Xof::into_seed
            if (!which && stg == 3) {
                // the locally computed part vs the one in the public share (prio3.rs:2790-2805)
                sc->rand_diff[1] = !f_eq(sd, ld128(ps2_parts + 16 * d.role));
                st128(own, sd);
            }
        }
    }
    if (!which) {
        // flp.rs:413-419 for both runs: r^m == 1 is FlpError::Query
        F128 r1 = sc->qr[0], r = sc->qr[1];
        F128 r1m = f_sqr_n(r1, P.log2m), rm = f_sqr_n(r, P.log2m);
        if (f_eq(r1m, f_one()) || f_eq(rm, f_one())) { status[i] = ST_ERR_QUERY_RAND; return; }
#if !(defined(WH_HASH_EXP) && WH_HASH_EXP == 1)     // timing experiment (results invalid): no weights and tables
        weights_and_tables(P, W, i, r, rm, sc->jr[P.jr_len]);
#endif
    }
}

// ------------------------------------------------- draft-07 prepare_init: parse and hash ----
// VdafKey::decode_with_param (vdaf.rs:219-233) as far as structure goes:
//   agg_id u8 | public share: num_aggregators joint-rand parts | input share | nonce[16]
// input share (prio3.rs:1745-1759): Leader  meas[input_len] | proof[proof_len] | blind[16]
//                                   Helper  meas_seed[16] | proof_seed[16] | blind[16]
__global__ void k_parse_base(Prio3Dev P, Workspace W, const uint8_t *blob, const uint64_t *offsets, uint32_t n,
                             int32_t *status, uint8_t *prep_shares, uint8_t *out_shares) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *rep = blob + offsets[i];
    // a descending offset pair is an empty report (ST_ERR_SHORT), not a length of ~2^64
    uint64_t rep_len = offsets[i + 1] >= offsets[i] ? offsets[i + 1] - offsets[i] : 0;
    RepDesc d;
    d.base = rep; d.meas = d.proof1 = d.proof2 = d.tail = d.jr_blind = d.seeds = nullptr; d.role = 0; d.hslot = 0;
    int32_t st = ST_OK;
    if (rep_len < 1) st = ST_ERR_SHORT;
    if (!st) {
        d.role = rep[0];
        if (d.role >= P.num_aggregators) st = ST_ERR_AGG_ID;
    }
    if (!st) {
        const uint64_t head = 1 + 16ull * P.num_aggregators;
        const uint64_t share_len = d.role == 0 ? 16ull * (P.input_len + P.proof_len) + 16 : 48;
        if (rep_len < head + share_len + 16) st = ST_ERR_SHORT;
        if (!st) {
            d.tail = rep + head + share_len;            // nonce
            if (d.role == 0) {
                d.meas = rep + head;
                d.proof2 = d.meas + 16ull * P.input_len;
                d.jr_blind = d.proof2 + 16ull * P.proof_len;
                W.llist[atomicAdd(&W.counters[1], 1u)] = i;
            } else {
                d.hslot = atomicAdd(&W.counters[0], 1u);
                W.hlist[d.hslot] = i;
                uint8_t *row = W.hrows + (uint64_t)d.hslot * W.hrow_bytes;
                d.meas = row;
                d.proof2 = row + 16ull * P.input_len;
                d.seeds = rep + head;
                d.jr_blind = rep + head + 32;
            }
        }
    }
    if (st) {
        F128 *out_row = reinterpret_cast<F128 *>(out_shares) + (uint64_t)i * P.output_len;
        for (uint32_t e = 0; e < P.output_len; e++) out_row[e] = f_zero();
        F128 *prep = reinterpret_cast<F128 *>(prep_shares) + (uint64_t)i * P.ver_stride;
        for (uint32_t e = 0; e < P.ver_stride; e++) prep[e] = f_zero();
    }
    W.desc[i] = d;
    status[i] = st;
}

// Prio3::prepare_init up to the query (prio3.rs:2232-2300): query randomness XOF(verify_key, 5) .
// nonce; own joint-rand part XOF(blind, 7) . [agg_id] . nonce . enc(meas); joint-rand seed over
// the public share's parts with the own part substituted (:482-493); joint randomness XOF(seed, 3).
// The joint randomness goes where the batched path keeps run 2's, so the query kernels read it
// the same way; the seed is kept for prepare_next's check (:2407-2424).
// first_stage 1: the own joint-rand part is already in sc->own (k_meas computed it beside the Helper's share expansion).
__global__ void __launch_bounds__(256) k_hash_base(Prio3Dev P, Workspace W, uint64_t vk0, uint64_t vk1, int32_t *status, uint32_t first_stage) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i;
    if (!report_of_slot(W, slot, i) || status[i] != ST_OK) return;
    const RepDesc d = W.desc[i];
    RepScal *sc = &W.scal[i];
    const uint8_t *parts = d.base + 1, *nonce = d.tail;
    uint8_t *own = reinterpret_cast<uint8_t *>(&sc->own);
    SpongeG s;
    F128 sd = f_zero();
    uint32_t sub = 0;                   // seed stage: first part of this pass (three parts a pass, wave-uniform)
#pragma unroll 1
    for (uint32_t stg = first_stage; stg < 4; stg++) {
        bool last_sub = true;
        uint32_t usage = 0, l1 = 0, l2 = 0, l3 = 0, emit_n = 0;
        uint64_t k0 = 0, k1 = 0;
        const uint8_t *p1 = nullptr, *p2 = nullptr, *p3 = nullptr;
        uint8_t *emit_p = nullptr;
        switch (__builtin_amdgcn_readfirstlane(stg)) {
        case 0:
            usage = DST_JR_PART; k0 = ld64(d.jr_blind); k1 = ld64(d.jr_blind + 8);
            p1 = d.base; l1 = 1; p2 = nonce; l2 = 16; p3 = d.meas; l3 = 16u * P.input_len;
            break;
        case 1:
            usage = DST_JR_SEED;
            last_sub = seed_parts(parts, own, d.role, P.num_aggregators, sub, p1, l1, p2, l2, p3, l3);
            break;
        case 2:
            usage = DST_JOINT_RAND; k0 = sd.lo; k1 = sd.hi; emit_n = P.jr_len;
            emit_p = reinterpret_cast<uint8_t *>(sc->jr + P.jr_len);
            break;
        default:
            usage = DST_QUERY_RAND; k0 = vk0; k1 = vk1; p1 = nonce; l1 = 16; emit_n = 1;
            emit_p = reinterpret_cast<uint8_t *>(sc->qr + 1);
            break;
        }
        if (!sub) s.init_xof(P.alg_id, usage, k0, k1);
        s.absorb3(p1, l1, p2, l2, p3, l3);
        if (!last_sub) { sub += 3; stg--; continue; }
        sub = 0;
        s.finalize();
        if (emit_n) {
            xof_emit(s, emit_p, emit_n, P.sampler_bound);
        } else {
            sd = F128{s.st[0], s.st[1]};
            if (stg == 0) st128(own, sd);
            else sc->hash = sd;                     // the joint-rand seed
        }
    }
    const F128 r = sc->qr[1];
    const F128 rm = f_sqr_n(r, P.log2m);
    if (f_eq(rm, f_one())) { status[i] = ST_ERR_QUERY_RAND; return; }      // flp.rs:413-419
    weights_and_tables(P, W, i, r, rm, sc->jr[P.jr_len]);
}

// prepare_shares_to_prepare_message + prepare_next (prio3.rs:2347-2442) for a batch: sum the
// aggregators' verifier shares, decide (flp.rs:483-512), derive the joint-rand seed from the
// parts in the prepare shares and compare it with the one prepare_init used.  One wavefront per
// report.  shares[a] points at aggregator a's n prepare shares ((verifier_len + 1) elements each).
// the joint-rand seeds of a sub-batch, out of the workspace (which the next sub-batch reuses) into the prepare state
__global__ void k_keep_seeds(Workspace W, uint32_t nb, F128 *seeds) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb) seeds[i] = W.scal[i].hash;
}
__global__ void __launch_bounds__(256) k_finish_base(Prio3Dev P, const F128 *jr_seeds, const uint8_t *const *shares, uint32_t n,
                                                     uint8_t *prep_msgs, int32_t *status) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + wave;
    if (i >= n || status[i] != ST_OK) return;
    const uint32_t vl = P.verifier_len, na = P.num_aggregators;
    const uint64_t row = 16ull * (vl + 1) * i;
    auto ver = [&](uint32_t e, uint32_t &bad) -> F128 {        // sum of the shares' element e
        F128 acc = f_zero();
        for (uint32_t a = 0; a < na; a++) {
            F128 v = ld128g(shares[a] + row + 16ull * e);
            bad |= !f_in_range(v.lo, v.hi);
            acc = f_add(acc, v);
        }
        return acc;
    };
    uint32_t bad = 0;
    bool ok;
    if (P.circuit == CIRCUIT_AVERAGE) {
        F128 a = ver(0, bad), b = ver(1, bad), c = ver(2, bad);
        ok = f_is_zero(a) && f_eq(f_sub(f_sqr(b), b), c);
    } else {
        F128 acc = f_zero();
        for (uint32_t k = lane; k < P.chunk; k += 64) acc = f_add(acc, f_mul(ver(1 + 2 * k, bad), ver(2 + 2 * k, bad)));
        acc = wave_sum(acc);
        F128 first = ver(0, bad), last = ver(1 + P.arity, bad);
        ok = f_is_zero(first) && f_eq(acc, last);
    }
    bad = wave_or(bad);
    if (lane == 0) {
        // joint-rand seed over the parts the aggregators sent (prio3.rs:2380-2389)
        SpongeG s;
        s.init_xof(P.alg_id, DST_JR_SEED, 0ull, 0ull);
        for (uint32_t a = 0; a < na; a++) s.absorb_f(ld128(shares[a] + row + 16ull * vl));
        s.finalize();
        const F128 seed = F128{s.st[0], s.st[1]};
        st128(prep_msgs + 16ull * i, seed);
        int32_t st = ST_OK;
        if (bad) st = ST_ERR_MODULUS;
        else if (!ok) st = ST_ERR_DECIDE;
        else if (!f_eq(seed, jr_seeds[i])) st = ST_ERR_JOINT_RAND;
        status[i] = st;
    }
}

// ------------------------------------------------------ flp::Type::query / decide, bare ----
// Type::query (flp.rs:359-479) on caller-supplied inputs: input[n][input_len], proof[n][proof_len],
// query_rand[n][1], joint_rand[n][jr_len].  This kernel stands in for parse + hash: descriptors that
// point at the rows, the randomness where the query kernels read it, and the per-report tables.
__global__ void __launch_bounds__(256) k_flp_setup(Prio3Dev P, Workspace W, const uint8_t *input, const uint8_t *proof,
                                                   const uint8_t *query_rand, const uint8_t *joint_rand, uint32_t n,
                                                   int32_t *status) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { W.counters[1] = n; W.counters[0] = 0; }
    if (i >= n) return;
    W.llist[i] = i;
    RepDesc d;
    d.base = d.tail = d.jr_blind = d.seeds = d.proof1 = nullptr; d.role = 0; d.hslot = 0;
    d.meas = input + 16ull * P.input_len * i;
    d.proof2 = proof + 16ull * P.proof_len * i;
    W.desc[i] = d;
    RepScal *sc = &W.scal[i];
    int32_t st = ST_OK;
    const F128 r = ld128(query_rand + 16ull * i);
    if (!f_in_range(r.lo, r.hi)) st = ST_ERR_MODULUS;
    for (uint32_t q = 0; q < P.jr_len; q++) {
        const F128 v = ld128(joint_rand + 16ull * (P.jr_len * (uint64_t)i + q));
        if (!f_in_range(v.lo, v.hi)) st = ST_ERR_MODULUS;
        sc->jr[P.jr_len + q] = v;
    }
    sc->qr[0] = sc->qr[1] = r;
    sc->own = f_zero();
    if (!st) {
        const F128 rm = f_sqr_n(r, P.log2m);
        if (f_eq(rm, f_one())) st = ST_ERR_QUERY_RAND;                       // flp.rs:413-419
        else weights_and_tables(P, W, i, r, rm, sc->jr[P.jr_len]);
    }
    status[i] = st;
}

// Type::decide (flp.rs:483-512) on n verifiers of verifier_len elements: one wavefront each.
__global__ void __launch_bounds__(256) k_flp_decide(Prio3Dev P, const uint8_t *verifiers, uint32_t n, int32_t *ok_out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + wave;
    if (i >= n) return;
    const uint8_t *v = verifiers + 16ull * P.verifier_len * i;
    uint32_t bad = 0;
    bool ok;
    auto el = [&](uint32_t e) { F128 x = ld128(v + 16ull * e); bad |= !f_in_range(x.lo, x.hi); return x; };
    if (P.circuit == CIRCUIT_AVERAGE) {
        F128 a = el(0), b = el(1), c = el(2);
        ok = f_is_zero(a) && f_eq(f_sub(f_sqr(b), b), c);
    } else {
        F128 acc = f_zero();
        for (uint32_t k = lane; k < P.chunk; k += 64) acc = f_add(acc, f_mul(el(1 + 2 * k), el(2 + 2 * k)));
        acc = wave_sum(acc);
        F128 first = el(0), last = el(1 + P.arity);
        ok = f_is_zero(first) && f_eq(acc, last);
    }
    bad = wave_or(bad);
    if (lane == 0) ok_out[i] = bad ? ST_ERR_MODULUS : (ok ? 1 : 0);
}

// --------------------------------------------------------------------------------- query ----
// One workgroup of S = blockDim.x / 64 wavefronts per report.  Type::query for the
// ParallelSum<Mul> circuits (flp.rs:359-479, types.rs:623-639, 410-437, 742-777) in barycentric
// form: wire w's polynomial evaluated at r is sum_j L_j(r) f[w][j], with
// f[2k][j] = rho^(i+1) x_i, f[2k+1][j] = x_i - h, i = (j-1) chunk + k, f[w][0] = proof[w];
// v[0] = <s, G> (sum of the gadget polynomial over the call points).  Then decide on both public
// proofs (flp.rs:483-512), the differences, and the RLC tag (prio3.rs:2703-2871).
//
// The measurement share is streamed once.  Wave w owns the gadget slots k = 64 w + lane (+ 64 S,
// ...) and carries A_k = sum_j M_j x_(j,k) and B_k = sum_j L_j x_(j,k) as lazy column sums
// (f128.hpp), M_j = L_j rho^(chunk (j-1)); the weights (limbs, written by the hash kernel) sit in
// LDS and are read as broadcasts.  SumVec's truncate (types.rs:641-648) needs the same elements
// in the other direction (all bits of one output in one lane), so every wave parks the elements
// of eight calls in LDS as they arrive and turns them into outputs from there; shapes that do
// not tile this way are truncated from global memory at the end.  The gadget polynomial and the
// decide checks are done by the first and last wave; partial results meet in LDS.

constexpr uint32_t QT_CALLS = 8;                        // calls per truncate tile
constexpr uint32_t QT_STAGE_U4 = QT_CALLS * 64 * 17 / 16;   // uint4 slots per wave: 16 B of padding every 16 elements
__device__ __forceinline__ uint32_t stage_slot(uint32_t p) { return p + (p >> 4); }

// sum_l 2^l x[e bits + l] for outputs e0 + lane < e1, read from global memory
__device__ __forceinline__ void truncate_block(const Prio3Dev &P, const uint8_t *meas, F128 *out_row, uint32_t e0,
                                               uint32_t e1, uint32_t lane) {
    const uint32_t e = e0 + lane;
    if (e >= e1) return;
    const uint8_t *bits = meas + 16ull * e * P.bits;
    if (P.bits <= 32) {
        LazySmall acc;
        f_small_zero(acc);
        uint32_t l = 0;
        for (; l + 8 <= P.bits; l += 8) {       // eight loads in flight
            F128 x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = ld128g(bits + 16ull * (l + u));
#pragma unroll
            for (int u = 0; u < 8; u++) f_small_mac(acc, x[u], 1u << (l + u));
        }
        for (; l < P.bits; l++) f_small_mac(acc, ld128g(bits + 16ull * l), 1u << l);
        out_row[e] = f_small_reduce(acc);
    } else {
        F128 acc = f_zero();
        for (uint32_t l = P.bits; l-- > 0;) acc = f_add(f_dbl(acc), ld128g(bits + 16ull * l));
        out_row[e] = acc;
    }
}

struct QueryShared {      // behind the weight rows in LDS
    F128 tag[4], sum_x[4], dec[2];
    uint32_t bad[4];
};

// BASE: draft-07 prepare_init -- the verifier share itself is the output (`tags` then points at
// the prepare shares, (verifier_len + 1) elements per report: verifier | own joint-rand part).
#ifndef WH_Q_WAVES
#define WH_Q_WAVES 3
#endif
#ifdef WH_Q_EXP                             // timing experiments (results invalid): a phase left out
#define WH_Q_SKIP(n_) (WH_Q_EXP == (n_))
#else
#define WH_Q_SKIP(n_) false
#endif
template <bool HIST, bool BASE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WH_Q_WAVES, WH_Q_WAVES))) k_query_psum(Prio3Dev P, Workspace W, uint8_t *tags,
                                                    uint8_t *out_shares, int32_t *status) {
    extern __shared__ uint32_t lds_w[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, S = blockDim.x >> 6;
    uint32_t i;
    if (!report_of_slot(W, blockIdx.x, i)) return;
    uint8_t *tag_out = tags + 16ull * i;
    F128 *ver_row = reinterpret_cast<F128 *>(tags) + (uint64_t)i * P.ver_stride;            // BASE
    F128 *out_row = reinterpret_cast<F128 *>(out_shares) + (uint64_t)i * P.output_len;
    if (status[i] != ST_OK) {               // uniform over the workgroup
        for (uint32_t e = threadIdx.x; e < P.output_len; e += blockDim.x) out_row[e] = f_zero();
        if (BASE) for (uint32_t e = threadIdx.x; e < P.ver_stride; e += blockDim.x) ver_row[e] = f_zero();
        else if (threadIdx.x == 0) st128(tag_out, f_zero());
        return;
    }
    const uint32_t calls = P.calls;
    uint32_t *wl = lds_w;                   // [calls][12]: L limbs, M limbs, pad
    QueryShared *sh = reinterpret_cast<QueryShared *>(lds_w + 12 * (uint64_t)calls);
    uint4 *stage = reinterpret_cast<uint4 *>(sh + 1) + (uint64_t)wave * QT_STAGE_U4;
    const RepDesc d = W.desc[i];
    const RepScal *sc = &W.scal[i];
    const uint8_t *pp0 = d.tail + 16 * P.num_aggregators + 8, *pp1 = pp0 + 16ull * P.verifier_len;
    const F128 *rlc = W.rlc + (uint64_t)i * rlc_len(P);
    const bool leader = d.role == 0;
    uint32_t bad = 0;                           // any element >= p (FieldError::ModulusOverflow)

    // weight limbs (written by the hash kernel) to LDS
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(W.wlimbs + (uint64_t)i * 12 * calls);
        uint4 *dst = reinterpret_cast<uint4 *>(wl);
        for (uint32_t t = threadIdx.x; t < 3 * calls; t += blockDim.x) dst[t] = src[t];
    }
    const RepPow *pw = &W.pows[i];
    const F128 L0 = pw->L0, h_sumL = pw->h_sumL;
    const F128 rho_lane = f_mul(pw->rho_lo[lane & 7], pw->rho_hi[lane >> 3]);      // rho^(lane + 1)
    F128 rho_kb = pw->rho64[wave];                                                  // rho^kb
    __syncthreads();

    F128 tag = f_zero();
    LazySmall sum_x;                            // Histogram's sum check
    f_small_zero(sum_x);
    uint32_t top = 0;                           // largest top word seen; all ones: look again, exactly
    const bool trunc = P.circuit == CIRCUIT_SUMVEC;
    // truncate through LDS: the 64 slots of a wave hold whole outputs, 8 consecutive elements per lane
    const bool lds_trunc = trunc && (P.bits == 8 || P.bits == 16 || P.bits == 32) && (P.chunk & 63) == 0;
    const uint32_t lpo = P.bits >> 3;           // lanes per output in that case

    // every call is whole and every wave's 64 slots exist: no bounds to check inside the stream (uniform over the launch)
    const bool fast = P.input_len == calls * P.chunk && (P.chunk & 63) == 0 && (calls & 7) == 0;
    for (uint32_t kb = 64 * wave; kb < P.chunk; kb += 64 * S) {
        const uint32_t k = kb + lane;
        Lazy cA, cB;
        f_lazy_zero(cA);
        f_lazy_zero(cB);
        F128 Af = f_zero(), Bf = f_zero();
        uint32_t since = 0;
        uint32_t xl[4][5];
        // element u of the group of four that starts at call jj: range pre-check, limbs, Histogram's sum, truncate staging
        // (c = position of the call inside its tile of eight; stage_slot(64 c + lane) = 68 c + lane + lane / 16)
        uint4 *stage_lane = stage + lane + (lane >> 4);
        auto take = [&](uint32_t c, int u, const F128 &v) {
            top = max(top, (uint32_t)(v.hi >> 32));
            f_split26(v, xl[u]);
            if (HIST) f_small_mac(sum_x, v, 1u);
            if (lds_trunc)
                stage_lane[68 * c] = make_uint4((uint32_t)v.lo, (uint32_t)(v.lo >> 32), (uint32_t)v.hi, (uint32_t)(v.hi >> 32));
        };
        // the two wire sums of the four elements, weights of calls jj .. jj + 3 from LDS (broadcast reads)
        auto wires = [&](uint32_t jj, bool clamp) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t jc = !clamp || jj + u < calls ? jj + u : calls - 1;   // a call past the end has x = 0
                const uint4 *src = reinterpret_cast<const uint4 *>(wl + 12 * jc);
                const uint4 w0 = src[0], w1 = src[1], w2 = src[2];
                const uint32_t wL[5] = {w0.x, w0.y, w0.z, w0.w, w1.x};
                const uint32_t wM[5] = {w1.y, w1.z, w1.w, w2.x, w2.y};
#if defined(WH_Q_EXP) && WH_Q_EXP == 1     // timing experiment (results invalid): no wire arithmetic
                cA.c[0] += xl[u][0] ^ wM[0]; cB.c[0] += xl[u][4] ^ wL[0];
#else
                f_lazy_mac(cA, xl[u], wM);
                f_lazy_mac(cB, xl[u], wL);
#endif
            }
            since += 4;
            if (since + 4 > LAZY_MAX_TERMS) {
                Af = f_add(Af, f_lazy_reduce(cA));
                Bf = f_add(Bf, f_lazy_reduce(cB));
                f_lazy_zero(cA);
                f_lazy_zero(cB);
                since = 0;
            }
        };
        // a tile of eight calls is parked: lane handles positions 8 lane .. 8 lane + 7
        // (call t = lane / 8 of the tile, slots 8 (lane % 8) ..), lpo lanes make one output
        auto tile = [&](uint32_t jj) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const uint32_t j0 = jj & ~7u, t = lane >> 3, kk0 = (lane & 7) * 8;
            const uint32_t l0 = kk0 & (P.bits - 1);
            LazySmall acc;
            f_small_zero(acc);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint4 v = stage[stage_slot(lane * 8 + q)];
                f_small_mac(acc, F128{v.x | ((uint64_t)v.y << 32), v.z | ((uint64_t)v.w << 32)}, 1u << (l0 + q));
            }
            for (uint32_t m = 1; m < lpo; m <<= 1)
#pragma unroll
                for (int q = 0; q < 4; q++) acc.c[q] += __shfl_xor((unsigned long long)acc.c[q], m, 64);
            const uint64_t idx0 = (uint64_t)(j0 + t) * P.chunk + kb + (kk0 - l0);
            const uint64_t e = idx0 / P.bits;
            if ((lane & (lpo - 1)) == 0 && j0 + t < calls && e < P.output_len) out_row[e] = f_small_reduce(acc);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        };
        if (fast) {
            const uint64_t stride = 16ull * P.chunk;
#if defined(WH_Q_EXP) && WH_Q_EXP == 2     // timing experiment (results invalid): every load hits the same lines
            const uint8_t *p = d.meas + 16ull * lane;
#define WH_Q_ADV(p_) (void)0
#else
            const uint8_t *p = d.meas + 16ull * k;
#define WH_Q_ADV(p_) p_ += 4 * stride
#endif
            F128 xb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) xb[u] = ld128g(p + u * stride);
            WH_Q_ADV(p);
            for (uint32_t jj = 0; jj < calls; jj += 8) {
                // cut the four elements that have arrived into limbs, put the next four loads in
                // flight straight away, then do the arithmetic under them; twice per tile of eight calls
#pragma unroll
                for (int u = 0; u < 4; u++) take(u, u, xb[u]);
#pragma unroll
                for (int u = 0; u < 4; u++) xb[u] = ld128g(p + u * stride);
                WH_Q_ADV(p);
                wires(jj, false);
#pragma unroll
                for (int u = 0; u < 4; u++) take(4 + u, u, xb[u]);
                if (jj + 8 < calls) {
#pragma unroll
                    for (int u = 0; u < 4; u++) xb[u] = ld128g(p + u * stride);
                    WH_Q_ADV(p);
                }
                wires(jj + 4, false);
#if !(defined(WH_Q_EXP) && WH_Q_EXP == 3)     // timing experiment (results invalid): no truncate tiles
                if (lds_trunc) tile(jj + 4);
#endif
            }
#undef WH_Q_ADV
        } else {
            // element of call jj (0-based) in this lane's slot, or element 0 (masked) outside the share
            auto fetch = [&](uint32_t jj, bool &ok) -> F128 {
                const uint64_t idx = (uint64_t)jj * P.chunk + k;
                ok = jj < calls && k < P.chunk && idx < P.input_len;
                return ld128g(d.meas + 16ull * (ok ? idx : 0));
            };
            F128 xb[4];
            bool okb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) xb[u] = fetch(u, okb[u]);
            for (uint32_t jj = 0; jj < calls; jj += 4) {
#pragma unroll
                for (int u = 0; u < 4; u++) take((jj & 4) + u, u, okb[u] ? xb[u] : f_zero());
#pragma unroll
                for (int u = 0; u < 4; u++) xb[u] = fetch(jj + u + 4, okb[u]);
                wires(jj, true);
                if (lds_trunc && ((jj & 4) || jj + 4 >= calls)) tile(jj);
            }
        }
        // this lane's slot: seeds (call 0), differences against the public proof, RLC
        if (k < P.chunk && !WH_Q_SKIP(6)) {
            // every load of this block first, in one go: left to the compiler they sit in front of their uses, one memory
            // latency after the other (six per slot block)
            F128 s_even = ld128g(d.proof2 + 16ull * (2 * k)), s_odd = ld128g(d.proof2 + 16ull * (2 * k + 1));
            F128 q_even = f_zero(), q_odd = f_zero(), c_even = f_zero(), c_odd = f_zero();
            if (!BASE) {
                q_even = ld128g(pp1 + 16ull * (1 + 2 * k)); q_odd = ld128g(pp1 + 16ull * (2 + 2 * k));
                c_even = ld128g(reinterpret_cast<const uint8_t *>(rlc + 1 + 2 * k)); c_odd = ld128g(reinterpret_cast<const uint8_t *>(rlc + 2 + 2 * k));
            }
            __builtin_amdgcn_sched_barrier(0);
            const F128 A = f_add(Af, f_lazy_reduce(cA));
            const F128 B = f_add(Bf, f_lazy_reduce(cB));
            const F128 rk = f_mul(rho_lane, rho_kb);                                    // rho^(k + 1)
            bad |= !f_in_range(s_even.lo, s_even.hi) | !f_in_range(s_odd.lo, s_odd.hi);
            F128 v_even = f_add(f_mul(L0, s_even), f_mul(rk, A));
            F128 v_odd = f_sub(f_add(f_mul(L0, s_odd), B), h_sumL);
            if (BASE) {
                ver_row[1 + 2 * k] = v_even;
                ver_row[2 + 2 * k] = v_odd;
            } else {
                bad |= !f_in_range(q_even.lo, q_even.hi) | !f_in_range(q_odd.lo, q_odd.hi);
                if (leader) { v_even = f_sub(v_even, q_even); v_odd = f_sub(v_odd, q_odd); }
                tag = f_add(tag, f_add(f_mul(c_even, v_even), f_mul(c_odd, v_odd)));
            }
        }
        if (kb + 64 * S < P.chunk) rho_kb = f_mul(rho_kb, pw->rho64[S]);
    }
    // the exact range check of the share, in the rare case that a top word was all ones
    if (wave_or(top == 0xffffffffu))
        for (uint32_t e = lane; e < P.input_len; e += 64) {
            F128 x = ld128g(d.meas + 16ull * e);
            bad |= !f_in_range(x.lo, x.hi);
        }
    // truncate (types.rs:641-648 / 439-442) -> out share, where it was not done on the way
    if (trunc && !lds_trunc) {
        for (uint32_t e0 = 64 * wave; e0 < P.output_len; e0 += 64 * S)
            truncate_block(P, d.meas, out_row, e0, e0 + 64 < P.output_len ? e0 + 64 : P.output_len, lane);
    } else if (!trunc) {
        for (uint32_t e = threadIdx.x; e < P.output_len; e += blockDim.x) out_row[e] = ld128g(d.meas + 16ull * e);
    }

    // first wave: gadget polynomial G = proof2[arity ..]: G(r) and <s, G>
    F128 g_at_r = f_zero(), v0 = f_zero();
    if (wave == 0 && !WH_Q_SKIP(5)) {
        const uint8_t *G = d.proof2 + 16ull * P.arity;
        Lazy cg, cv;
        f_lazy_zero(cg);
        f_lazy_zero(cv);
        F128 rpow = f_mul(pw->r_lo[lane & 7], pw->r_hi[lane >> 3]), r64 = pw->r64;
        uint32_t since = 0;
        // eight coefficients (and their s-table entries) in flight per lane: one memory latency per 512 coefficients
        for (uint32_t e0 = lane; e0 < P.gpoly_len; e0 += 512) {
            F128 cc[8], ss[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t e = e0 + 64 * u, ec = e < P.gpoly_len ? e : 0;
                cc[u] = ld128g(G + 16ull * ec);
                ss[u] = P.s_tab[ec];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (e0 + 64 * u >= P.gpoly_len) break;
                const F128 c = cc[u];
                bad |= !f_in_range(c.lo, c.hi);
                uint32_t cl[5], t1[5], t2[5];
                f_split26(c, cl);
                f_split26(rpow, t1);
                f_split26(ss[u], t2);
                f_lazy_mac(cg, cl, t1);
                f_lazy_mac(cv, cl, t2);
                rpow = f_mul(rpow, r64);
                if (++since == LAZY_MAX_TERMS) {
                    g_at_r = f_add(g_at_r, f_lazy_reduce(cg));
                    v0 = f_add(v0, f_lazy_reduce(cv));
                    f_lazy_zero(cg);
                    f_lazy_zero(cv);
                    since = 0;
                }
            }
        }
        g_at_r = wave_sum(f_add(g_at_r, f_lazy_reduce(cg)));
        v0 = wave_sum(f_add(v0, f_lazy_reduce(cv)));
    }
    // last wave: decide on the two public proofs (prio3.rs:2762-2786): pp[0] == 0 and
    // sum_k pp[1+2k] pp[2+2k] == pp[1+arity]; Leader only: the proof share of run 1 is hashed,
    // never evaluated, but it still has to decode
    if (!BASE && wave == S - 1) {
        if (leader && !WH_Q_SKIP(4))
            for (uint32_t e = lane; e < P.proof_len; e += 64) {
                F128 v = ld128g(d.proof1 + 16ull * e);
                bad |= !f_in_range(v.lo, v.hi);
            }
        // first and last element of both public proofs and this lane's first pair of each: eight loads in one go
        const uint32_t k0 = lane < P.chunk ? lane : 0;
        F128 fl[2][2], ab0[2][2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint8_t *pp = t ? pp1 : pp0;
            fl[t][0] = ld128g(pp); fl[t][1] = ld128g(pp + 16ull * (1 + P.arity));
            ab0[t][0] = ld128g(pp + 16ull * (1 + 2 * k0)); ab0[t][1] = ld128g(pp + 16ull * (2 + 2 * k0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint8_t *pp = t ? pp1 : pp0;
            F128 acc = f_zero();
            for (uint32_t kk = lane; kk < P.chunk; kk += 64) {
                F128 a = ab0[t][0], b = ab0[t][1];
                if (kk != lane) { a = ld128g(pp + 16ull * (1 + 2 * kk)); b = ld128g(pp + 16ull * (2 + 2 * kk)); }
                bad |= !f_in_range(a.lo, a.hi) | !f_in_range(b.lo, b.hi);
                acc = f_add(acc, f_mul(a, b));
            }
            acc = wave_sum(acc);
            F128 first = fl[t][0], last = fl[t][1];
            bad |= !f_in_range(first.lo, first.hi) | !f_in_range(last.lo, last.hi);
            bool ok = f_is_zero(first) && f_eq(acc, last);
            if (lane == 0) sh->dec[t] = ok ? f_zero() : f_one();
        }
    }
    tag = wave_sum(tag);
    bad = wave_or(bad);
    F128 sx = f_zero();
    if (HIST) sx = wave_sum(f_small_reduce(sum_x));
    if (lane == 0) {
        sh->tag[wave] = tag;
        sh->sum_x[wave] = sx;
        sh->bad[wave] = bad;
    }
    // the last step's operands, one set per lane te < 8 of the first wave, fetched ahead of the barrier:
    //   te 0: rlc[0] (v[0] - pp1[0])      te 1: rlc[1 + arity] (G(r) - pp1[1 + arity])   te 2, 3: decide flags -> rlc[vl], rlc[vl + 1]
    //   te 4, 5: rand_diff -> rlc[vl + 2], rlc[vl + 3]      te 6: rlc[vl + 4] (+-hash)      te 7: the wire part of the tag
    const uint32_t vl_ = P.verifier_len, te = lane & 7;
    F128 t_rlc = f_zero(), t_q = f_zero(), hs = f_zero();
    uint32_t t_flag = 0;
    if (!BASE && wave == 0) {
        t_rlc = rlc[te == 0 ? 0 : (te == 1 ? 1 + P.arity : vl_ + (te < 7 ? te - 2 : 0))];
        t_q = ld128g(pp1 + 16ull * (te == 1 ? 1 + P.arity : 0));
        hs = sc->hash;
        t_flag = sc->rand_diff[te == 5];
    }
    __syncthreads();
    bad = 0;
    for (uint32_t w = 0; w < S; w++) bad |= sh->bad[w];
    if (bad) {
        for (uint32_t e = threadIdx.x; e < P.output_len; e += blockDim.x) out_row[e] = f_zero();
        if (BASE) for (uint32_t e = threadIdx.x; e < P.ver_stride; e += blockDim.x) ver_row[e] = f_zero();
    }
    if (wave != 0) return;
    tag = f_zero();
    sx = f_zero();
    for (uint32_t w = 0; w < S; w++) { tag = f_add(tag, sh->tag[w]); sx = f_add(sx, sh->sum_x[w]); }
    if (HIST) {
        // types.rs:428-436: rho' * range_check + rho'^2 * (sum x - h)
        const F128 rho2 = sc->jr[P.jr_len + 1];
        v0 = f_add(f_mul(rho2, v0), f_mul(f_sqr(rho2), f_sub(sx, P.half)));
    }
    if (BASE) {
        // Prio3PrepareShare (prio3.rs:1953-1970): verifier share | joint-rand part
        if (lane == 0) {
            if (bad) status[i] = ST_ERR_MODULUS;
            else { ver_row[0] = v0; ver_row[1 + P.arity] = g_at_r; if (P.ver_stride > vl_) ver_row[vl_] = sc->own; }
        }
        return;
    }
    // eight lanes side by side, one multiplication deep (it was one lane doing ten in a row)
    const F128 dv = te == 0 ? v0 : g_at_r;
    const F128 mb = te < 2 ? (leader ? f_sub(dv, t_q) : dv) : (leader ? f_neg(hs) : hs);     // share of zero: Leader -H, Helper +H (prio3.rs:2844-2855)
    F128 term = f_mul(t_rlc, mb);
    if (te == 2 || te == 3) term = f_is_zero(sh->dec[te - 2]) ? f_zero() : t_rlc;
    if (te == 4 || te == 5) term = t_flag ? t_rlc : f_zero();
    if (te == 7) term = tag;
#pragma unroll
    for (int msk = 4; msk >= 1; msk >>= 1) term = f_add(term, shfl_xor_f(term, msk));
    if (lane == 0) {
        if (bad) { status[i] = ST_ERR_MODULUS; term = f_zero(); }
        // prepare_encoded_key: the Helper-role tag is negated (main.rs:53-59)
        st128(tag_out, leader ? term : f_neg(term));
    }
}

// Average (Sum) circuit: one PolyEval(x^2 - x) gadget of arity 1 called once per input bit
// (types.rs:241-313, call_gadget_on_vec_entries :686-698, gadgets.rs:185-188).  One wavefront per
// report; m <= 128 so a wave holds the whole problem.
//   v[0] = sum_{i=1..calls} rho^i G(w^i)       -- G at the call points by folding G to m
//                                                 coefficients and one m-point NTT in LDS
//   v[1] = L_0 proof[0] + sum_j L_j x_{j-1}    -- the single wire polynomial at r
//   v[2] = G(r)
// decide on a public proof: pp[0] == 0 and pp[1]^2 - pp[1] == pp[2].
// Shape of the work: every load of the report is issued before the first use (one memory latency, not one per phase);
// the powers of rho and r come from the tables the chain / hash lane left (RepPow), never from a per-lane
// exponentiation; G(r) is Horner over the <= 4 coefficients of a lane times r^lane; the four 64-lane sums (G(r), v[0], v[1],
// the out share) go through one reduce-scatter (7 additions instead of 24); the last step's eight products and checks
// sit in eight lanes side by side (one multiplication deep) instead of one lane doing ten in a row.
__device__ __forceinline__ F128 sel_f(bool c, const F128 &a, const F128 &b) { return F128{c ? a.lo : b.lo, c ? a.hi : b.hi}; }
__device__ __forceinline__ F128 readlane_f(const F128 &v, int l) {
    const uint32_t a = __builtin_amdgcn_readlane((int)(uint32_t)v.lo, l), b = __builtin_amdgcn_readlane((int)(uint32_t)(v.lo >> 32), l);
    const uint32_t c = __builtin_amdgcn_readlane((int)(uint32_t)v.hi, l), d = __builtin_amdgcn_readlane((int)(uint32_t)(v.hi >> 32), l);
    return F128{a | ((uint64_t)b << 32), c | ((uint64_t)d << 32)};
}
template <bool BASE>
__global__ void __launch_bounds__(256) k_query_polyeval(Prio3Dev P, Workspace W, uint8_t *tags,
                                                        uint8_t *out_shares, int32_t *status) {
    extern __shared__ F128 lds[];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t m = P.m, lg = P.log2m, calls = P.calls;
    // w^k, k < m / 2, for the butterflies: once per workgroup
    F128 *tw = lds + 4ull * m;
    for (uint32_t k = threadIdx.x; k < (m >> 1); k += blockDim.x) tw[k] = P.omega[k];
    __syncthreads();
    const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + wave;
    uint32_t i;
    if (!report_of_slot(W, slot, i)) return;
    i = __builtin_amdgcn_readfirstlane(i);          // wave-uniform: what hangs off it comes through scalar loads
    uint8_t *tag_out = tags + 16ull * i;
    F128 *ver_row = reinterpret_cast<F128 *>(tags) + (uint64_t)i * P.ver_stride;   // BASE: verifier[3] (| joint-rand part)
    F128 *out_row = reinterpret_cast<F128 *>(out_shares) + (uint64_t)i;   // output_len == 1
    if (status[i] != ST_OK) {
        if (lane == 0) {
            out_row[0] = f_zero();
            if (BASE) { for (uint32_t e = 0; e < P.ver_stride; e++) ver_row[e] = f_zero(); }
            else st128(tag_out, f_zero());
        }
        return;
    }
    F128 *buf = lds + (uint64_t)wave * m;
    const RepDesc d = W.desc[i];
    const RepScal *sc = &W.scal[i];
    const RepPow *pw = &W.pows[i];
    const uint8_t *pp0 = d.tail + 16 * P.num_aggregators + 8, *pp1 = pp0 + 16ull * P.verifier_len;
    const uint8_t *rlc = reinterpret_cast<const uint8_t *>(W.rlc + (uint64_t)i * rlc_len(P));
    const uint8_t *Lg = reinterpret_cast<const uint8_t *>(W.weights + (uint64_t)i * P.n_weights);
    const bool leader = d.role == 0;
    uint32_t bad = 0;

    // ---- all loads of the report ----
    const uint8_t *G = d.proof2 + 16ull * P.arity;
    F128 gc[4];                                     // G_e, e = lane + 64 u  (gpoly_len = 2 m - 1 <= 255)
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t e = lane + 64 * u;
        gc[u] = ld128g(G + 16ull * (e < P.gpoly_len ? e : 0));
        if (e >= P.gpoly_len) gc[u] = f_zero();
    }
    F128 xv[2], lw[2];                              // x_(j-1) and L_j, j = 1 + lane + 64 u
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint32_t j = 1 + lane + 64 * u, jc = j <= calls ? j : 1;
        xv[u] = ld128g(d.meas + 16ull * (jc - 1));
        lw[u] = ld128g(Lg + 16ull * jc);
        if (j > calls) xv[u] = f_zero();
    }
    const F128 seed0 = ld128g(d.proof2), L0 = ld128g(Lg);
    const F128 rho_a = pw->rho_lo[lane & 7], rho_b = pw->rho_hi[lane >> 3], rho64 = pw->rho64[1];   // rho^((lane & 7) + 1), rho^(8 (lane >> 3))
    const F128 r_a = pw->r_lo[lane & 7], r_b = pw->r_hi[lane >> 3], r64 = pw->r64;
    // the last step, one operand set per lane e < 8:
    //   e = 0..2  rlc[e] * (ver[e] - pp1[e])       e = 3, 4  decide(pp0 / pp1) -> rlc[e] or 0
    //   e = 5, 6  rand_diff[e - 5] -> rlc[e] or 0   e = 7     rlc[7] * (+-hash)
    const uint32_t te = lane & 7;
    F128 t_rlc = f_zero(), t_a = f_zero(), t_b = f_zero(), t_c = f_zero();
    uint32_t t_flag = 0;
    if (!BASE) {
        t_rlc = ld128g(rlc + 16ull * te);
        const uint8_t *pp = te == 3 ? pp0 : pp1;
        t_a = ld128g(pp + 16ull * (te < 3 ? te : 0));          // e < 3: pp1[e];  e = 3, 4: pp[0]
        t_b = ld128g(pp + 16);
        t_c = ld128g(pp + 32);
        t_flag = sc->rand_diff[te == 6];
        if (te == 7) t_b = sc->hash;
    }
    if (!BASE && leader)
        for (uint32_t e = lane; e < P.proof_len; e += 64) {       // proof share of run 1: hashed, never evaluated, but it has to decode
            const F128 v = ld128g(d.proof1 + 16ull * e);
            bad |= !f_in_range(v.lo, v.hi);
        }
    __builtin_amdgcn_sched_barrier(0);

    // ---- G folded mod (X^m - 1) into LDS, bit-reversed; G(r) = sum_lane r^lane Horner_{r^64}(G_lane, G_lane+64, ...) ----
#pragma unroll
    for (int u = 0; u < 4; u++) bad |= !f_in_range(gc[u].lo, gc[u].hi);
    {
        // coefficient e + m sits in slot u + 2 (m = 128), in slot 1 (m = 64), or in lane e + m's slot 0 (m <= 32); slots
        // past the polynomial's end hold zero
        F128 up = gc[1];
        if (m < 64) up = F128{__shfl((unsigned long long)gc[0].lo, (int)((lane + m) & 63), 64), __shfl((unsigned long long)gc[0].hi, (int)((lane + m) & 63), 64)};
        const F128 fold0 = f_add(gc[0], m == 128 ? gc[2] : up);
        if (lane < m) buf[__brev(lane) >> (32 - lg)] = fold0;
        if (m == 128) buf[__brev(lane + 64) >> (32 - lg)] = f_add(gc[1], gc[3]);
    }
    F128 g_p = gc[0];
    if (P.gpoly_len > 64) {            // uniform
        g_p = gc[1];
        if (P.gpoly_len > 128) g_p = f_add(f_mul(f_add(f_mul(gc[3], r64), gc[2]), r64), gc[1]);
        g_p = f_add(f_mul(g_p, r64), gc[0]);
    }
    g_p = f_mul(g_p, f_mul(r_a, r_b));                                          // r^lane
    // ---- in-place radix-2 NTT, one butterfly per lane and stage (m / 2 <= 64 butterflies) ----
    for (uint32_t l = 1; l <= lg; l++) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint32_t half = 1u << (l - 1);
        if (lane < (m >> 1)) {
            const uint32_t k = lane & (half - 1), base = ((lane >> (l - 1)) << l) + k;
            const F128 u = buf[base], v = f_mul(buf[base + half], tw[k << (lg - l)]);
            buf[base] = f_add(u, v);
            buf[base + half] = f_sub(u, v);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- v[0], v[1], truncate (field.rs:256-269): terms j = 1 + lane (+ 64) ----
    F128 v0_p = f_zero(), v1_p = f_zero(), out_p = f_zero();
    const F128 rp = f_mul(rho_a, rho_b);                                        // rho^(lane + 1)
    {
        const uint32_t j = 1 + lane;
        bad |= !f_in_range(xv[0].lo, xv[0].hi);
        const F128 a0 = f_mul(rp, buf[j <= calls ? j : 0]), a1 = f_mul(lw[0], xv[0]);
        const F128 a2 = f_mul(lane < 64 ? F128{1ull << lane, 0} : f_zero(), xv[0]);
        if (j <= calls) { v0_p = a0; v1_p = a1; out_p = a2; }
    }
    if (calls > 64) {                  // uniform: bits up to 127
        const uint32_t j = 65 + lane;
        bad |= !f_in_range(xv[1].lo, xv[1].hi);
        const F128 a0 = f_mul(f_mul(rp, rho64), buf[j <= calls ? j : 0]), a1 = f_mul(lw[1], xv[1]);
        const F128 a2 = f_mul(F128{0, 1ull << lane}, xv[1]);
        if (j <= calls) { v0_p = f_add(v0_p, a0); v1_p = f_add(v1_p, a1); out_p = f_add(out_p, a2); }
    }
    bad |= !f_in_range(seed0.lo, seed0.hi);
    {
        const F128 sterm = f_mul(L0, seed0);                                    // the wire's seed (call 0)
        if (lane == 0) v1_p = f_add(v1_p, sterm);
    }
    // ---- the four sums in one reduce-scatter: rows of 16 lanes end up with G(r), v[0], v[1], out ----
    F128 red;
    {
        const bool h5 = lane & 32, h4 = lane & 16;
        F128 ka = f_add(sel_f(h5, v1_p, g_p), shfl_xor_f(sel_f(h5, g_p, v1_p), 32));
        F128 kb = f_add(sel_f(h5, out_p, v0_p), shfl_xor_f(sel_f(h5, v0_p, out_p), 32));
        red = f_add(sel_f(h4, kb, ka), shfl_xor_f(sel_f(h4, ka, kb), 16));
#pragma unroll
        for (int msk = 8; msk >= 1; msk >>= 1) red = f_add(red, shfl_xor_f(red, msk));
    }
    const F128 g_at_r = readlane_f(red, 0), v0 = readlane_f(red, 16), v1 = readlane_f(red, 32);
    F128 out = readlane_f(red, 48);
    if (BASE) {
        bad = wave_or(bad);
        if (lane == 0) {
            F128 a0 = v0, a1 = v1, a2 = g_at_r;
            if (bad) { status[i] = ST_ERR_MODULUS; out = f_zero(); a0 = a1 = a2 = f_zero(); }
            out_row[0] = out;
            ver_row[0] = a0; ver_row[1] = a1; ver_row[2] = a2;
            if (P.ver_stride > 3) ver_row[3] = bad ? f_zero() : sc->own;
        }
        return;
    }
    // ---- differences, decide, RLC tag (prio3.rs:2703-2871), eight lanes side by side ----
    const bool dec_lane = te == 3 || te == 4;
    F128 ma = t_rlc, mb;
    {
        const F128 ver = te == 0 ? v0 : (te == 1 ? v1 : g_at_r);
        const F128 dv = leader ? f_sub(ver, t_a) : ver;
        mb = te < 3 ? dv : (leader ? f_neg(t_b) : t_b);                          // e = 7: share of zero, Leader -H, Helper +H (:2844-2855)
        if (dec_lane) { ma = t_b; mb = t_b; }                                    // pp[1]^2
        if (te < 5) bad |= !f_in_range(t_a.lo, t_a.hi);
        if (dec_lane) bad |= !f_in_range(t_b.lo, t_b.hi) | !f_in_range(t_c.lo, t_c.hi);
    }
    F128 term = f_mul(ma, mb);
    if (dec_lane) {
        const bool ok = f_is_zero(t_a) && f_eq(f_sub(term, t_b), t_c);
        term = ok ? f_zero() : t_rlc;
    } else if (te == 5 || te == 6) {
        term = t_flag ? t_rlc : f_zero();
    }
#pragma unroll
    for (int msk = 4; msk >= 1; msk >>= 1) term = f_add(term, shfl_xor_f(term, msk));
    bad = wave_or(bad);
    if (lane == 0) {
        F128 tag = term;
        if (bad) { status[i] = ST_ERR_MODULUS; tag = f_zero(); out = f_zero(); }
        out_row[0] = out;
        st128(tag_out, leader ? tag : f_neg(tag));
    }
}

// ParallelSum circuits with at most QTINY_MAX_CALLS (two) gadget calls of at most 64 slots (Histogram of 128 with chunk 64: two
// calls): the same Type::query + decide + RLC tag as k_query_psum (flp.rs:359-512, types.rs:410-437, 623-639,
// prio3.rs:2703-2871), one WAVEFRONT per report.  k_query_psum streams a long share through a pipeline; for a share of a few
// elements per lane that pipeline is a chain of ten dependent memory latencies around ~3 000 instructions (1.1 ms per 100 000
// Histogram reports, 0.16 of the HBM rate).  Here every byte of the report a lane needs is requested before the first use --
// the elements of its slot, its seeds, both public-proof pairs, its RLC coefficients, its gadget coefficient, the tables of
// the chain kernel -- the six 64-lane sums share one reduce-scatter, and the last step's products sit in eight lanes.
constexpr uint32_t QTINY_MAX_CALLS = 2;
#ifdef WH_QTINY_WAVES
#define WH_QTINY_ATTR __attribute__((amdgpu_waves_per_eu(WH_QTINY_WAVES, WH_QTINY_WAVES)))
#else
#define WH_QTINY_ATTR
#endif
template <bool HIST, bool BASE>
__global__ void __launch_bounds__(256) WH_QTINY_ATTR k_query_psum_tiny(Prio3Dev P, Workspace W, uint8_t *tags, uint8_t *out_shares, int32_t *status) {
    // the report index is wave-uniform: said so explicitly, everything that hangs off it (descriptor, tables, weight limbs,
    // scalars) then comes through scalar loads into SGPRs instead of one VGPR copy per lane
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + wave;
    uint32_t i;
    if (!report_of_slot(W, slot, i)) return;
    i = __builtin_amdgcn_readfirstlane(i);
    uint8_t *tag_out = tags + 16ull * i;
    F128 *ver_row = reinterpret_cast<F128 *>(tags) + (uint64_t)i * P.ver_stride;            // BASE
    F128 *out_row = reinterpret_cast<F128 *>(out_shares) + (uint64_t)i * P.output_len;
    if (status[i] != ST_OK) {               // uniform over the wavefront
        for (uint32_t e = lane; e < P.output_len; e += 64) out_row[e] = f_zero();
        if (BASE) for (uint32_t e = lane; e < P.ver_stride; e += 64) ver_row[e] = f_zero();
        else if (lane == 0) st128(tag_out, f_zero());
        return;
    }
    const uint32_t chunk = P.chunk, calls = P.calls, arity = P.arity, vl = P.verifier_len;
    const RepDesc d = W.desc[i];
    const RepScal *sc = &W.scal[i];
    const RepPow *pw = &W.pows[i];
    const uint8_t *pp0 = d.tail + 16 * P.num_aggregators + 8, *pp1 = pp0 + 16ull * vl;
    const uint8_t *rlc = reinterpret_cast<const uint8_t *>(W.rlc + (uint64_t)i * rlc_len(P));
    const bool leader = d.role == 0;
    const bool act = lane < chunk;
    const uint32_t k = act ? lane : 0;
    uint32_t bad = 0;

    // ---- every load of the report ----
    F128 x[QTINY_MAX_CALLS];
    bool xok[QTINY_MAX_CALLS];
#pragma unroll
    for (uint32_t u = 0; u < QTINY_MAX_CALLS; u++) {
        const uint32_t idx = u * chunk + k;
        xok[u] = u < calls && act && idx < P.input_len;
        x[u] = ld128g(d.meas + 16ull * (xok[u] ? idx : 0));
        if (!xok[u]) x[u] = f_zero();
    }
    const F128 s_even = ld128g(d.proof2 + 16ull * (2 * k)), s_odd = ld128g(d.proof2 + 16ull * (2 * k + 1));
    F128 q_even = f_zero(), q_odd = f_zero(), c_even = f_zero(), c_odd = f_zero(), a0 = f_zero(), b0 = f_zero();
    F128 t_first = f_zero(), t_last = f_zero(), hs = f_zero(), t_rlc = f_zero();
    uint32_t rd0 = 0, rd1 = 0;
    const uint32_t te = lane & 7;
    if (!BASE) {
        q_even = ld128g(pp1 + 16ull * (1 + 2 * k)); q_odd = ld128g(pp1 + 16ull * (2 + 2 * k));
        a0 = ld128g(pp0 + 16ull * (1 + 2 * k)); b0 = ld128g(pp0 + 16ull * (2 + 2 * k));
        c_even = ld128g(rlc + 16ull * (1 + 2 * k)); c_odd = ld128g(rlc + 16ull * (2 + 2 * k));
        // first and last element of a public proof: pp0's in lane te = 2 (its decide), pp1's elsewhere (decide in te = 3, the
        // differences of v[0] and G(r) in te = 0, 1)
        t_first = ld128g(te == 2 ? pp0 : pp1); t_last = ld128g((te == 2 ? pp0 : pp1) + 16ull * (1 + arity));
        // the last step's coefficient of lane te: rlc[0], rlc[1 + arity], rlc[vl .. vl + 4]
        t_rlc = ld128g(rlc + 16ull * (te == 0 ? 0 : (te == 1 ? 1 + arity : vl + (te < 7 ? te - 2 : 0))));
        hs = sc->hash;
        rd0 = sc->rand_diff[0]; rd1 = sc->rand_diff[1];
    }
    const uint8_t *G = d.proof2 + 16ull * arity;
    const uint32_t ge = lane < P.gpoly_len ? lane : 0;
    F128 gcoef = ld128g(G + 16ull * ge);
    const F128 stab = P.s_tab[ge];
    const F128 L0 = pw->L0, h_sumL = pw->h_sumL;
    const F128 rho_a = pw->rho_lo[k & 7], rho_b = pw->rho_hi[k >> 3];
    const F128 r_a = pw->r_lo[lane & 7], r_b = pw->r_hi[(lane >> 3) & 7];
    F128 rho2 = f_zero();
    if (HIST) rho2 = sc->jr[P.jr_len + 1];
    const uint4 *wsrc = reinterpret_cast<const uint4 *>(W.wlimbs + (uint64_t)i * 12 * calls);
    uint4 wq[QTINY_MAX_CALLS][3];
#pragma unroll
    for (uint32_t u = 0; u < QTINY_MAX_CALLS; u++)
#pragma unroll
        for (int q = 0; q < 3; q++) wq[u][q] = wsrc[3 * (u < calls ? u : 0) + q];
    if (!BASE && leader)
        for (uint32_t e = lane; e < P.proof_len; e += 64) {       // proof share of run 1: hashed, never evaluated, but it has to decode
            const F128 v = ld128g(d.proof1 + 16ull * e);
            bad |= !f_in_range(v.lo, v.hi);
        }
    __builtin_amdgcn_sched_barrier(0);

    // ---- the two wire sums of this lane's slot (lazy columns), Histogram's element sum ----
    Lazy cA, cB;
    f_lazy_zero(cA);
    f_lazy_zero(cB);
    F128 sx_p = f_zero();
#pragma unroll
    for (uint32_t u = 0; u < QTINY_MAX_CALLS; u++) {
        if (u < calls) {                        // uniform
            bad |= !f_in_range(x[u].lo, x[u].hi);
            uint32_t xl[5];
            f_split26(x[u], xl);
            const uint32_t wL[5] = {wq[u][0].x, wq[u][0].y, wq[u][0].z, wq[u][0].w, wq[u][1].x};
            const uint32_t wM[5] = {wq[u][1].y, wq[u][1].z, wq[u][1].w, wq[u][2].x, wq[u][2].y};
            f_lazy_mac(cA, xl, wM);
            f_lazy_mac(cB, xl, wL);
            if (HIST) sx_p = f_add(sx_p, x[u]);
        }
    }
    const F128 A = f_lazy_reduce(cA), B = f_lazy_reduce(cB);
    const F128 rk = f_mul(rho_a, rho_b);                                            // rho^(k + 1)
    bad |= act && (!f_in_range(s_even.lo, s_even.hi) | !f_in_range(s_odd.lo, s_odd.hi));
    F128 v_even = f_add(f_mul(L0, s_even), f_mul(rk, A));
    F128 v_odd = f_sub(f_add(f_mul(L0, s_odd), B), h_sumL);
    F128 tag_p = f_zero(), acc0_p = f_zero(), acc1_p = f_zero();
    if (!BASE) {
        bad |= act && (!f_in_range(q_even.lo, q_even.hi) | !f_in_range(q_odd.lo, q_odd.hi) | !f_in_range(a0.lo, a0.hi) | !f_in_range(b0.lo, b0.hi));
        if (leader) { v_even = f_sub(v_even, q_even); v_odd = f_sub(v_odd, q_odd); }
        const F128 t0 = f_add(f_mul(c_even, v_even), f_mul(c_odd, v_odd));
        const F128 t1 = f_mul(a0, b0), t2 = f_mul(q_even, q_odd);                  // decide's sums (flp.rs:483-512)
        if (act) { tag_p = t0; acc0_p = t1; acc1_p = t2; }
    }
    // gadget polynomial: G(r) and <s, G>, one coefficient per lane (gpoly_len <= 15)
    bad |= lane < P.gpoly_len && !f_in_range(gcoef.lo, gcoef.hi);
    if (lane >= P.gpoly_len) gcoef = f_zero();
    const F128 g_p = f_mul(gcoef, f_mul(r_a, r_b)), v0_p = f_mul(gcoef, stab);
    // ---- six sums in one reduce-scatter: the lanes 8 g .. 8 g + 7 end up with sum g ----
    F128 red;
    {
        const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
        const F128 z = f_zero();
        // values 0..7: tag, sx, acc0, acc1 | G(r), v0, -, -
        const F128 k0 = f_add(sel_f(h5, g_p, tag_p), shfl_xor_f(sel_f(h5, tag_p, g_p), 32));
        const F128 k1 = f_add(sel_f(h5, v0_p, sx_p), shfl_xor_f(sel_f(h5, sx_p, v0_p), 32));
        const F128 k2 = f_add(sel_f(h5, z, acc0_p), shfl_xor_f(sel_f(h5, acc0_p, z), 32));
        const F128 k3 = f_add(sel_f(h5, z, acc1_p), shfl_xor_f(sel_f(h5, acc1_p, z), 32));
        const F128 m0 = f_add(sel_f(h4, k2, k0), shfl_xor_f(sel_f(h4, k0, k2), 16));
        const F128 m1 = f_add(sel_f(h4, k3, k1), shfl_xor_f(sel_f(h4, k1, k3), 16));
        red = f_add(sel_f(h3, m1, m0), shfl_xor_f(sel_f(h3, m0, m1), 8));
#pragma unroll
        for (int msk = 4; msk >= 1; msk >>= 1) red = f_add(red, shfl_xor_f(red, msk));
    }
    // group g = 4 h5 + 2 h4 + h3: 0 tag, 1 sx, 2 acc0, 3 acc1, 4 G(r), 5 v0
    const F128 tag_w = readlane_f(red, 0), sx = readlane_f(red, 8), acc0 = readlane_f(red, 16), acc1 = readlane_f(red, 24);
    const F128 g_at_r = readlane_f(red, 32);
    F128 v0 = readlane_f(red, 40);
    if (HIST) v0 = f_add(f_mul(rho2, v0), f_mul(f_sqr(rho2), f_sub(sx, P.half)));   // types.rs:428-436
    if (BASE) {
        bad = wave_or(bad);
        if (bad) {
            for (uint32_t e = lane; e < P.output_len; e += 64) out_row[e] = f_zero();
            for (uint32_t e = lane; e < P.ver_stride; e += 64) ver_row[e] = f_zero();
            if (lane == 0) status[i] = ST_ERR_MODULUS;
            return;
        }
        if (act) { ver_row[1 + 2 * k] = v_even; ver_row[2 + 2 * k] = v_odd; }
        if (lane == 0) { ver_row[0] = v0; ver_row[1 + arity] = g_at_r; if (P.ver_stride > vl) ver_row[vl] = sc->own; }
    } else {
        // decide on the two public proofs (prio3.rs:2762-2786), differences, RLC (:2859-2871): lane te holds one term
        if (te == 2 || te == 3) bad |= !f_in_range(t_first.lo, t_first.hi) | !f_in_range(t_last.lo, t_last.hi);
        const bool ok = f_is_zero(t_first) && f_eq(te == 2 ? acc0 : acc1, t_last);
        const F128 d0 = leader ? f_sub(v0, t_first) : v0, dg = leader ? f_sub(g_at_r, t_last) : g_at_r;
        const F128 mb = te == 0 ? d0 : (te == 1 ? dg : (leader ? f_neg(hs) : hs));   // share of zero: Leader -H, Helper +H (:2844-2855)
        F128 term = f_mul(t_rlc, mb);
        if (te == 2 || te == 3) term = ok ? f_zero() : t_rlc;
        if (te == 4) term = rd0 ? t_rlc : f_zero();
        if (te == 5) term = rd1 ? t_rlc : f_zero();
        if (te == 7) term = tag_w;
#pragma unroll
        for (int msk = 4; msk >= 1; msk >>= 1) term = f_add(term, shfl_xor_f(term, msk));
        bad = wave_or(bad);
        if (lane == 0) {
            if (bad) { status[i] = ST_ERR_MODULUS; term = f_zero(); }
            // prepare_encoded_key: the Helper-role tag is negated (main.rs:53-59)
            st128(tag_out, leader ? term : f_neg(term));
        }
        if (bad) {
            for (uint32_t e = lane; e < P.output_len; e += 64) out_row[e] = f_zero();
            return;
        }
    }
    // ---- out share: truncate (types.rs:641-648) / the elements themselves (:439-442) ----
    if (HIST) {
#pragma unroll
        for (uint32_t u = 0; u < QTINY_MAX_CALLS; u++)
            if (xok[u]) out_row[u * chunk + k] = x[u];
    } else {
        for (uint32_t e0 = 0; e0 < P.output_len; e0 += 64)
            truncate_block(P, d.meas, out_row, e0, e0 + 64 < P.output_len ? e0 + 64 : P.output_len, lane);
    }
}

// ----------------------------------------------------------------------------- aggregate ----
// Aggregatable::merge over the kept reports (vdaf.rs:658-676, main.rs:108-118): column sums
// of out_shares[n][len] mod p.  Pure HBM streaming: 16 B per (report, element).  Each block
// owns a slab of reports; sums are carried in 192 bits and reduced once.
__global__ void __launch_bounds__(256) k_agg_partial(const F128 *rows, const uint8_t *keep, uint64_t n, uint32_t len,
                                                     uint32_t rows_per_block, F128 *partials) {
    uint64_t r0 = (uint64_t)blockIdx.x * rows_per_block;
    uint64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    const uint32_t tpr = len < blockDim.x ? len : blockDim.x;  // threads across one row
    const uint32_t rstep = blockDim.x / tpr;                   // rows per sweep
    const uint32_t sub = threadIdx.x / tpr, col0 = threadIdx.x % tpr;
    __shared__ F128 red[256];
    F128 mine = f_zero();
    if (sub < rstep)
        for (uint32_t e = col0; e < len; e += tpr) {
            uint64_t lo = 0, hi = 0, ov = 0;
            for (uint64_t r = r0 + sub; r < r1; r += rstep) {
                if (keep && !keep[r]) continue;
                F128 v = rows[r * len + e];
                u128 s = (((u128)hi << 64) | lo) + f_u(v);
                ov += s < f_u(v);
                lo = (uint64_t)s; hi = (uint64_t)(s >> 64);
            }
            // fold the overflow count: ov * 2^128
            mine = f_reduce256(lo, hi, ov, 0);
            if (rstep == 1) partials[(uint64_t)blockIdx.x * len + e] = mine;
        }
    if (rstep == 1) return;                     // uniform over the launch
    // short rows (len < 256): rstep rows were swept side by side; one partial row per workgroup all the same, so that the
    // second pass reads as many rows as there are workgroups, not rstep times as many (Average: len = 1, rstep = 256)
    red[threadIdx.x] = mine;
    __syncthreads();
    if (sub == 0) {
        F128 acc = mine;
        for (uint32_t k = 1; k < rstep; k++) acc = f_add(acc, red[k * tpr + col0]);
        partials[(uint64_t)blockIdx.x * len + col0] = acc;
    }
}
// Long rows (a Field128 sketch: 258 x 2^depth elements): thread per column.  The sum of column e over rows [r0, r1) in 192 bits;
// eight 16-byte loads in flight per lane (keep[] is uniform over the lanes: a scalar branch), one carry chain after them.
__device__ __forceinline__ void col_sum192(const F128 *rows, const uint8_t *keep, uint64_t r0, uint64_t r1, uint32_t len, uint32_t e,
                                           uint64_t &lo, uint64_t &hi, uint64_t &ov) {
    auto add = [&](const F128 &v) {
        const u128 s = (((u128)hi << 64) | lo) + f_u(v);
        ov += s < f_u(v);
        lo = (uint64_t)s; hi = (uint64_t)(s >> 64);
    };
    uint64_t r = r0;
    for (; r + 8 <= r1; r += 8) {
        F128 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            v[u] = f_zero();
            if (!keep || keep[r + u]) v[u] = rows[(r + u) * len + e];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) add(v[u]);
    }
    for (; r < r1; r++)
        if (!keep || keep[r]) add(rows[r * len + e]);
}
// grid.y slabs of rows, one partial row per slab
__global__ void __launch_bounds__(256) k_agg_partial_cols(const F128 *rows, const uint8_t *keep, uint64_t n, uint32_t len,
                                                          uint32_t rows_per_slab, F128 *partials) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= len) return;
    const uint64_t r0 = (uint64_t)blockIdx.y * rows_per_slab, r1 = r0 + rows_per_slab < n ? r0 + rows_per_slab : n;
    uint64_t lo = 0, hi = 0, ov = 0;
    col_sum192(rows, keep, r0, r1, len, e, lo, hi, ov);
    partials[(uint64_t)blockIdx.y * len + e] = f_reduce256(lo, hi, ov, 0);
}
// one slab (the columns alone fill the chip): the sum goes straight to `out`, on top of `extra` (may be NULL, may be `out`:
// every lane reads its own element before it writes it) -- no partial rows, no second pass
__global__ void __launch_bounds__(256) k_agg_cols_direct(const F128 *rows, const uint8_t *keep, uint64_t n, uint32_t len,
                                                         const F128 *extra, F128 *out) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= len) return;
    uint64_t lo = 0, hi = 0, ov = 0;
    if (extra) { const F128 x = extra[e]; lo = x.lo; hi = x.hi; }      // any 128-bit value: the reduction below makes it canonical
    col_sum192(rows, keep, 0, n, len, e, lo, hi, ov);
    out[e] = f_reduce256(lo, hi, ov, 0);
}
// second pass over the partial sums: 16 columns per workgroup, 16 lanes down each column; `extra` (may be NULL, may be
// `out`) is one more row: the running sum of an accumulating caller
__global__ void __launch_bounds__(256) k_agg_final(const F128 *partials, uint32_t nparts, uint32_t len, const F128 *extra, F128 *out) {
    __shared__ F128 red[256];
    const uint32_t col = blockIdx.x * 16 + (threadIdx.x & 15), grp = threadIdx.x >> 4;
    uint64_t lo = 0, hi = 0, ov = 0;
    if (col < len)
        for (uint32_t q = grp; q < nparts; q += 16) {
            F128 v = partials[(uint64_t)q * len + col];
            u128 s = (((u128)hi << 64) | lo) + f_u(v);
            ov += s < f_u(v);
            lo = (uint64_t)s; hi = (uint64_t)(s >> 64);
        }
    red[threadIdx.x] = f_reduce256(lo, hi, ov, 0);
    __syncthreads();
    for (uint32_t s = 8; s >= 1; s >>= 1) {
        if (grp < s) red[threadIdx.x] = f_add(red[threadIdx.x], red[threadIdx.x + 16 * s]);
        __syncthreads();
    }
    if (grp == 0 && col < len) {
        F128 v = red[threadIdx.x];
        if (extra) { const F128 x = extra[col]; v = f_add(v, f_reduce256(x.lo, x.hi, 0, 0)); }
        out[col] = v;
    }
}

// Heavy hitters (server-hh/src/main.rs:72-104, hhcore/src/bucket.rs:48-55): per-counter u32
// wrapping sums over the kept clients' bucket vectors.  4 B per counter, HBM streaming.
__global__ void __launch_bounds__(256) k_hh_sum_u32(const uint32_t *buckets, const uint8_t *keep, uint64_t n,
                                                    uint64_t words, uint32_t clients_per_block, uint32_t *acc) {
    uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (w >= words) return;
    uint64_t c0 = (uint64_t)blockIdx.y * clients_per_block;
    uint64_t c1 = c0 + clients_per_block < n ? c0 + clients_per_block : n;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (w + 4 <= words && (words & 3) == 0) {
        // sixteen vectors in flight per lane (8: +1.5 % on the step, 32: +0.5 %): the launch that runs beside the proof kernels holds few wavefronts per CU (its
        // workgroups ask for LDS they do not use, api.hip: hh_sum_on), and they must keep HBM busy on their own
#ifndef WH_HHSUM_INFLIGHT
#define WH_HHSUM_INFLIGHT 16
#endif
        uint64_t c = c0;
        for (; c + WH_HHSUM_INFLIGHT <= c1; c += WH_HHSUM_INFLIGHT) {
            uint4 v[WH_HHSUM_INFLIGHT];
#pragma unroll
            for (int u = 0; u < WH_HHSUM_INFLIGHT; u++) {
                v[u] = make_uint4(0, 0, 0, 0);
                if (!keep || keep[c + u]) v[u] = *reinterpret_cast<const uint4 *>(buckets + (c + u) * words + w);
            }
#pragma unroll
            for (int u = 0; u < WH_HHSUM_INFLIGHT; u++) { a0 += v[u].x; a1 += v[u].y; a2 += v[u].z; a3 += v[u].w; }
        }
        for (; c < c1; c++) {
            if (keep && !keep[c]) continue;
            uint4 v = *reinterpret_cast<const uint4 *>(buckets + c * words + w);
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
        atomicAdd(acc + w, a0); atomicAdd(acc + w + 1, a1); atomicAdd(acc + w + 2, a2); atomicAdd(acc + w + 3, a3);
    } else {
        for (uint64_t q = w; q < words && q < w + 4; q++) {
            uint32_t a = 0;
            for (uint64_t c = c0; c < c1; c++)
                if (!keep || keep[c]) a += buckets[c * words + q];
            atomicAdd(acc + q, a);
        }
    }
}

// ------------------------------------------------------------------- unit-parity kernels ----
__global__ void k_test_f128(const F128 *a, const F128 *b, F128 *out, uint32_t n, int op) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F128 r;
    switch (op) {
    case 0: r = f_add(a[i], b[i]); break;
    case 1: r = f_sub(a[i], b[i]); break;
    case 2: r = f_mul(a[i], b[i]); break;
    case 3: r = f_inv(a[i]); break;
    case 4:
    case 5: {   // the lazy dot product: one term, and LAZY_MAX_TERMS equal terms
        uint32_t x[5], y[5];
        f_split26(a[i], x);
        f_split26(b[i], y);
        Lazy acc;
        f_lazy_zero(acc);
        for (uint32_t t = 0; t < (op == 4 ? 1u : LAZY_MAX_TERMS); t++) f_lazy_mac(acc, x, y);
        r = f_lazy_reduce(acc);
        break;
    }
    default: {  // 6: sum_l 2^l a over l < 32 through the small-multiplier columns
        LazySmall acc;
        f_small_zero(acc);
        for (uint32_t l = 0; l < 32; l++) f_small_mac(acc, a[i], 1u << l);
        r = f_small_reduce(acc);
        break;
    }
    }
    out[i] = r;
}
// Field64 (f64.hpp): op 0 add, 1 sub, 2 mul, 3 inv
__global__ void k_test_f64(const uint64_t *a, const uint64_t *b, uint64_t *out, uint32_t n, int op) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = op == 0 ? g_add(a[i], b[i]) : op == 1 ? g_sub(a[i], b[i]) : op == 2 ? g_mul(a[i], b[i]) : g_inv(a[i]);
}
// XofShake128::seed_stream(seed, dst(alg_id, usage), binder).into_field_vec(n) per lane
__global__ void __launch_bounds__(64) k_test_xof(const uint8_t *seeds, uint32_t alg_id, uint32_t usage, const uint8_t *binder,
                           uint32_t binder_len, uint32_t n_elems, uint8_t *out, uint32_t n, F128 bound) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SpongeG s;
    s.init_xof(alg_id, usage, seeds + 16ull * i);
    s.absorb(binder, binder_len);
    s.finalize();
    xof_emit(s, out + 16ull * n_elems * i, n_elems, bound);
}

}  // namespace wh
