// CPU client: makes valid batched Prio3 reports (the input of the GPU path).
//
// Mirrors what client-batch-prio3 does per client (client-batch-prio3/src/main.rs:67-153):
// Prio3::shard_batched = shard_with_random_new (vdaf/prio3.rs:537-991) -- two FLP proofs over
// fused joint randomness, Fiat-Shamir query randomness, two public proofs -- then
// VdafBatchedKey::get_encoded (vdaf.rs:148-162) for aggregator 0 (Leader) and 1 (Helper).
// Arithmetic is on canonical residues (f128.hpp); nothing here touches oracle/.
#include <cstring>
#include <thread>
#include <vector>
#include "host_math.hpp"
#include "keccak.hpp"

namespace wh {
namespace {

typedef std::vector<F128> Vec;

// sampler bound: p, unless a test lowered it (wh_client_set_test_sampler_bound)
F128 g_sampler_bound = F128{P_LO, P_HI};

Vec expand(uint32_t alg, uint32_t usage, const uint8_t *seed, const uint8_t *binder, uint32_t blen, uint32_t n) {
    Sponge s;
    s.init_xof(alg, usage, seed);
    if (blen) s.absorb(binder, blen);
    s.finalize();
    Vec out(n);
    xof_emit(s, reinterpret_cast<uint8_t *>(out.data()), n, g_sampler_bound);
    return out;
}
F128 derive_seed(uint32_t alg, const std::vector<F128> &parts) {
    Sponge s;
    s.init_xof(alg, DST_JR_SEED, 0ull, 0ull);
    for (const F128 &part : parts) s.absorb_f(part);
    return s.into_seed();
}
void absorb_vec(Sponge &s, const Vec &v) { s.absorb(reinterpret_cast<const uint8_t *>(v.data()), (uint32_t)(16 * v.size())); }

struct Flp {
    wh_prio3_params p;
    HostSizes z;
    // recorded wire values, [arity][1 + calls]; entry 0 is the prover's random seed for the wire
    void wire_values(const Vec &x, const Vec &seeds, const F128 *jr, Vec &fv) const {
        const uint32_t stride = 1 + z.calls;
        fv.assign((size_t)z.arity * stride, f_zero());
        for (uint32_t w = 0; w < z.arity; w++) fv[(size_t)w * stride] = seeds[w];
        const F128 minus_one = f_neg(f_one());  // input - 1/num_shares with num_shares = 1
        if (p.circuit == WH_AVERAGE) {
            for (uint32_t j = 1; j <= z.calls; j++) fv[j] = x[j - 1];
            return;
        }
        F128 rp = jr[0];
        for (uint32_t j = 1; j <= z.calls; j++)
            for (uint32_t k = 0; k < p.chunk; k++) {
                uint32_t i = (j - 1) * p.chunk + k;
                if (i < z.input_len) {
                    fv[(size_t)(2 * k) * stride + j] = f_mul(rp, x[i]);
                    fv[(size_t)(2 * k + 1) * stride + j] = f_sub(x[i], f_one());
                    rp = f_mul(rp, jr[0]);
                } else {
                    fv[(size_t)(2 * k + 1) * stride + j] = minus_one;
                }
            }
    }
    // gadget on plain wire values at call j
    F128 gadget_at(const Vec &fv, uint32_t j) const {
        const uint32_t stride = 1 + z.calls;
        if (p.circuit == WH_AVERAGE) return f_sub(f_sqr(fv[j]), fv[j]);
        F128 acc = f_zero();
        for (uint32_t k = 0; k < p.chunk; k++)
            acc = f_add(acc, f_mul(fv[(size_t)(2 * k) * stride + j], fv[(size_t)(2 * k + 1) * stride + j]));
        return acc;
    }
    // Type::prove (flp.rs:231-347).  Keeps the wire polynomials (coefficients) and the recorded
    // wire values so that the public proof can be evaluated once r is known.
    void prove(const Vec &x, const Vec &prove_rand, const F128 *jr, Vec &proof, Vec &fcoef, Vec &fv) const {
        const uint32_t stride = 1 + z.calls, m = z.m, n2 = 2 * m;
        wire_values(x, prove_rand, jr, fv);
        proof.assign(z.proof_len, f_zero());
        fcoef.assign((size_t)z.arity * m, f_zero());
        Vec acc(n2, f_zero()), a(n2), b(n2), f(m);
        auto wire_poly = [&](uint32_t w, Vec &evals) {
            for (uint32_t j = 0; j < m; j++) f[j] = j < stride ? fv[(size_t)w * stride + j] : f_zero();
            ntt(f, true);                                     // coefficients of the wire polynomial
            evals.assign(n2, f_zero());
            for (uint32_t j = 0; j < m; j++) { evals[j] = f[j]; fcoef[(size_t)w * m + j] = f[j]; }
            ntt(evals, false);
        };
        for (uint32_t w = 0; w < z.arity; w++) proof[w] = prove_rand[w];
        if (p.circuit == WH_AVERAGE) {
            wire_poly(0, a);
            for (uint32_t i = 0; i < n2; i++) acc[i] = f_sub(f_sqr(a[i]), a[i]);
        } else {
            for (uint32_t w = 0; w < z.arity; w += 2) {
                wire_poly(w, a);
                wire_poly(w + 1, b);
                for (uint32_t i = 0; i < n2; i++) acc[i] = f_add(acc[i], f_mul(a[i], b[i]));
            }
        }
        ntt(acc, true);                                       // gadget polynomial coefficients
        for (uint32_t i = 0; i < z.gpoly_len; i++) proof[z.arity + i] = acc[i];
    }
    // Type::query with num_shares = 1 on the true input (prio3.rs:951-969) at randomness r
    int32_t public_proof(const Vec &x, const Vec &proof, const Vec &fcoef, const Vec &fv, const F128 *jr, const F128 &r,
                         Vec &pub) const {
        if (f_eq(f_sqr_n(r, z.log2m), f_one())) return WH_ERR_QUERY_RAND;   // flp.rs:413-419
        pub.assign(z.verifier_len, f_zero());
        for (uint32_t w = 0; w < z.arity; w++) pub[1 + w] = horner(fcoef.data() + (size_t)w * z.m, z.m, r);
        pub[1 + z.arity] = horner(proof.data() + z.arity, z.gpoly_len, r);
        // circuit output: the gadget polynomial at call point j IS the gadget on the recorded values
        F128 v0 = f_zero();
        if (p.circuit == WH_AVERAGE) {
            F128 rp = jr[0];
            for (uint32_t j = 1; j <= z.calls; j++) { v0 = f_add(v0, f_mul(rp, gadget_at(fv, j))); rp = f_mul(rp, jr[0]); }
        } else {
            for (uint32_t j = 1; j <= z.calls; j++) v0 = f_add(v0, gadget_at(fv, j));
            if (p.circuit == WH_HISTOGRAM) {
                F128 sum_check = f_neg(f_one());
                for (uint32_t i = 0; i < z.input_len; i++) sum_check = f_add(sum_check, x[i]);
                v0 = f_add(f_mul(jr[1], v0), f_mul(f_sqr(jr[1]), sum_check));
            }
        }
        pub[0] = v0;
        return WH_OK;
    }
};

int32_t encode_measurement(const wh_prio3_params &p, const HostSizes &z, const uint64_t *meas, Vec &x) {
    x.assign(z.input_len, f_zero());
    switch (p.circuit) {
    case WH_SUMVEC:
        for (uint32_t e = 0; e < p.len; e++) {
            if (p.bits < 64 && (meas[e] >> p.bits)) return WH_ERR_PARAM;
            for (uint32_t l = 0; l < p.bits && l < 64; l++) if ((meas[e] >> l) & 1) x[(size_t)e * p.bits + l] = f_one();
        }
        return WH_OK;
    case WH_HISTOGRAM:
        if (meas[0] >= p.len) return WH_ERR_PARAM;
        x[meas[0]] = f_one();
        return WH_OK;
    default:
        if (p.bits < 64 && (meas[0] >> p.bits)) return WH_ERR_PARAM;
        for (uint32_t l = 0; l < p.bits && l < 64; l++) if ((meas[0] >> l) & 1) x[l] = f_one();
        return WH_OK;
    }
}

uint8_t *put(uint8_t *o, const void *src, size_t n) { memcpy(o, src, n); return o + n; }
uint8_t *put_vec(uint8_t *o, const Vec &v) { return put(o, v.data(), 16 * v.size()); }

}  // namespace
}  // namespace wh

using namespace wh;

extern "C" int32_t wh_client_shard_batched(const wh_prio3_params *params, const uint64_t *measurement,
                                           const uint8_t nonce[16], const uint8_t *rand,
                                           const uint8_t client_id_be[16], uint8_t *leader_report,
                                           uint8_t *helper_report) {
    Flp flp;
    flp.p = *params;
    int32_t rc = host_sizes(flp.p, flp.z);
    if (rc) return rc;
    const HostSizes &z = flp.z;
    const uint32_t alg = z.alg_id;
    // The servers are Alice and Bob; shard_batched is generic in the number of aggregators (its own test runs three,
    // prio3.rs:3077-3085).  random_seeds order: per Helper measurement, proof 1, proof 2, joint-rand blind (prio3.rs:579-581,
    // 589); Leader's joint-rand blind (:647); the two prove seeds (:705, :753); per Helper query-rand blind (:826); Leader's
    // (:883): 16 (5 na - 1) bytes.
    const uint32_t na = flp.p.num_aggregators, nh = na - 1;
    const uint8_t *l_jr_blind = rand + 64 * (size_t)nh, *prove_seed[2] = {l_jr_blind + 16, l_jr_blind + 32},
                  *h_qr_blinds = l_jr_blind + 48, *l_qr_blind = h_qr_blinds + 16 * (size_t)nh;
    auto h_meas_seed = [&](uint32_t a) { return rand + 64 * (size_t)(a - 1); };
    auto h_proof_seed = [&](uint32_t a, int t) { return rand + 64 * (size_t)(a - 1) + 16 + 16 * t; };
    auto h_jr_blind = [&](uint32_t a) { return rand + 64 * (size_t)(a - 1) + 48; };
    Vec x;
    if ((rc = encode_measurement(flp.p, z, measurement, x))) return rc;
    const uint8_t zero = 0;
    std::vector<F128> jr_parts(na), qr_parts(na);
    Sponge s;
    Vec l_meas = x;
    for (uint32_t a = 1; a < na; a++) {
        const uint8_t id = (uint8_t)a;
        Vec h_meas = expand(alg, DST_MEAS, h_meas_seed(a), &id, 1, z.input_len);
        for (uint32_t i = 0; i < z.input_len; i++) l_meas[i] = f_sub(l_meas[i], h_meas[i]);
        s.init_xof(alg, DST_JR_PART, h_jr_blind(a)); s.absorb(&id, 1); s.absorb(nonce, 16); absorb_vec(s, h_meas);
        jr_parts[a] = s.into_seed();
    }
    s.init_xof(alg, DST_JR_PART, l_jr_blind); s.absorb(&zero, 1); s.absorb(nonce, 16); absorb_vec(s, l_meas);
    jr_parts[0] = s.into_seed();
    F128 seed = derive_seed(alg, jr_parts);
    Vec jr = expand(alg, DST_JOINT_RAND, reinterpret_cast<const uint8_t *>(&seed), nullptr, 0, 2 * z.jr_len);

    Vec proof[2], l_proof[2], pub[2], fcoef[2], fv[2];
    for (int t = 0; t < 2; t++) {                                            // prio3.rs:703-782
        Vec prove_rand = expand(alg, DST_PROVE_RAND, prove_seed[t], nullptr, 0, z.arity);
        flp.prove(x, prove_rand, jr.data() + t * z.jr_len, proof[t], fcoef[t], fv[t]);
        l_proof[t] = proof[t];
        for (uint32_t a = 1; a < na; a++) {
            const uint8_t id = (uint8_t)a;
            const Vec h = expand(alg, DST_PROOF, h_proof_seed(a, t), &id, 1, z.proof_len);
            for (uint32_t i = 0; i < z.proof_len; i++) l_proof[t][i] = f_sub(l_proof[t][i], h[i]);
        }
    }
    for (uint32_t a = 1; a < na; a++) {                                      // prio3.rs:806-874
        const uint8_t id = (uint8_t)a;
        s.init_xof(alg, DST_QR_PART, h_qr_blinds + 16 * (size_t)(a - 1)); absorb_vec(s, jr);
        for (int t = 0; t < 2; t++) absorb_vec(s, expand(alg, DST_PROOF, h_proof_seed(a, t), &id, 1, z.proof_len));
        qr_parts[a] = s.into_seed();
    }
    s.init_xof(alg, DST_QR_PART, l_qr_blind); absorb_vec(s, jr); absorb_vec(s, l_proof[0]); absorb_vec(s, l_proof[1]);
    qr_parts[0] = s.into_seed();
    seed = derive_seed(alg, qr_parts);  // DST_JR_SEED again, prio3.rs:933
    Vec qr = expand(alg, DST_QUERY_RAND, reinterpret_cast<const uint8_t *>(&seed), nullptr, 0, 2);
    for (int t = 0; t < 2; t++)
        if ((rc = flp.public_proof(x, proof[t], fcoef[t], fv[t], jr.data() + t * z.jr_len, qr[t], pub[t]))) return rc;

    for (uint32_t a = 0; a < na; a++) {
        uint8_t *o = a ? helper_report + (size_t)(a - 1) * z.helper_len : leader_report;
        o = put(o, client_id_be, 16);
        o = put(o, jr_parts.data(), 16 * (size_t)na);
        *o++ = (uint8_t)a;
        if (a == 0) {
            o = put_vec(o, l_meas); o = put_vec(o, l_proof[0]); o = put(o, l_jr_blind, 16); o = put_vec(o, l_proof[1]);
        } else {
            o = put(o, h_meas_seed(a), 16); o = put(o, h_proof_seed(a, 0), 16); o = put(o, h_jr_blind(a), 16); o = put(o, h_proof_seed(a, 1), 16);
        }
        o = put(o, qr_parts.data(), 16 * (size_t)na);
        uint8_t nq[8];
        for (int i = 0; i < 8; i++) nq[i] = (uint8_t)((uint64_t)z.verifier_len >> (56 - 8 * i));
        o = put(o, nq, 8);
        o = put_vec(o, pub[0]); o = put_vec(o, pub[1]);
        o = put(o, a ? h_qr_blinds + 16 * (size_t)(a - 1) : l_qr_blind, 16);
        o = put(o, nonce, 16);
    }
    return WH_OK;
}

extern "C" int32_t wh_client_shard_many(const wh_prio3_params *params, const uint64_t *measurements, uint32_t meas_words,
                                        const uint8_t *nonces, const uint8_t *rands, uint64_t first_client_id, uint64_t n,
                                        uint32_t nthreads, uint8_t *leader_reports, uint8_t *helper_reports) {
    HostSizes z;
    int32_t rc = host_sizes(*params, z);
    if (rc) return rc;
    const uint64_t nh = params->num_aggregators - 1, rand_len = 16 * (5ull * params->num_aggregators - 1);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    std::vector<int32_t> rcs(nthreads, 0);
    auto work = [&](uint32_t t) {
        for (uint64_t i = t; i < n; i += nthreads) {
            uint8_t id[16] = {0};
            uint64_t cid = first_client_id + i;
            for (int b = 0; b < 8; b++) id[15 - b] = (uint8_t)(cid >> (8 * b));
            int32_t r = wh_client_shard_batched(params, measurements + i * meas_words, nonces + 16 * i, rands + rand_len * i, id,
                                                leader_reports + i * (uint64_t)z.leader_len,
                                                helper_reports + i * nh * (uint64_t)z.helper_len);
            if (r) rcs[t] = r;
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    for (int32_t r : rcs) if (r) return r;
    return WH_OK;
}

extern "C" void wh_client_set_test_sampler_bound(const uint8_t bound[16]) {
    using namespace wh;
    F128 b = F128{P_LO, P_HI};
    if (bound) {
        F128 v;
        memcpy(&v, bound, 16);
        if (f_in_range(v.lo, v.hi)) b = v;
    }
    g_sampler_bound = b;
}

// ---- flp::Type surface on the CPU (the prover is client-side work) ----
extern "C" int32_t wh_flp_encode(const wh_prio3_params *params, const uint64_t *measurement, uint8_t *input) {
    HostSizes z;
    if (!params || !measurement || !input) return WH_ERR_PARAM;
    int32_t rc = host_sizes(*params, z);
    if (rc) return rc;
    Vec x;
    if ((rc = encode_measurement(*params, z, measurement, x))) return rc;
    memcpy(input, x.data(), 16 * x.size());
    return WH_OK;
}
extern "C" int32_t wh_flp_prove(const wh_prio3_params *params, const uint8_t *input, const uint8_t *prove_rand,
                                const uint8_t *joint_rand, uint8_t *proof) {
    Flp flp;
    if (!params || !input || !prove_rand || !joint_rand || !proof) return WH_ERR_PARAM;
    flp.p = *params;
    int32_t rc = host_sizes(flp.p, flp.z);
    if (rc) return rc;
    const HostSizes &z = flp.z;
    Vec x(z.input_len), pr(z.arity), jr(z.jr_len), out, fcoef, fv;
    memcpy(x.data(), input, 16 * x.size());
    memcpy(pr.data(), prove_rand, 16 * pr.size());
    memcpy(jr.data(), joint_rand, 16 * jr.size());
    for (const Vec *v : {&x, &pr, &jr})
        for (const F128 &e : *v) if (!f_in_range(e.lo, e.hi)) return WH_ERR_MODULUS;
    flp.prove(x, pr, jr.data(), out, fcoef, fv);
    memcpy(proof, out.data(), 16 * out.size());
    return WH_OK;
}
