/*
 * ORACLE (test infrastructure).  Fully linear proof system: prove / query / decide /
 * truncate and the SumVec, Histogram and Average validity circuits, restated from
 *   batched-flpcp-rs/src/flp.rs:231-347 (prove), 359-479 (query), 483-512 (decide),
 *                               596-745 (Prove/QueryShimGadget), 749-758 (lengths)
 *   batched-flpcp-rs/src/flp/gadgets.rs:76-79 (Mul::call), 185-188 (PolyEval::call),
 *                               250-257 (ParallelSum::call), 47-72,138-181,259-276 (call_poly)
 *   batched-flpcp-rs/src/flp/types.rs:241-313 (Average), 378-471 (Histogram),
 *                               572-677 (SumVec), 686-698, 742-777 (range checks)
 *   batched-flpcp-rs/src/field.rs:211-269 (bit-vector encode / decode)
 * `query` keeps the reference's shape on purpose (one inverse FFT + Horner per wire,
 * one inv() per call): this file is also what bench.py times as the CPU baseline.
 */
#include "oracle_internal.h"
#include <string.h>

uint32_t wo_alg_id(const wo_params *p) {
    if (p->id) return p->id;
    switch (p->circuit) {
    case WO_SUMVEC: return 2;          /* types.rs:577 */
    case WO_HISTOGRAM: return 3;       /* types.rs:383 */
    default: return 0xFFFF0000u;       /* types.rs:242 */
    }
}

static uint32_t next_pow2(uint32_t x) { uint32_t r = 1; while (r < x) r <<= 1; return r; }

int wo_sizes_of(const wo_params *p, wo_sizes *s) {
    wo_init();
    memset(s, 0, sizeof *s);
    if (p->num_aggregators < 2 || p->num_aggregators > 254) return WO_ERR_PARAM;
    switch (p->circuit) {
    case WO_SUMVEC:
        if (!p->bits || !p->len || !p->chunk || p->bits > 127) return WO_ERR_PARAM; /* types.rs:515-535 */
        s->input_len = p->bits * p->len;
        s->calls = (s->input_len + p->chunk - 1) / p->chunk; /* types.rs:541-544 */
        s->arity = 2 * p->chunk;
        s->output_len = p->len;
        s->joint_rand_len = 1;
        break;
    case WO_HISTOGRAM:
        if (!p->len || !p->chunk) return WO_ERR_PARAM;       /* types.rs:337-351 */
        s->input_len = p->len;
        s->calls = (p->len + p->chunk - 1) / p->chunk;       /* types.rs:353-356 */
        s->arity = 2 * p->chunk;
        s->output_len = p->len;
        s->joint_rand_len = 2;
        break;
    case WO_AVERAGE:
        if (!p->bits || p->bits > 127) return WO_ERR_PARAM;  /* valid_integer_bitlength */
        s->input_len = p->bits;
        s->calls = p->bits;
        s->arity = 1;
        s->output_len = 1;
        s->joint_rand_len = 1;
        break;
    default: return WO_ERR_PARAM;
    }
    s->m = next_pow2(1 + s->calls);                 /* flp.rs:749-751 */
    s->proof_len = s->arity + 2 * (s->m - 1) + 1;   /* types.rs:654-656, 448-450, 290-292 */
    s->verifier_len = 2 + s->arity;                 /* types.rs:658-660, 294-296 */
    s->prove_rand_len = s->arity;
    s->query_rand_len = 1;
    /* VdafBatchedKey::encode, vdaf.rs:148-162, 2 aggregators, joint rand present */
    uint32_t common = 16 + 16 * p->num_aggregators + 1 + 16 * p->num_aggregators + 8 + 2 * 16 * s->verifier_len + 16 + 16;
    s->leader_report_len = common + 16 * (s->input_len + s->proof_len) + 16 + 16 * s->proof_len;
    s->helper_report_len = common + 16 + 16 + 16 + 16;
    return 0;
}

/* ------------------------------------------------------------- shim gadget ---- */
typedef struct {
    const wo_sizes *sz;
    uint32_t circuit;
    int query;       /* 0: ProveShimGadget, 1: QueryShimGadget */
    u128 *f_vals;    /* [arity][1+calls] */
    u128 *p_vals;    /* query only: gadget polynomial at the 2m-th roots of unity */
    uint32_t step, ct;
} shim_t;

/* Mul / ParallelSum / PolyEval ::call on plain values */
static u128 inner_call(uint32_t circuit, const wo_sizes *sz, const u128 *inp) {
    if (circuit == WO_AVERAGE) {
        /* PolyEval::call with range_checker = poly_range_check(0,2) = [0,-1,1]: Horner */
        u128 poly[3] = {0, f_neg(f_from_int(1)), f_from_int(1)};
        return poly_eval(poly, 3, inp[0]);
    }
    u128 out = 0; /* ParallelSum::call over Mul::call, gadgets.rs:250-257 */
    for (uint32_t k = 0; k < sz->arity; k += 2) out = f_add(out, f_mul(inp[k], inp[k + 1]));
    return out;
}

static u128 shim_call(shim_t *g, const u128 *inp) {
    uint32_t stride = 1 + g->sz->calls;
    for (uint32_t w = 0; w < g->sz->arity; w++) g->f_vals[w * stride + g->ct] = inp[w];
    u128 out = g->query ? g->p_vals[g->ct * g->step] : inner_call(g->circuit, g->sz, inp);
    g->ct++;
    return out;
}

/* types.rs:742-777 */
static u128 parallel_sum_range_checks(shim_t *g, const u128 *input, uint32_t input_len, u128 jr,
                                      uint32_t chunk, uint32_t num_shares) {
    u128 shares_inv = f_inv(f_from_int(num_shares));
    u128 output = 0, r_power = jr;
    u128 *padded = wo_alloc(2 * chunk * sizeof(u128));
    for (uint32_t base = 0; base < input_len; base += chunk) {
        uint32_t n = input_len - base < chunk ? input_len - base : chunk;
        for (uint32_t i = 0; i < n; i++) {
            padded[2 * i] = f_mul(r_power, input[base + i]);
            padded[2 * i + 1] = f_sub(input[base + i], shares_inv);
            r_power = f_mul(r_power, jr);
        }
        for (uint32_t i = n; i < chunk; i++) {
            padded[2 * i] = 0;
            padded[2 * i + 1] = f_neg(shares_inv);
        }
        output = f_add(output, shim_call(g, padded));
    }
    wo_free(padded);
    return output;
}

static u128 circuit_valid(const wo_params *p, const wo_sizes *sz, shim_t *g, const u128 *input,
                          const u128 *jr, uint32_t num_shares) {
    switch (p->circuit) {
    case WO_SUMVEC: /* types.rs:623-639 */
        return parallel_sum_range_checks(g, input, sz->input_len, jr[0], p->chunk, num_shares);
    case WO_HISTOGRAM: { /* types.rs:410-437 */
        u128 range_check = parallel_sum_range_checks(g, input, sz->input_len, jr[0], p->chunk, num_shares);
        u128 sum_check = f_neg(f_mul(f_from_int(1), f_inv(f_from_int(num_shares))));
        for (uint32_t i = 0; i < sz->input_len; i++) sum_check = f_add(sum_check, input[i]);
        return f_add(f_mul(jr[1], range_check), f_mul(f_mul(jr[1], jr[1]), sum_check));
    }
    default: { /* Average: call_gadget_on_vec_entries, types.rs:686-698 */
        u128 range_check = 0, r = jr[0];
        for (uint32_t i = 0; i < sz->input_len; i++) {
            range_check = f_add(range_check, f_mul(r, shim_call(g, &input[i])));
            r = f_mul(r, jr[0]);
        }
        return range_check;
    }
    }
}

/* ------------------------------------------------------------------- prove ---- */
int flp_prove(const wo_params *p, const u128 *input, const u128 *prove_rand, const u128 *joint_rand, u128 *proof) {
    wo_sizes sz;
    int rc = wo_sizes_of(p, &sz);
    if (rc) return rc;
    uint32_t stride = 1 + sz.calls, m = sz.m, n = 2 * m;
    shim_t g = {&sz, p->circuit, 0, wo_alloc((size_t)sz.arity * stride * sizeof(u128)), NULL, 0, 1};
    for (uint32_t w = 0; w < sz.arity; w++) g.f_vals[w * stride] = prove_rand[w]; /* flp.rs:607-621 */
    (void)circuit_valid(p, &sz, &g, input, joint_rand, 1);                         /* flp.rs:307 */

    /* wire polynomials f[w] = INTT_m(f_vals[w]) (flp.rs:319-333); gadget polynomial
     * G = sum_k f[2k]*f[2k+1] (ParallelSum<Mul>::call_poly) or f^2 - f (PolyEval with
     * x^2 - x).  The products are taken in the size-2m evaluation domain and brought back
     * with one inverse transform; over an exact field this equals the reference's
     * direct / FFT poly_mul paths (gadgets.rs:47-72,138-181) coefficient for coefficient. */
    u128 m_inv = f_inv(f_from_int(m)), n_inv = f_inv(f_from_int(n));
    u128 *f = wo_alloc(m * sizeof(u128)), *ea = wo_alloc(n * sizeof(u128)), *eb = wo_alloc(n * sizeof(u128));
    u128 *acc = wo_alloc(n * sizeof(u128)), *coef = wo_alloc(n * sizeof(u128));
    for (uint32_t w = 0; w < sz.arity; w++) proof[w] = g.f_vals[w * stride];
    if (p->circuit == WO_AVERAGE) {
        fft(f, g.f_vals, stride, m); fft_inv_finish(f, m, m_inv);
        fft(ea, f, m, n);
        for (uint32_t i = 0; i < n; i++) acc[i] = f_sub(f_mul(ea[i], ea[i]), ea[i]);
    } else {
        for (uint32_t w = 0; w < sz.arity; w += 2) {
            fft(f, g.f_vals + (size_t)w * stride, stride, m); fft_inv_finish(f, m, m_inv);
            fft(ea, f, m, n);
            fft(f, g.f_vals + (size_t)(w + 1) * stride, stride, m); fft_inv_finish(f, m, m_inv);
            fft(eb, f, m, n);
            for (uint32_t i = 0; i < n; i++) acc[i] = f_add(acc[i], f_mul(ea[i], eb[i]));
        }
    }
    fft(coef, acc, n, n); fft_inv_finish(coef, n, n_inv);
    for (uint32_t i = 0; i < 2 * (m - 1) + 1; i++) proof[sz.arity + i] = coef[i]; /* flp.rs:336-345 */
    wo_free(f); wo_free(ea); wo_free(eb); wo_free(acc); wo_free(coef); wo_free(g.f_vals);
    return 0;
}

/* ------------------------------------------------------------------- query ---- */
int flp_query(const wo_params *p, const u128 *input, const u128 *proof, const u128 *query_rand,
              const u128 *joint_rand, uint32_t num_shares, u128 *verifier) {
    wo_sizes sz;
    int rc = wo_sizes_of(p, &sz);
    if (rc) return rc;
    uint32_t stride = 1 + sz.calls, m = sz.m;
    u128 r = query_rand[0];
    /* flp.rs:413-419: r^m == 1 -> FlpError::Query */
    if (f_pow(r, m) == f_from_int(1)) return WO_ERR_QUERY_RAND;

    /* QueryShimGadget::new, flp.rs:678-712 */
    uint32_t size = 2 * m; /* (m*degree).next_power_of_two() */
    shim_t g = {&sz, p->circuit, 1, wo_alloc((size_t)sz.arity * stride * sizeof(u128)),
                wo_alloc(size * sizeof(u128)), 2, 1};
    const u128 *gpoly = proof + sz.arity;
    uint32_t gpoly_len = 2 * (m - 1) + 1;
    for (uint32_t w = 0; w < sz.arity; w++) g.f_vals[w * stride] = proof[w];
    fft(g.p_vals, gpoly, gpoly_len, size);
    u128 p_at_r = poly_eval(gpoly, gpoly_len, r);

    verifier[0] = circuit_valid(p, &sz, &g, input, joint_rand, num_shares); /* flp.rs:448-449 */

    /* flp.rs:452-475: per wire, inverse FFT then Horner at r */
    u128 m_inv = f_inv(f_from_int(m));
    u128 *f = wo_alloc(m * sizeof(u128));
    for (uint32_t w = 0; w < sz.arity; w++) {
        fft(f, g.f_vals + (size_t)w * stride, stride, m);
        fft_inv_finish(f, m, m_inv);
        verifier[1 + w] = poly_eval(f, m, r);
    }
    verifier[1 + sz.arity] = p_at_r;
    wo_free(f); wo_free(g.f_vals); wo_free(g.p_vals);
    return 0;
}

/* flp.rs:483-512 */
int flp_decide(const wo_params *p, const u128 *verifier) {
    wo_sizes sz;
    if (wo_sizes_of(p, &sz)) return 0;
    if (verifier[0] != 0) return 0;
    u128 e = inner_call(p->circuit, &sz, verifier + 1);
    return e == verifier[1 + sz.arity];
}

/* field.rs:256-269 */
static u128 decode_bits(const u128 *bits, uint32_t n) {
    u128 decoded = 0;
    for (uint32_t l = 0; l < n; l++) decoded = f_add(decoded, f_mul(f_from_int((u128)1 << l), bits[l]));
    return decoded;
}
int flp_truncate(const wo_params *p, const u128 *input, u128 *out) {
    switch (p->circuit) {
    case WO_SUMVEC: /* types.rs:641-648 */
        for (uint32_t e = 0; e < p->len; e++) out[e] = decode_bits(input + (size_t)e * p->bits, p->bits);
        return 0;
    case WO_HISTOGRAM: /* types.rs:439-442 */
        memcpy(out, input, (size_t)p->len * sizeof(u128));
        return 0;
    default: /* types.rs:280-284 */
        out[0] = decode_bits(input, p->bits);
        return 0;
    }
}
/* field.rs:211-232, types.rs:582-606, 388-393, 247-250 */
int flp_encode(const wo_params *p, const uint64_t *meas, u128 *input) {
    u128 one = f_from_int(1);
    switch (p->circuit) {
    case WO_SUMVEC:
        for (uint32_t e = 0; e < p->len; e++) {
            if (p->bits < 64 && (meas[e] >> p->bits)) return WO_ERR_PARAM;
            for (uint32_t l = 0; l < p->bits; l++) input[(size_t)e * p->bits + l] = (l < 64 && ((meas[e] >> l) & 1)) ? one : 0;
        }
        return 0;
    case WO_HISTOGRAM:
        if (meas[0] >= p->len) return WO_ERR_PARAM;
        for (uint32_t i = 0; i < p->len; i++) input[i] = 0;
        input[meas[0]] = one;
        return 0;
    default:
        if (p->bits < 64 && (meas[0] >> p->bits)) return WO_ERR_PARAM;
        for (uint32_t l = 0; l < p->bits; l++) input[l] = (l < 64 && ((meas[0] >> l) & 1)) ? one : 0;
        return 0;
    }
}

/* --------------------------------------------------------------- C wrappers ---- */
static u128 *ld_vec(const uint8_t *b, size_t n) {
    u128 *v = wo_alloc(n * sizeof(u128));
    for (size_t i = 0; i < n; i++) {
        u128 x = 0;
        for (int k = 0; k < 16; k++) x |= (u128)b[16 * i + k] << (8 * k);
        v[i] = f_from_int(x % WO_P);
    }
    return v;
}
static void st_vec(const u128 *v, size_t n, uint8_t *b) { for (size_t i = 0; i < n; i++) f_encode(v[i], b + 16 * i); }

int wo_flp_prove(const wo_params *p, const uint8_t *input, const uint8_t *prove_rand, const uint8_t *joint_rand, uint8_t *proof) {
    wo_sizes sz; int rc = wo_sizes_of(p, &sz); if (rc) return rc;
    u128 *in = ld_vec(input, sz.input_len), *pr = ld_vec(prove_rand, sz.prove_rand_len), *jr = ld_vec(joint_rand, sz.joint_rand_len);
    u128 *pf = wo_alloc(sz.proof_len * sizeof(u128));
    rc = flp_prove(p, in, pr, jr, pf);
    if (!rc) st_vec(pf, sz.proof_len, proof);
    wo_free(in); wo_free(pr); wo_free(jr); wo_free(pf);
    return rc;
}
int wo_flp_query(const wo_params *p, const uint8_t *input, const uint8_t *proof, const uint8_t *query_rand,
                 const uint8_t *joint_rand, uint32_t num_shares, uint8_t *verifier) {
    wo_sizes sz; int rc = wo_sizes_of(p, &sz); if (rc) return rc;
    u128 *in = ld_vec(input, sz.input_len), *pf = ld_vec(proof, sz.proof_len), *qr = ld_vec(query_rand, 1), *jr = ld_vec(joint_rand, sz.joint_rand_len);
    u128 *v = wo_alloc(sz.verifier_len * sizeof(u128));
    rc = flp_query(p, in, pf, qr, jr, num_shares, v);
    if (!rc) st_vec(v, sz.verifier_len, verifier);
    wo_free(in); wo_free(pf); wo_free(qr); wo_free(jr); wo_free(v);
    return rc;
}
int wo_flp_decide(const wo_params *p, const uint8_t *verifier) {
    wo_sizes sz; if (wo_sizes_of(p, &sz)) return 0;
    u128 *v = ld_vec(verifier, sz.verifier_len);
    int ok = flp_decide(p, v);
    wo_free(v);
    return ok;
}
int wo_flp_truncate(const wo_params *p, const uint8_t *input, uint8_t *out) {
    wo_sizes sz; int rc = wo_sizes_of(p, &sz); if (rc) return rc;
    u128 *in = ld_vec(input, sz.input_len), *o = wo_alloc(sz.output_len * sizeof(u128));
    rc = flp_truncate(p, in, o);
    if (!rc) st_vec(o, sz.output_len, out);
    wo_free(in); wo_free(o);
    return rc;
}
int wo_flp_encode(const wo_params *p, const uint64_t *meas, uint8_t *input) {
    wo_sizes sz; int rc = wo_sizes_of(p, &sz); if (rc) return rc;
    u128 *in = wo_alloc(sz.input_len * sizeof(u128));
    rc = flp_encode(p, meas, in);
    if (!rc) st_vec(in, sz.input_len, input);
    wo_free(in);
    return rc;
}
