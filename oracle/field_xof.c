/*
 * ORACLE (test infrastructure).  Field128 Montgomery arithmetic, SHAKE128 XOF,
 * rejection sampler, iterative FFT -- restated from the reference:
 *   batched-flpcp-rs/src/fp.rs:34-273,391-421   (FieldParameters, FP128)
 *   batched-flpcp-rs/src/field.rs:405-423,598-607 (LE codec, < p check)
 *   batched-flpcp-rs/src/vdaf/xof.rs:127-139,197-218 (XofShake128 framing)
 *   batched-flpcp-rs/src/prng.rs:56-99           (rejection sampling)
 *   batched-flpcp-rs/src/fft.rs:31-124           (DFT, inv_finish, bitrev)
 *   batched-flpcp-rs/src/polynomial.rs:171-183   (Horner)
 * SHAKE128 itself lives in the un-vendored `sha3 = "0.10.8"` crate; FIPS 202 is
 * restated here and pinned by vdaf/test_vec/07/XofShake128.json.
 */
#include "oracle_internal.h"
#include <string.h>

/* ------------------------------------------------------------------ field ---- */
/* p = 2^128 - 7*2^66 + 1 (fp.rs:392). */
const u128 WO_P = ((u128)0xffffffffffffffe4ULL << 64) | 1ULL;
static const uint64_t MU = 0xffffffffffffffffULL; /* -p^-1 mod 2^64 (fp.rs:393) */
static u128 R2;       /* (2^128)^2 mod p, Montgomery form of 2^128 */
static u128 ROOTS[67]; /* ROOTS[l] = 2^l-th principal root, Montgomery form */
static int inited;

/* fp.rs:37-53 */
u128 f_add(u128 x, u128 y) {
    u128 z = x + y;
    int carry = z < x;
    u128 s = z - WO_P;
    int b0 = z < WO_P;
    /* b1 = borrow of (carry - b0) */
    return (carry < b0) ? z : s;
}
/* fp.rs:57-72 */
u128 f_sub(u128 x, u128 y) {
    u128 z = x - y;
    return (x < y) ? z + WO_P : z;
}
u128 f_neg(u128 x) { return f_sub(0, x); }

/* fp.rs:84-213: schoolbook 2x2 limbs, two REDC rounds with w = mu*z_i, final
 * conditional subtraction.  Written over a 5-limb accumulator. */
u128 f_mul(u128 x, u128 y) {
    uint64_t xl[2] = {(uint64_t)x, (uint64_t)(x >> 64)};
    uint64_t yl[2] = {(uint64_t)y, (uint64_t)(y >> 64)};
    uint64_t pl[2] = {(uint64_t)WO_P, (uint64_t)(WO_P >> 64)};
    uint64_t z[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 2; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 2; j++) {
            u128 t = (u128)xl[i] * yl[j] + z[i + j] + c;
            z[i + j] = (uint64_t)t;
            c = (uint64_t)(t >> 64);
        }
        z[i + 2] = c;
    }
    for (int i = 0; i < 2; i++) {
        uint64_t w = MU * z[i];
        uint64_t c = 0;
        for (int j = 0; j < 2; j++) {
            u128 t = (u128)w * pl[j] + z[i + j] + c;
            z[i + j] = (uint64_t)t;
            c = (uint64_t)(t >> 64);
        }
        for (int k = i + 2; k < 5 && c; k++) {
            u128 t = (u128)z[k] + c;
            z[k] = (uint64_t)t;
            c = (uint64_t)(t >> 64);
        }
    }
    u128 prod = ((u128)z[3] << 64) | z[2];
    if (z[4] || prod >= WO_P) prod -= WO_P;
    return prod;
}
/* fp.rs:217-226 */
u128 f_pow(u128 x, u128 e) {
    u128 t = f_from_int(1);
    int top = -1;
    for (int i = 127; i >= 0; i--)
        if ((e >> i) & 1) { top = i; break; }
    for (int i = top; i >= 0; i--) {
        t = f_mul(t, t);
        if ((e >> i) & 1) t = f_mul(t, x);
    }
    return t;
}
u128 f_inv(u128 x) { return f_pow(x, WO_P - 2); } /* fp.rs:231-233 */
static u128 modp(u128 x) { return x >= WO_P ? x - WO_P : x; }
u128 f_from_int(u128 v) { wo_init(); return modp(f_mul(v, R2)); } /* montgomery(), fp.rs:251-253 */
u128 f_to_int(u128 m) { return modp(f_mul(m, 1)); }               /* residue(), fp.rs:271-273 */

u128 f_root(unsigned l) { wo_init(); return ROOTS[l]; }

void wo_init(void) {
    if (inited) return;
    /* r = 2^128 mod p, r2 = r^2 mod p by doubling (no Montgomery needed yet) */
    u128 t = 1;
    for (int i = 0; i < 256; i++) t = f_add(t, t);
    R2 = t;
    inited = 1;
    /* generator of the 2^66 subgroup: 7^((p-1)/2^66)  (draft-irtf-cfrg-vdaf-07 table 3;
     * equals residue(FP128.g), fp.rs:395) */
    u128 g = f_pow(f_from_int(7), (WO_P - 1) >> 66);
    ROOTS[66] = g;
    for (int i = 65; i >= 0; i--) ROOTS[i] = f_mul(ROOTS[i + 1], ROOTS[i + 1]); /* fp.rs:311-315 */
}

/* field.rs:405-423 try_from_bytes (mask is all-ones for Field128, fp.rs:397) */
int f_decode(const uint8_t b[16], u128 *out) {
    u128 v = 0;
    for (int i = 0; i < 16; i++) v |= (u128)b[i] << (8 * i);
    if (v >= WO_P) return WO_ERR_MODULUS;
    *out = f_from_int(v);
    return 0;
}
/* field.rs:598-607 */
void f_encode(u128 m, uint8_t b[16]) {
    u128 v = f_to_int(m);
    for (int i = 0; i < 16; i++) b[i] = (uint8_t)(v >> (8 * i));
}

/* ------------------------------------------------------------- keccak/shake ---- */
static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
static inline uint64_t rol(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

static void keccak_f(uint64_t a[25]) {
    for (int r = 0; r < 24; r++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], RHO[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[r];
    }
}

#define RATE 168
void shake_init(shake_t *s) { memset(s, 0, sizeof *s); }
void shake_absorb(shake_t *s, const uint8_t *in, size_t n) {
    for (size_t i = 0; i < n; i++) {
        s->st[s->pos >> 3] ^= (uint64_t)in[i] << (8 * (s->pos & 7));
        if (++s->pos == RATE) { keccak_f(s->st); s->pos = 0; }
    }
}
void shake_finalize(shake_t *s) {
    s->st[s->pos >> 3] ^= (uint64_t)0x1f << (8 * (s->pos & 7));
    s->st[(RATE - 1) >> 3] ^= (uint64_t)0x80 << (8 * ((RATE - 1) & 7));
    keccak_f(s->st);
    s->pos = 0;
}
void shake_squeeze(shake_t *s, uint8_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (s->pos == RATE) { keccak_f(s->st); s->pos = 0; }
        out[i] = (uint8_t)(s->st[s->pos >> 3] >> (8 * (s->pos & 7)));
        s->pos++;
    }
}
void wo_shake128(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
    shake_t s; shake_init(&s); shake_absorb(&s, in, in_len); shake_finalize(&s); shake_squeeze(&s, out, out_len);
}

/* xof.rs:200-209: absorb [len(dst)] || dst || seed */
void xof_init(shake_t *s, const uint8_t seed[16], const uint8_t *dst, size_t dst_len) {
    uint8_t l = (uint8_t)dst_len;
    shake_init(s);
    shake_absorb(s, &l, 1);
    shake_absorb(s, dst, dst_len);
    shake_absorb(s, seed, 16);
}
/* xof.rs:127-132 */
void xof_into_seed(shake_t *s, uint8_t out[16]) { shake_finalize(s); shake_squeeze(s, out, 16); }
/* prng.rs:73-99 + field.rs from_random_rejection: successive 16-byte chunks of the
 * stream, little-endian, chunks >= p are skipped.  (The 32-element buffer of the
 * reference is a multiple of 16 bytes, so chunk boundaries are those of the stream.) */
/* Test hook (not in the reference): chunks >= wo_sampler_bound are skipped, default p.  With a
 * bound below p the skip branch, which SHAKE output reaches with probability 2^-59 per element,
 * runs for every XOF in the protocol; accepted values are still canonical field elements. */
static u128 wo_sampler_bound = WO_P;
void wo_set_test_sampler_bound(const uint8_t bound[16]) {
    u128 v = 0;
    if (bound) for (int i = 0; i < 16; i++) v |= (u128)bound[i] << (8 * i);
    wo_sampler_bound = (bound && v <= WO_P) ? v : WO_P;
}
void xof_next_vec(shake_t *s, u128 *out, size_t n) {
    size_t got = 0;
    while (got < n) {
        uint8_t b[16];
        shake_squeeze(s, b, 16);
        u128 v = 0;
        for (int i = 0; i < 16; i++) v |= (u128)b[i] << (8 * i);
        if (v >= wo_sampler_bound) continue;
        out[got++] = f_from_int(v);
    }
}
void xof_expand(const uint8_t seed[16], const uint8_t *dst, size_t dst_len, const uint8_t *binder,
                size_t binder_len, u128 *out, size_t n) {
    shake_t s;
    xof_init(&s, seed, dst, dst_len);
    shake_absorb(&s, binder, binder_len);
    shake_finalize(&s);
    xof_next_vec(&s, out, n);
}

/* ---------------------------------------------------------------------- fft ---- */
static size_t bitrev(unsigned d, size_t x) { /* fft.rs:118-124 */
    size_t y = 0;
    for (unsigned i = 0; i < d; i++) y += ((x >> i) & 1) << (d - i);
    return y >> 1;
}
static unsigned log2_exact(size_t n) { unsigned d = 0; while (((size_t)1 << d) < n) d++; return d; }

/* fft.rs:31-85 */
int fft(u128 *outp, const u128 *inp, size_t n_in, size_t size) {
    unsigned d = log2_exact(size);
    if (size != (size_t)1 << d || d > 20) return WO_ERR_PARAM;
    for (size_t i = 0; i < size; i++) {
        size_t j = bitrev(d, i);
        outp[i] = j < n_in ? inp[j] : 0;
    }
    for (unsigned l = 1; l <= d; l++) {
        u128 w = f_from_int(1);
        u128 r = f_root(l);
        size_t y = (size_t)1 << (l - 1);
        size_t chunk = (size / y) >> 1;
        for (size_t i = 0; i < y; i++) {
            if (i) w = f_mul(w, r);
            for (size_t j = 0; j < chunk; j++) {
                size_t x = (j << l) + i;
                u128 u = outp[x];
                u128 v = i ? f_mul(w, outp[x + y]) : outp[x + y];
                outp[x] = f_add(u, v);
                outp[x + y] = f_sub(u, v);
            }
        }
    }
    return 0;
}
/* fft.rs:102-115 */
void fft_inv_finish(u128 *outp, size_t size, u128 size_inv) {
    outp[0] = f_mul(outp[0], size_inv);
    if (size > 1) outp[size >> 1] = f_mul(outp[size >> 1], size_inv);
    for (size_t i = 1; i < size >> 1; i++) {
        u128 tmp = f_mul(outp[i], size_inv);
        outp[i] = f_mul(outp[size - i], size_inv);
        outp[size - i] = tmp;
    }
}
/* polynomial.rs:171-183 */
u128 poly_eval(const u128 *poly, size_t n, u128 at) {
    if (!n) return 0;
    u128 r = poly[n - 1];
    for (size_t i = n - 1; i-- > 0;) r = f_add(f_mul(r, at), poly[i]);
    return r;
}

/* ------------------------------------------------------------ C API wrappers ---- */
static u128 ld_res(const uint8_t b[16]) { /* residue bytes -> Montgomery, no range check */
    u128 v = 0;
    for (int i = 0; i < 16; i++) v |= (u128)b[i] << (8 * i);
    return f_from_int(v % WO_P);
}
void wo_f128_mul(const uint8_t a[16], const uint8_t b[16], uint8_t o[16]) { f_encode(f_mul(ld_res(a), ld_res(b)), o); }
void wo_f128_add(const uint8_t a[16], const uint8_t b[16], uint8_t o[16]) { f_encode(f_add(ld_res(a), ld_res(b)), o); }
void wo_f128_sub(const uint8_t a[16], const uint8_t b[16], uint8_t o[16]) { f_encode(f_sub(ld_res(a), ld_res(b)), o); }
void wo_f128_inv(const uint8_t a[16], uint8_t o[16]) { f_encode(f_inv(ld_res(a)), o); }
static void st128(u128 v, uint8_t b[16]) { for (int i = 0; i < 16; i++) b[i] = (uint8_t)(v >> (8 * i)); }
void wo_f128_constants(uint8_t p[16], uint8_t r2[16], uint8_t g_mont[16], uint64_t *mu) {
    wo_init();
    st128(WO_P, p); st128(R2, r2); st128(ROOTS[66], g_mont); *mu = MU;
}
void wo_f128_root(unsigned l, uint8_t out[16]) { f_encode(f_root(l), out); }
void wo_xof_expand_f128(const uint8_t seed[16], const uint8_t *dst, size_t dst_len, const uint8_t *binder,
                        size_t binder_len, size_t n, uint8_t *out) {
    u128 *v = wo_alloc(n * sizeof(u128));
    xof_expand(seed, dst, dst_len, binder, binder_len, v, n);
    for (size_t i = 0; i < n; i++) f_encode(v[i], out + 16 * i);
    wo_free(v);
}
void wo_xof_derive_seed(const uint8_t seed[16], const uint8_t *dst, size_t dst_len, const uint8_t *binder,
                        size_t binder_len, uint8_t out[16]) {
    shake_t s;
    xof_init(&s, seed, dst, dst_len);
    shake_absorb(&s, binder, binder_len);
    xof_into_seed(&s, out);
}
int wo_fft_f128(const uint8_t *in, size_t n_in, size_t size, uint8_t *out) {
    u128 *a = wo_alloc(n_in * sizeof(u128) + 16), *o = wo_alloc(size * sizeof(u128));
    for (size_t i = 0; i < n_in; i++) a[i] = ld_res(in + 16 * i);
    int rc = fft(o, a, n_in, size);
    if (!rc) for (size_t i = 0; i < size; i++) f_encode(o[i], out + 16 * i);
    wo_free(a); wo_free(o);
    return rc;
}
int wo_ifft_f128(const uint8_t *in, size_t n_in, size_t size, uint8_t *out) {
    u128 *a = wo_alloc(n_in * sizeof(u128) + 16), *o = wo_alloc(size * sizeof(u128));
    for (size_t i = 0; i < n_in; i++) a[i] = ld_res(in + 16 * i);
    int rc = fft(o, a, n_in, size);
    if (!rc) {
        fft_inv_finish(o, size, f_inv(f_from_int(size)));
        for (size_t i = 0; i < size; i++) f_encode(o[i], out + 16 * i);
    }
    wo_free(a); wo_free(o);
    return rc;
}
