/*
 * ORACLE (test infrastructure).  Heavy-hitter key chains: verifiable DPF generation and
 * evaluation, and the server's eval_and_check, restated from
 *   common/src/prg.rs:128-241     PrgSeed::{get_lsb, expand_direction, convert, mmo_hash2to4}
 *   common/src/prg.rs:280-475     FixedKeyPrgStream: AES-128 under the all-zero key, Matyas-Meyer-Oseas
 *                                 out = AES_0(ctr) ^ ctr; `set_key` loads the COUNTER; inc_be adds 1 to the
 *                                 upper 64-bit lane (_mm_add_epi64 with set_epi64x(1, 0)); refill8 for
 *                                 requests above 64 bytes
 *   common/src/prg.rs:517-631     HasherStream / mmo_hash4to4: the same MMO under the key [1; 16]
 *   hhcore/src/dpf.rs:85-145      gen_correction, gen_output_correction
 *   hhcore/src/vdpf.rs:54-214     VDPFKey::{gen, eval_all}
 *   hhcore/src/protocol.rs:30-176 gen_key_chain, eval_and_check
 *   hhcore/src/bucket.rs:95-110, common/src/group.rs:116-120,294-317  sampling of u32 / Bucket<u32>
 *   common/src/group.rs:437-516   IntModN (note `one()` has val 0: the index hashed at every leaf of
 *                                 eval_all stays 0, while gen hashes the real alpha; both servers do the
 *                                 same, so their proofs still agree -- reproduced here as written)
 *   server-hh/src/main.rs:255-265 testing id = AES_vk(id.to_le_bytes()) read little endian
 * Wire format: bincode 1.3.3 defaults (serialize/src/lib.rs:72-80): fixed-width little-endian integers,
 * u64 sequence lengths, bool as one byte, arrays/tuples/structs in line.
 *
 * Third-party pieces not in /root/reference: `aes = "0.8"` (FIPS-197 AES-128), `sha256 = "1.1.2"`
 * (FIPS 180-4 SHA-256, lower-case hex), `bincode = "1.3.3"`.  AES is pinned by the FIPS-197 C.1 vector and
 * SHA-256 by Python's hashlib in tests/test_oracle_vdpf.py.  The reference has no known-answer vector
 * for any VDPF value (its tests draw from thread_rng): "parity unpinned" for the tag and bucket VALUES;
 * what is pinned are the reference's own properties (vdpf.rs:225-305, protocol.rs:191-231): equal proofs
 * on the two servers, reconstruction to beta at alpha and zero elsewhere, unequal proofs for malformed keys.
 */
#include "oracle_internal.h"
#include "wh_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ AES-128, fixed keys */
typedef struct { uint8_t rk[176]; } aes_rk;
static aes_rk K0, K1;
static uint8_t SB[256];
static int tables_ready, have_aesni;
static pthread_once_t once = PTHREAD_ONCE_INIT;

static uint8_t xt(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }
static void expand_key(const uint8_t key[16], aes_rk *out) {
    uint8_t *rk = out->rk, rcon = 1;
    memcpy(rk, key, 16);
    for (int i = 16; i < 176; i += 4) {
        uint8_t t[4] = {rk[i - 4], rk[i - 3], rk[i - 2], rk[i - 1]};
        if (i % 16 == 0) {
            uint8_t u = t[0];
            t[0] = SB[t[1]] ^ rcon; t[1] = SB[t[2]]; t[2] = SB[t[3]]; t[3] = SB[u];
            rcon = xt(rcon);
        }
        for (int k = 0; k < 4; k++) rk[i + k] = rk[i - 16 + k] ^ t[k];
    }
}
static void init_tables(void) {
    /* S-box from the AES block function itself: S[x] = first byte of SubBytes on a state of x's,
     * recovered through wo_aes128_encrypt_block's building blocks would be circular, so rebuild it:
     * inverse in GF(2^8) then the affine map (FIPS 197 5.1.1) */
    for (int x = 0; x < 256; x++) {
        uint8_t inv = 0;
        if (x) {
            for (int y = 1; y < 256 && !inv; y++) {
                uint8_t a = (uint8_t)x, b = (uint8_t)y, r = 0;
                while (b) { if (b & 1) r ^= a; a = xt(a); b >>= 1; }
                if (r == 1) inv = (uint8_t)y;
            }
        }
        uint8_t s = inv, r = inv;
        for (int i = 0; i < 4; i++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        SB[x] = s ^ 0x63;
    }
    uint8_t k[16];
    memset(k, 0, 16); expand_key(k, &K0);
    memset(k, 1, 16); expand_key(k, &K1);
#if defined(__x86_64__)
    have_aesni = __builtin_cpu_supports("aes") && !getenv("WO_NO_AESNI");
#endif
    tables_ready = 1;
}
static void aes_portable(const aes_rk *k, const uint8_t in[16], uint8_t out[16]) {
    uint8_t s[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ k->rk[i];
    for (int round = 1; round <= 10; round++) {
        uint8_t t[16];
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) t[4 * c + r] = SB[s[4 * ((c + r) & 3) + r]];
        if (round < 10)
            for (int c = 0; c < 4; c++) {
                uint8_t *a = t + 4 * c, b0 = a[0], b1 = a[1], b2 = a[2], b3 = a[3];
                a[0] = xt(b0) ^ xt(b1) ^ b1 ^ b2 ^ b3;
                a[1] = b0 ^ xt(b1) ^ xt(b2) ^ b2 ^ b3;
                a[2] = b0 ^ b1 ^ xt(b2) ^ xt(b3) ^ b3;
                a[3] = xt(b0) ^ b0 ^ b1 ^ b2 ^ xt(b3);
            }
        for (int i = 0; i < 16; i++) s[i] = t[i] ^ k->rk[16 * round + i];
    }
    memcpy(out, s, 16);
}
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("aes,sse2"))) static void aes_ni(const aes_rk *k, const uint8_t in[16], uint8_t out[16]) {
    const __m128i *rk = (const __m128i *)k->rk;
    __m128i s = _mm_xor_si128(_mm_loadu_si128((const __m128i *)in), _mm_loadu_si128(rk));
    for (int r = 1; r < 10; r++) s = _mm_aesenc_si128(s, _mm_loadu_si128(rk + r));
    s = _mm_aesenclast_si128(s, _mm_loadu_si128(rk + 10));
    _mm_storeu_si128((__m128i *)out, s);
}
#endif
#if defined(__x86_64__)
/* eight counters at once, as refill8 hands them to encrypt_blocks (prg.rs:368-403): the AES-NI pipeline stays full */
__attribute__((target("aes,sse2"))) static void mmo8_ni(const aes_rk *k, const uint8_t ctr[16], uint8_t out[128]) {
    const __m128i *rk = (const __m128i *)k->rk;
    const __m128i one_hi = _mm_set_epi64x(1, 0);
    __m128i c[8], s[8];
    c[0] = _mm_loadu_si128((const __m128i *)ctr);
    for (int i = 1; i < 8; i++) c[i] = _mm_add_epi64(c[i - 1], one_hi);
    const __m128i k0 = _mm_loadu_si128(rk);
    for (int i = 0; i < 8; i++) s[i] = _mm_xor_si128(c[i], k0);
    for (int r = 1; r < 10; r++) {
        const __m128i kr = _mm_loadu_si128(rk + r);
        for (int i = 0; i < 8; i++) s[i] = _mm_aesenc_si128(s[i], kr);
    }
    const __m128i kl = _mm_loadu_si128(rk + 10);
    for (int i = 0; i < 8; i++) _mm_storeu_si128((__m128i *)(out + 16 * i), _mm_xor_si128(_mm_aesenclast_si128(s[i], kl), c[i]));
}
#endif
static void aes_enc(const aes_rk *k, const uint8_t in[16], uint8_t out[16]) {
#if defined(__x86_64__)
    if (have_aesni) { aes_ni(k, in, out); return; }
#endif
    aes_portable(k, in, out);
}
/* Matyas-Meyer-Oseas under a fixed key (prg.rs:311-331, 569-593) */
static void mmo(const aes_rk *k, const uint8_t in[16], uint8_t out[16]) {
    uint8_t c[16], e[16];
    memcpy(c, in, 16);
    aes_enc(k, c, e);
    for (int i = 0; i < 16; i++) out[i] = e[i] ^ c[i];
}
/* inc_be (prg.rs:449-452): +1 on the upper 64-bit lane */
static void ctr_add(uint8_t c[16], uint64_t k) {
    uint64_t hi = 0;
    for (int i = 0; i < 8; i++) hi |= (uint64_t)c[8 + i] << (8 * i);
    hi += k;
    for (int i = 0; i < 8; i++) c[8 + i] = (uint8_t)(hi >> (8 * i));
}

/* ------------------------------------------------------------------ SHA-256 (FIPS 180-4) */
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha_block(uint32_t h[8], const uint8_t b[64]) {
    uint32_t w[64], v[8];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)b[4 * i] << 24 | (uint32_t)b[4 * i + 1] << 16 | (uint32_t)b[4 * i + 2] << 8 | b[4 * i + 3];
    for (int i = 16; i < 64; i++)
        w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] +
               (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
    memcpy(v, h, 32);
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = v[7] + (ror(v[4], 6) ^ ror(v[4], 11) ^ ror(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + SHA_K[i] + w[i];
        uint32_t t2 = (ror(v[0], 2) ^ ror(v[0], 13) ^ ror(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
        v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) h[i] += v[i];
}
void wo_sha256(const uint8_t *in, size_t n, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t i = 0;
    for (; i + 64 <= n; i += 64) sha_block(h, in + i);
    uint8_t tail[128];
    size_t r = n - i, tl = r + 9 <= 64 ? 64 : 128;
    memset(tail, 0, sizeof tail);
    if (r) memcpy(tail, in + i, r);
    tail[r] = 0x80;
    uint64_t bits = (uint64_t)n * 8;
    for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    sha_block(h, tail);
    if (tl == 128) sha_block(h, tail + 64);
    for (int k = 0; k < 8; k++) { out[4 * k] = (uint8_t)(h[k] >> 24); out[4 * k + 1] = (uint8_t)(h[k] >> 16); out[4 * k + 2] = (uint8_t)(h[k] >> 8); out[4 * k + 3] = (uint8_t)h[k]; }
}
/* sha256::digest: lower-case hex of the hash */
static void sha_hex(const uint8_t *in, size_t n, char out[64]) {
    uint8_t d[32];
    wo_sha256(in, n, d);
    for (int i = 0; i < 32; i++) { out[2 * i] = "0123456789abcdef"[d[i] >> 4]; out[2 * i + 1] = "0123456789abcdef"[d[i] & 15]; }
}

/* ------------------------------------------------------------------ PrgSeed operations */
#define BKT 258 /* Bucket<u32>: 256 data + sign_data + ctr (bucket.rs:6-25) */
#define MAXD 16

typedef struct { uint8_t seed[16]; uint8_t b[2]; } cw_t;
typedef struct {
    uint8_t key_id, root[16];
    uint32_t depth;
    cw_t cw[MAXD];
    uint8_t cs[64];
    uint32_t oc[BKT]; /* OutputCorrection: a Bucket for the main key, oc[0] for a support key */
} vkey;

/* expand_direction(true, true) (prg.rs:133-174): control bits come from the two low bits of the seed itself */
static void prg_expand(const uint8_t seed[16], uint8_t left[16], uint8_t right[16], int bits[2]) {
    uint8_t c[16];
    memcpy(c, seed, 16);
    bits[0] = (c[0] & 1) == 0; bits[1] = (c[0] & 2) == 0;
    c[0] &= 0xfc;
    mmo(&K0, c, left);
    ctr_add(c, 1);
    mmo(&K0, c, right);
}
/* convert::<u32> (prg.rs:180-198 -> next_u32 -> fill_bytes -> refill): first 4 bytes of MMO(seed) */
static uint32_t convert_u32(const uint8_t seed[16]) {
    uint8_t o[16];
    mmo(&K0, seed, o);
    return (uint32_t)o[0] | (uint32_t)o[1] << 8 | (uint32_t)o[2] << 16 | (uint32_t)o[3] << 24;
}
/* convert::<Bucket<u32>>: 1032 bytes through refill8 (nine batches of eight counters, the last partly used) */
static void convert_bucket(const uint8_t seed[16], uint32_t out[BKT]) {
    uint8_t c[16], buf[72 * 16];
    memcpy(c, seed, 16);
#if defined(__x86_64__)
    if (have_aesni) {
        for (int i = 0; i < 9; i++) { mmo8_ni(&K0, c, buf + 128 * i); ctr_add(c, 8); }
    } else
#endif
    for (int i = 0; i < 72; i++) { mmo(&K0, c, buf + 16 * i); ctr_add(c, 1); }
    for (int i = 0; i < BKT; i++) out[i] = (uint32_t)buf[4 * i] | (uint32_t)buf[4 * i + 1] << 8 | (uint32_t)buf[4 * i + 2] << 16 | (uint32_t)buf[4 * i + 3] << 24;
}
/* mmo_hash2to4 (prg.rs:202-241); x_bytes = GIn::as_bytes zero-padded to a block */
static void hash2to4(const uint8_t seed[16], const uint8_t x_bytes[16], uint8_t out[64]) {
    mmo(&K0, x_bytes, out);
    mmo(&K0, seed, out + 16);
    mmo(&K0, out, out + 32);
    mmo(&K0, out + 16, out + 48);
}
static void hash4to4(const uint8_t in[64], uint8_t out[64]) { /* prg.rs:599-631 */
    for (int i = 0; i < 4; i++) mmo(&K1, in + 16 * i, out + 16 * i);
}
static void xor_n(uint8_t *d, const uint8_t *a, int n) { for (int i = 0; i < n; i++) d[i] ^= a[i]; }

/* ------------------------------------------------------------------ VDPF gen (vdpf.rs:54-113) */
/* words: BKT for Bucket<u32> output, 1 for u32.  x_bytes: as_bytes of alpha.  alpha bits MSB first. */
static void vdpf_gen(uint32_t depth, uint32_t alpha, const uint8_t x_bytes[16], const uint32_t *beta, int words,
                     shake_t *rng, vkey *k0, vkey *k1) {
    for (;;) {
        uint8_t seeds[2][16], roots[2][16];
        int bits[2] = {0, 1};
        shake_squeeze(rng, roots[0], 16); /* PrgSeed::random() x2; thread_rng replaced by a seeded stream */
        shake_squeeze(rng, roots[1], 16);
        memcpy(seeds, roots, 32);
        cw_t cw[MAXD];
        for (uint32_t l = 0; l < depth; l++) {
            int cur = (alpha >> (depth - 1 - l)) & 1; /* decompose(): MSB to LSB (group.rs:492-500) */
            /* gen_correction (dpf.rs:85-122) */
            uint8_t ex[2][2][16];
            int eb[2][2];
            for (int p = 0; p < 2; p++) prg_expand(seeds[p], ex[p][0], ex[p][1], eb[p]);
            int special = cur, fix = !cur;
            for (int i = 0; i < 16; i++) cw[l].seed[i] = ex[0][fix][i] ^ ex[1][fix][i];
            cw[l].b[0] = (uint8_t)(eb[0][0] ^ eb[1][0] ^ cur ^ 1);
            cw[l].b[1] = (uint8_t)(eb[0][1] ^ eb[1][1] ^ cur);
            for (int p = 0; p < 2; p++) {
                memcpy(seeds[p], ex[p][special], 16);
                int nb = eb[p][special];
                if (bits[p]) { xor_n(seeds[p], cw[l].seed, 16); nb ^= cw[l].b[special]; }
                bits[p] = nb;
            }
        }
        uint8_t pr0[64], pr1[64];
        hash2to4(seeds[0], x_bytes, pr0);
        hash2to4(seeds[1], x_bytes, pr1);
        xor_n(pr0, pr1, 64); /* cs */
        bits[0] = seeds[0][0] & 1; bits[1] = seeds[1][0] & 1;
        if (bits[0] == bits[1]) continue; /* resample (vdpf.rs:86-91) */
        /* gen_output_correction (dpf.rs:124-145) */
        uint32_t c0[BKT], c1[BKT], oc[BKT];
        if (words == BKT) { convert_bucket(seeds[0], c0); convert_bucket(seeds[1], c1); }
        else { c0[0] = convert_u32(seeds[0]); c1[0] = convert_u32(seeds[1]); }
        for (int i = 0; i < words; i++) { oc[i] = beta[i] - c0[i] + c1[i]; if (bits[1]) oc[i] = 0u - oc[i]; }
        vkey *ks[2] = {k0, k1};
        for (int p = 0; p < 2; p++) {
            memset(ks[p], 0, sizeof(vkey));
            ks[p]->key_id = (uint8_t)p; ks[p]->depth = depth;
            memcpy(ks[p]->root, roots[p], 16);
            memcpy(ks[p]->cw, cw, sizeof cw);
            memcpy(ks[p]->cs, pr0, 64);
            memcpy(ks[p]->oc, oc, 4 * (size_t)words);
        }
        return;
    }
}

/* ------------------------------------------------------------------ VDPF eval_all (vdpf.rs:123-214) */
/* x_mode 0: IntModN, whose one() is 0, so the hashed index never moves; 1: bool / integers, index counts up.
 * out[2^depth][words]; proof: SHA-256 of pi as 64 lower-case hex characters. */
static void vdpf_eval_all(const vkey *k, int words, int x_mode, uint32_t *out, char proof_hex[64]) {
    const uint32_t depth = k->depth, n = 1u << depth;
    uint8_t (*seed)[16] = malloc((size_t)(2 * n) * 16);
    uint8_t *bit = malloc(2 * n);
    memcpy(seed[0], k->root, 16); bit[0] = k->key_id;
    for (uint32_t i = 0, level = 0; i + 1 < n; i++) {
        if (i + 1 == (2u << level)) level++; /* node i sits at level floor(log2(i+1)) */
        int eb[2];
        uint8_t *l = seed[2 * i + 1], *r = seed[2 * i + 2];
        prg_expand(seed[i], l, r, eb);
        if (bit[i]) {
            xor_n(l, k->cw[level].seed, 16); eb[0] ^= k->cw[level].b[0];
            xor_n(r, k->cw[level].seed, 16); eb[1] ^= k->cw[level].b[1];
        }
        bit[2 * i + 1] = (uint8_t)eb[0]; bit[2 * i + 2] = (uint8_t)eb[1];
    }
    uint8_t pi[64];
    memcpy(pi, k->cs, 64);
    for (uint32_t x = 0; x < n; x++) {
        const uint8_t *s = seed[n - 1 + x];
        uint8_t xb[16] = {0}, tp[64], h[64];
        if (x_mode) { xb[0] = (uint8_t)x; xb[1] = (uint8_t)(x >> 8); }
        hash2to4(s, xb, tp);
        int t = s[0] & 1; /* get_lsb of the leaf seed, not the carried control bit */
        uint32_t *w = out + (size_t)x * words;
        if (words == BKT) convert_bucket(s, w); else w[0] = convert_u32(s);
        if (t) { for (int i = 0; i < words; i++) w[i] += k->oc[i]; xor_n(tp, k->cs, 64); }
        if (k->key_id) for (int i = 0; i < words; i++) w[i] = 0u - w[i];
        xor_n(tp, pi, 64);       /* h_prime_in = pi ^ corrected */
        hash4to4(tp, h);
        xor_n(pi, h, 64);
    }
    sha_hex(pi, 64, proof_hex);
    free(seed); free(bit);
}

/* ------------------------------------------------------------------ bincode layout of KeyChain<u32> */
#define SUPPORT_KEYS 257
#define SUPPORT_LEN (1 + 16 + 8 + 18 + 64 + 4)
static size_t main_len(uint32_t depth) { return 1 + 16 + 8 + 18 * (size_t)depth + 64 + 8 + 4 * 256 + 4 + 4; }
uint64_t wo_hh_key_chain_len(uint32_t depth) { return 16 + main_len(depth) + 8 + (uint64_t)SUPPORT_KEYS * SUPPORT_LEN; }

static void put64(uint8_t *p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i)); }
static void put32(uint8_t *p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (uint8_t)(v >> (8 * i)); }
static uint64_t get64(const uint8_t *p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i); return v; }
static uint32_t get32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

static uint8_t *enc_vkey(uint8_t *p, const vkey *k, int words) {
    *p++ = k->key_id;
    memcpy(p, k->root, 16); p += 16;
    put64(p, k->depth); p += 8;
    for (uint32_t l = 0; l < k->depth; l++) { memcpy(p, k->cw[l].seed, 16); p[16] = k->cw[l].b[0]; p[17] = k->cw[l].b[1]; p += 18; }
    memcpy(p, k->cs, 64); p += 64;
    if (words == BKT) { put64(p, 256); p += 8; }
    for (int i = 0; i < words; i++) { put32(p, k->oc[i]); p += 4; }
    return p;
}
/* returns NULL when bincode would fail or a length disagrees with the configured shape */
static const uint8_t *dec_vkey(const uint8_t *p, uint32_t depth, int words, vkey *k) {
    memset(k, 0, sizeof *k);
    if (*p > 1) return NULL;
    k->key_id = *p++;
    memcpy(k->root, p, 16); p += 16;
    if (get64(p) != depth) return NULL;
    p += 8; k->depth = depth;
    for (uint32_t l = 0; l < depth; l++) {
        memcpy(k->cw[l].seed, p, 16);
        if (p[16] > 1 || p[17] > 1) return NULL;
        k->cw[l].b[0] = p[16]; k->cw[l].b[1] = p[17]; p += 18;
    }
    memcpy(k->cs, p, 64); p += 64;
    if (words == BKT) { if (get64(p) != 256) return NULL; p += 8; }
    for (int i = 0; i < words; i++) { k->oc[i] = get32(p); p += 4; }
    return p;
}

/* ------------------------------------------------------------------ gen_key_chain (protocol.rs:31-76) */
int wo_hh_gen_key_chain(uint32_t depth, const uint8_t id_le[16], uint32_t alpha, const uint32_t bucket[BKT],
                        const uint8_t rng_seed[32], uint8_t *kc0, uint8_t *kc1) {
    pthread_once(&once, init_tables);
    if (depth < 1 || depth > MAXD || alpha >> depth) return WO_ERR_PARAM;
    for (int i = 0; i < 257; i++) if (bucket[i] != 1u && bucket[i] != 0xffffffffu) return WO_ERR_PARAM; /* "Data is not -1 or 1" */
    shake_t rng;
    shake_init(&rng); shake_absorb(&rng, rng_seed, 32); shake_finalize(&rng);
    vkey *m = malloc(2 * sizeof(vkey));
    uint8_t xb[16] = {0};
    xb[0] = (uint8_t)alpha; xb[1] = (uint8_t)(alpha >> 8); /* IntModN::as_bytes: two bytes, little endian */
    vdpf_gen(depth, alpha, xb, bucket, BKT, &rng, &m[0], &m[1]);
    uint8_t *p[2] = {kc0, kc1};
    for (int s = 0; s < 2; s++) {
        memcpy(p[s], id_le, 16); p[s] += 16;
        p[s] = enc_vkey(p[s], &m[s], BKT);
        put64(p[s], SUPPORT_KEYS); p[s] += 8;
    }
    for (int i = 0; i < SUPPORT_KEYS; i++) {
        /* data[i] (then sign_data) == 1 -> point true, == -1 -> point false; payload 1 */
        uint32_t a = bucket[i] == 1u, one = 1;
        uint8_t ab[16] = {0};
        ab[0] = (uint8_t)a; /* bool::as_bytes */
        vdpf_gen(1, a, ab, &one, 1, &rng, &m[0], &m[1]);
        for (int s = 0; s < 2; s++) p[s] = enc_vkey(p[s], &m[s], 1);
    }
    free(m);
    return 0;
}

/* ------------------------------------------------------------------ eval_and_check (protocol.rs:78-176) */
int wo_hh_eval_and_check(uint32_t depth, const uint8_t *kc, uint64_t kc_len, uint32_t *dat, uint8_t tag_hex[64]) {
    pthread_once(&once, init_tables);
    if (depth < 1 || depth > MAXD) return WO_ERR_PARAM;
    if (kc_len < wo_hh_key_chain_len(depth)) return WO_ERR_SHORT;
    const uint32_t n = 1u << depth;
    vkey *k = malloc(sizeof(vkey));
    const uint8_t *p = dec_vkey(kc + 16, depth, BKT, k);
    if (!p || get64(p) != SUPPORT_KEYS) { free(k); return WO_ERR_KEY_FORMAT; }
    p += 8;
    const int key_id = k->key_id;
    char *proofs = malloc(64 * (SUPPORT_KEYS + 3));
    uint32_t *own = dat ? dat : malloc((size_t)n * BKT * 4);
    vdpf_eval_all(k, BKT, 0, own, proofs);
    uint32_t buck[BKT] = {0}, ones[SUPPORT_KEYS];
    for (uint32_t x = 0; x < n; x++) for (int i = 0; i < BKT; i++) buck[i] += own[(size_t)x * BKT + i];
    if (!dat) free(own);
    int rc = 0;
    for (int i = 0; i < SUPPORT_KEYS; i++) {
        p = dec_vkey(p, 1, 1, k);
        if (!p) { rc = WO_ERR_KEY_FORMAT; break; }
        uint32_t d[2];
        vdpf_eval_all(k, 1, 1, d, proofs + 64 * (1 + i));
        ones[i] = d[0] + d[1];
        buck[i] -= d[1] - d[0]; /* data[i], and sign_data for the last one (index 256) */
    }
    if (!rc) {
        if (key_id) { buck[257] -= 1; for (int i = 0; i < BKT; i++) buck[i] = 0u - buck[i]; }
        uint8_t bytes[4 * BKT];
        for (int i = 0; i < BKT; i++) put32(bytes + 4 * i, buck[i]);
        sha_hex(bytes, 4 * BKT, proofs + 64 * (SUPPORT_KEYS + 1));
        for (int i = 0; i < SUPPORT_KEYS; i++) { uint32_t v = key_id ? 0u - (ones[i] - 1) : ones[i]; put32(bytes + 4 * i, v); }
        sha_hex(bytes, 4 * SUPPORT_KEYS, proofs + 64 * (SUPPORT_KEYS + 2));
        sha_hex((const uint8_t *)proofs, 64 * (SUPPORT_KEYS + 3), (char *)tag_hex);
    }
    free(proofs); free(k);
    return rc;
}

/* the expansion loop of batch_collect_and_aggregate_per_run (server-hh/src/main.rs:255-265), `nthreads` host threads */
typedef struct {
    uint32_t depth; const uint8_t *vk, *keys; uint64_t stride, lo, hi;
    uint32_t *dat; uint8_t *tags, *ids; int32_t *status;
} hh_job;
static void *hh_worker(void *arg) {
    hh_job *j = arg;
    const uint64_t words = (uint64_t)BKT << j->depth;
    for (uint64_t i = j->lo; i < j->hi; i++) {
        const uint8_t *kc = j->keys + i * j->stride;
        j->status[i] = wo_hh_eval_and_check(j->depth, kc, j->stride, j->dat ? j->dat + i * words : NULL, j->tags + 64 * i);
        if (j->status[i]) { memset(j->tags + 64 * i, 0, 64); if (j->dat) memset(j->dat + i * words, 0, 4 * words); }
        wo_aes128_encrypt_block(j->vk, kc, j->ids + 16 * i); /* id is already little endian on the wire */
    }
    return NULL;
}
int wo_hh_eval_and_check_batch(uint32_t depth, const uint8_t verify_key[16], const uint8_t *keys, uint64_t stride,
                               uint64_t n, unsigned nthreads, uint32_t *dat, uint8_t *tags_hex, uint8_t *testing_ids,
                               int32_t *status) {
    pthread_once(&once, init_tables);
    if (depth < 1 || depth > MAXD || stride < wo_hh_key_chain_len(depth)) return WO_ERR_PARAM;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    hh_job jobs[256];
    for (unsigned t = 0; t < nthreads; t++) {
        jobs[t] = (hh_job){depth, verify_key, keys, stride, n * t / nthreads, n * (t + 1) / nthreads, dat, tags_hex, testing_ids, status};
        if (nthreads == 1) hh_worker(&jobs[t]); else pthread_create(&th[t], NULL, hh_worker, &jobs[t]);
    }
    if (nthreads > 1) for (unsigned t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    return 0;
}

/* unit surface for the tests */
void wo_mmo_fixed(int key_byte, const uint8_t in[16], uint8_t out[16]) {
    pthread_once(&once, init_tables);
    mmo(key_byte ? &K1 : &K0, in, out);
}
int wo_aes_backend(void) { pthread_once(&once, init_tables); return have_aesni; }
