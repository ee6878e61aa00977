/*
 * ORACLE (test infrastructure).  AES-128 single-block encryption (FIPS 197) for the
 * testing-id PRF, and the heavy-hitter bucket sum.
 *   common/src/prg.rs:639-655  Prf::compute_prf = AES128-ECB_k(block), via the un-vendored
 *                              `aes = "0.8"` crate; no test in the reference pins it
 *                              (prg.rs:694-701 is commented out) -> pinned here by the
 *                              FIPS-197 appendix C.1 vector instead ("parity unpinned"
 *                              w.r.t. the reference itself; ids only order reports).
 *   hhcore/src/bucket.rs:48-55, common/src/group.rs:88-91  Bucket::add = u32 wrapping_add
 *   server-hh/src/main.rs:72-104  reduce over clients, bad ids skipped
 */
#include "oracle_internal.h"
#include <string.h>

static uint8_t SBOX[256];
static int sbox_ready;
static uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }
static uint8_t gmul(uint8_t a, uint8_t b) {
    uint8_t r = 0;
    while (b) { if (b & 1) r ^= a; a = xtime(a); b >>= 1; }
    return r;
}
static void make_sbox(void) {
    /* multiplicative inverse in GF(2^8) followed by the affine map */
    for (int x = 0; x < 256; x++) {
        uint8_t inv = 0;
        if (x) for (int y = 1; y < 256; y++) if (gmul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        uint8_t s = inv, r = inv;
        for (int i = 0; i < 4; i++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        SBOX[x] = s ^ 0x63;
    }
    sbox_ready = 1;
}
void wo_aes128_encrypt_block(const uint8_t key[16], const uint8_t in[16], uint8_t out[16]) {
    if (!sbox_ready) make_sbox();
    uint8_t rk[176], s[16];
    memcpy(rk, key, 16);
    uint8_t rcon = 1;
    for (int i = 16; i < 176; i += 4) {
        uint8_t t[4] = {rk[i - 4], rk[i - 3], rk[i - 2], rk[i - 1]};
        if (i % 16 == 0) {
            uint8_t u = t[0];
            t[0] = SBOX[t[1]] ^ rcon; t[1] = SBOX[t[2]]; t[2] = SBOX[t[3]]; t[3] = SBOX[u];
            rcon = xtime(rcon);
        }
        for (int k = 0; k < 4; k++) rk[i + k] = rk[i - 16 + k] ^ t[k];
    }
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[i];
    for (int round = 1; round <= 10; round++) {
        uint8_t t[16];
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++) t[4 * c + r] = SBOX[s[4 * ((c + r) & 3) + r]]; /* SubBytes + ShiftRows */
        if (round < 10) {
            for (int c = 0; c < 4; c++) {
                uint8_t *a = t + 4 * c, b0 = a[0], b1 = a[1], b2 = a[2], b3 = a[3];
                a[0] = xtime(b0) ^ xtime(b1) ^ b1 ^ b2 ^ b3;
                a[1] = b0 ^ xtime(b1) ^ xtime(b2) ^ b2 ^ b3;
                a[2] = b0 ^ b1 ^ xtime(b2) ^ xtime(b3) ^ b3;
                a[3] = xtime(b0) ^ b0 ^ b1 ^ b2 ^ xtime(b3);
            }
        }
        for (int i = 0; i < 16; i++) s[i] = t[i] ^ rk[16 * round + i];
    }
    memcpy(out, s, 16);
}

void wo_hh_bucket_sum_u32(const uint32_t *buckets, const uint8_t *keep, uint64_t n, uint64_t words, uint32_t *acc) {
    memset(acc, 0, words * sizeof(uint32_t));
    for (uint64_t i = 0; i < n; i++) {
        if (keep && !keep[i]) continue;
        for (uint64_t w = 0; w < words; w++) acc[w] += buckets[i * words + w];
    }
}
