// Host-side AES-128 (FIPS 197) for table set-up and the CPU client: key schedule as little-endian column
// words and a one-table (Te0 + rotations) block function.  The device kernels use the same table and round-key
// words (hh_dev.hpp).  The reference gets AES from the un-vendored `aes = "0.8"` crate (common/src/prg.rs:5-9).
#pragma once
#include <cstdint>
#include <cstring>

namespace wh {

struct AesTables {
    uint8_t sbox[256];
    uint32_t te0[256];  // bytes (2s, s, s, 3s), low byte first: the MixColumns column of a row-0 input
    AesTables() {
        // S-box by walking the multiplicative group with generator 3: p runs over 3^k, q over 3^-k
        uint8_t p = 1, q = 1;
        do {
            p = (uint8_t)(p ^ (p << 1) ^ ((p & 0x80) ? 0x1b : 0));
            q ^= (uint8_t)(q << 1); q ^= (uint8_t)(q << 2); q ^= (uint8_t)(q << 4);
            if (q & 0x80) q ^= 0x09;
            const uint8_t x = (uint8_t)(q ^ rol8(q, 1) ^ rol8(q, 2) ^ rol8(q, 3) ^ rol8(q, 4));
            sbox[p] = (uint8_t)(x ^ 0x63);
        } while (p != 1);
        sbox[0] = 0x63;
        for (int i = 0; i < 256; i++) {
            const uint32_t s = sbox[i], s2 = ((s << 1) ^ ((s >> 7) * 0x11b)) & 0xff, s3 = s2 ^ s;
            te0[i] = s2 | (s << 8) | (s << 16) | (s3 << 24);
        }
    }
    static uint8_t rol8(uint8_t v, int n) { return (uint8_t)((v << n) | (v >> (8 - n))); }
};
inline const AesTables &aes_tables_host() { static const AesTables t; return t; }

struct AesKey {
    uint32_t rk[44];  // round r, column c at rk[4r + c]; byte `row` of the word is state row `row`
    explicit AesKey(const uint8_t key[16]) {
        const AesTables &t = aes_tables_host();
        memcpy(rk, key, 16);
        uint32_t rcon = 1;
        for (int i = 4; i < 44; i++) {
            uint32_t w = rk[i - 1];
            if ((i & 3) == 0) {
                w = (w >> 8) | (w << 24);  // RotWord on little-endian columns
                w = (uint32_t)t.sbox[w & 0xff] | (uint32_t)t.sbox[(w >> 8) & 0xff] << 8 |
                    (uint32_t)t.sbox[(w >> 16) & 0xff] << 16 | (uint32_t)t.sbox[w >> 24] << 24;
                w ^= rcon;
                rcon = ((rcon << 1) ^ ((rcon >> 7) * 0x11b)) & 0xff;
            }
            rk[i] = rk[i - 4] ^ w;
        }
    }
    static uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
    void encrypt(const uint8_t in[16], uint8_t out[16]) const {
        const AesTables &t = aes_tables_host();
        uint32_t s[4], u[4];
        memcpy(s, in, 16);
        for (int c = 0; c < 4; c++) s[c] ^= rk[c];
        for (int r = 1; r < 10; r++) {
            for (int c = 0; c < 4; c++)
                u[c] = t.te0[s[c] & 0xff] ^ rotl(t.te0[(s[(c + 1) & 3] >> 8) & 0xff], 8) ^
                       rotl(t.te0[(s[(c + 2) & 3] >> 16) & 0xff], 16) ^ rotl(t.te0[s[(c + 3) & 3] >> 24], 24) ^ rk[4 * r + c];
            memcpy(s, u, 16);
        }
        for (int c = 0; c < 4; c++)
            u[c] = ((uint32_t)t.sbox[s[c] & 0xff] | (uint32_t)t.sbox[(s[(c + 1) & 3] >> 8) & 0xff] << 8 |
                    (uint32_t)t.sbox[(s[(c + 2) & 3] >> 16) & 0xff] << 16 | (uint32_t)t.sbox[s[(c + 3) & 3] >> 24] << 24) ^ rk[40 + c];
        memcpy(out, u, 16);
    }
    // Matyas-Meyer-Oseas: AES_k(x) ^ x (common/src/prg.rs:311-331, 569-593)
    void mmo(const uint8_t in[16], uint8_t out[16]) const {
        uint8_t e[16];
        encrypt(in, e);
        for (int i = 0; i < 16; i++) out[i] = (uint8_t)(e[i] ^ in[i]);
    }
};

}  // namespace wh
