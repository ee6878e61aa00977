// CPU client for the heavy-hitter path: makes the key chains the GPU path consumes.
//
// Mirrors BCSketch::gen_key_chain (hhcore/src/protocol.rs:31-76): VDPFKey::gen (hhcore/src/vdpf.rs:54-113) with
// gen_correction / gen_output_correction (hhcore/src/dpf.rs:85-145) for the main key over IntModN<2^depth> with a
// Bucket<u32> payload, then one single-level key per data bit and one for the sign, payload 1; the result is laid
// out as bincode 1.3.3 writes a KeyChain<u32> (serialize/src/lib.rs:72-80).  thread_rng (prg.rs:249-254) is
// replaced by a SHAKE128 stream over a caller-supplied seed.  Nothing here touches oracle/.
#include <cstring>
#include <vector>
#include "../../include/whisper_gfx950.h"
#include "aes_host.hpp"
#include "hh_dev.hpp"
#include "keccak.hpp"

namespace wh {
namespace {

typedef uint8_t Seed[16];

struct Stream {  // SHAKE128 output stream
    Sponge s; uint32_t at = 0;
    explicit Stream(const uint8_t seed[32]) { s.reset(); s.absorb(seed, 32); s.finalize(); }
    void take(uint8_t *out, uint32_t n) {
        for (uint32_t i = 0; i < n; i++) {
            if (at == SHAKE128_RATE) { keccak_f1600(s.st); at = 0; }
            out[i] = (uint8_t)(s.st[at >> 3] >> (8 * (at & 7)));
            at++;
        }
    }
};

struct Prg {
    AesKey k0;
    Prg() : k0(zero16()) {}
    static const uint8_t *zero16() { static const uint8_t z[16] = {0}; return z; }
    static void bump(uint8_t c[16], uint64_t by) {  // inc_be: upper 64-bit lane (common/src/prg.rs:449-452)
        uint64_t hi;
        memcpy(&hi, c + 8, 8);
        hi += by;
        memcpy(c + 8, &hi, 8);
    }
    // expand_direction(true, true) (prg.rs:133-174)
    void expand(const Seed s, Seed left, Seed right, bool bits[2]) const {
        Seed c;
        memcpy(c, s, 16);
        bits[0] = !(c[0] & 1); bits[1] = !(c[0] & 2);
        c[0] &= 0xfc;
        k0.mmo(c, left);
        bump(c, 1);
        k0.mmo(c, right);
    }
    // convert::<T> (prg.rs:180-198): `words` u32 values from the counter stream that starts at the seed
    void convert(const Seed s, uint32_t *out, uint32_t words) const {
        Seed c, blk;
        memcpy(c, s, 16);
        for (uint32_t i = 0; i < words; i += 4) {
            k0.mmo(c, blk);
            bump(c, 1);
            memcpy(out + i, blk, 4 * (words - i < 4 ? words - i : 4));
        }
    }
    // mmo_hash2to4 (prg.rs:202-241)
    void hash2to4(const Seed s, const uint8_t x[16], uint8_t out[64]) const {
        k0.mmo(x, out); k0.mmo(s, out + 16); k0.mmo(out, out + 32); k0.mmo(out + 16, out + 48);
    }
};

struct Key {
    uint8_t key_id; Seed root;
    std::vector<uint8_t> cws;  // 18 bytes per level: seed, bit, bit
    uint8_t cs[64];
    std::vector<uint32_t> oc;
};

void gen(const Prg &g, uint32_t depth, uint32_t alpha, const uint8_t x[16], const uint32_t *beta, uint32_t words, Stream &rng, Key out[2]) {
    std::vector<uint32_t> conv0(words), conv1(words);
    for (;;) {
        Seed roots[2], cur[2];
        rng.take(roots[0], 16); rng.take(roots[1], 16);
        memcpy(cur, roots, 32);
        bool t[2] = {false, true};
        std::vector<uint8_t> cws(18 * depth);
        for (uint32_t l = 0; l < depth; l++) {
            const bool a = (alpha >> (depth - 1 - l)) & 1;  // MSB first (group.rs:492-500)
            Seed kids[2][2];
            bool kb[2][2];
            for (int p = 0; p < 2; p++) g.expand(cur[p], kids[p][0], kids[p][1], kb[p]);
            uint8_t *cw = &cws[18 * l];
            for (int i = 0; i < 16; i++) cw[i] = kids[0][!a][i] ^ kids[1][!a][i];  // the branch off the special path cancels
            cw[16] = (uint8_t)(kb[0][0] ^ kb[1][0] ^ a ^ 1);
            cw[17] = (uint8_t)(kb[0][1] ^ kb[1][1] ^ a);
            for (int p = 0; p < 2; p++) {
                bool nt = kb[p][a];
                memcpy(cur[p], kids[p][a], 16);
                if (t[p]) { for (int i = 0; i < 16; i++) cur[p][i] ^= cw[i]; nt ^= cw[16 + a] != 0; }
                t[p] = nt;
            }
        }
        uint8_t h0[64], h1[64];
        g.hash2to4(cur[0], x, h0); g.hash2to4(cur[1], x, h1);
        const bool b0 = cur[0][0] & 1, b1 = cur[1][0] & 1;
        if (b0 == b1) continue;  // vdpf.rs:86-91
        g.convert(cur[0], conv0.data(), words); g.convert(cur[1], conv1.data(), words);
        for (int p = 0; p < 2; p++) {
            out[p].key_id = (uint8_t)p;
            memcpy(out[p].root, roots[p], 16);
            out[p].cws = cws;
            for (int i = 0; i < 64; i++) out[p].cs[i] = h0[i] ^ h1[i];
            out[p].oc.resize(words);
            for (uint32_t i = 0; i < words; i++) {
                const uint32_t v = beta[i] - conv0[i] + conv1[i];
                out[p].oc[i] = b1 ? 0u - v : v;
            }
        }
        return;
    }
}

uint8_t *put_u64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); return p + 8; }
uint8_t *put_key(uint8_t *p, const Key &k, bool bucket) {
    *p++ = k.key_id;
    memcpy(p, k.root, 16); p += 16;
    p = put_u64(p, k.cws.size() / 18);
    memcpy(p, k.cws.data(), k.cws.size()); p += k.cws.size();
    memcpy(p, k.cs, 64); p += 64;
    if (bucket) p = put_u64(p, 256);  // Bucket.data is a Vec; sign_data and ctr follow in line
    memcpy(p, k.oc.data(), 4 * k.oc.size());
    return p + 4 * k.oc.size();
}

}  // namespace
}  // namespace wh

extern "C" int32_t wh_hh_gen_key_chain(uint32_t depth, const uint8_t id_le[16], uint32_t alpha, const uint32_t bucket[258],
                                       const uint8_t rng_seed[32], uint8_t *kc0, uint8_t *kc1) {
    using namespace wh;
    if (!id_le || !bucket || !rng_seed || !kc0 || !kc1 || depth < 1 || depth > 16 || (alpha >> depth)) return WH_ERR_PARAM;
    for (uint32_t i = 0; i < HH_SUPPORT; i++)
        if (bucket[i] != 1u && bucket[i] != 0xffffffffu) return WH_ERR_PARAM;  // "Data is not -1 or 1" (protocol.rs:46-47)
    const Prg g;
    Stream rng(rng_seed);
    Key pair[2];
    uint8_t x[16] = {0};
    x[0] = (uint8_t)alpha; x[1] = (uint8_t)(alpha >> 8);  // IntModN::as_bytes (group.rs:502-509)
    gen(g, depth, alpha, x, bucket, HH_BKT, rng, pair);
    uint8_t *p[2] = {kc0, kc1};
    for (int s = 0; s < 2; s++) {
        memcpy(p[s], id_le, 16);
        p[s] = put_key(p[s] + 16, pair[s], true);
        p[s] = put_u64(p[s], HH_SUPPORT);
    }
    for (uint32_t i = 0; i < HH_SUPPORT; i++) {
        const uint32_t point = bucket[i] == 1u, one = 1;
        uint8_t xb[16] = {0};
        xb[0] = (uint8_t)point;  // bool::as_bytes (common/src/lib.rs:163-165)
        gen(g, 1, point, xb, &one, 1, rng, pair);
        for (int s = 0; s < 2; s++) p[s] = put_key(p[s], pair[s], false);
    }
    return WH_OK;
}
