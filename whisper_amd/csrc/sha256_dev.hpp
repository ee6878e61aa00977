// SHA-256 compression for device code: the heavy-hitter proof strings (vdpf.hip) and the group test's batch proofs over
// string tags (grouptest.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace wh {

__device__ __forceinline__ uint32_t sha_rotl(uint32_t v, int n) { return __builtin_rotateleft32(v, n); }
__device__ __forceinline__ uint32_t sha_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// ---- SHA-256 (FIPS 180-4); `sha256 = "1.1.2"` digest() in the reference (vdpf.rs:211, protocol.rs:163-175) ----
__device__ __forceinline__ void sha_init(uint32_t (&h)[8]) {
    h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a;
    h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
}
__device__ __forceinline__ void sha_compress(uint32_t (&h)[8], uint32_t (&w)[16]) {
    constexpr uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    // sigma functions as one three-input XOR of rotates, Ch = bitop3 0xca (e ? f : g), Maj = bitop3 0xe8: 15 instructions a
    // round and 10 per schedule word -- a lone wavefront issues one every ~5 cycles, so the count is the latency
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const uint32_t x = w[(i + 1) & 15], y = w[(i + 14) & 15];
            w[i & 15] += sha_xor3(sha_rotl(x, 25), sha_rotl(x, 14), x >> 3) + sha_xor3(sha_rotl(y, 15), sha_rotl(y, 13), y >> 10) + w[(i + 9) & 15];
        }
        const uint32_t t1 = hh + sha_xor3(sha_rotl(e, 26), sha_rotl(e, 21), sha_rotl(e, 7)) + __builtin_amdgcn_bitop3_b32(e, f, g, 0xca) + K[i] + w[i & 15];
        const uint32_t t2 = sha_xor3(sha_rotl(a, 30), sha_rotl(a, 19), sha_rotl(a, 10)) + __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8);
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

}  // namespace wh
