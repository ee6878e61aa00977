// Field64 arithmetic for gfx950 device code: GF(p), p = 2^64 - 2^32 + 1 (reference:
// batched-flpcp-rs/src/fp.rs:359-389, FP64; generator of the 2^32 subgroup :366).  As for Field128
// (f128.hpp) an element is its canonical residue, which is also its 8-byte little-endian wire
// encoding (field.rs:598-607); the reference's Montgomery form (R = 2^128, fp.rs:84-213) computes the
// same field elements.  Reduction uses 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).
//
// No server of the reference instantiates Field64 (bin-utils/src/lib.rs:27: every Prio3 type is a
// Field128 one); this is the arithmetic layer only, exercised by wh_test_f64_op.
#pragma once
#include <stdint.h>
#include "f128.hpp"

namespace wh {

constexpr uint64_t P64 = 0xffffffff00000001ULL;
constexpr uint64_t EPS64 = 0xffffffffULL;        // 2^32 - 1 = 2^64 mod p

WH_HD uint64_t g_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    // a + b < 2p: one conditional subtraction, also when the 64-bit sum wrapped
    if (s < a || s >= P64) s -= P64;
    return s;
}
WH_HD uint64_t g_sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    if (a < b) d += P64;
    return d;
}
// (hi 2^64 + lo) mod p
WH_HD uint64_t g_reduce128(uint64_t lo, uint64_t hi) {
    const uint64_t hi_hi = hi >> 32, hi_lo = hi & EPS64;
    uint64_t t0 = lo - hi_hi;                     // - hi_hi 2^96
    if (lo < hi_hi) t0 -= EPS64;                  // wrapped: + 2^64 = + (2^32 - 1) too much, i.e. subtract it
    const uint64_t t1 = hi_lo * EPS64;            // hi_lo 2^64 = hi_lo (2^32 - 1), < 2^64
    uint64_t t2 = t0 + t1;
    if (t2 < t1) t2 += EPS64;                     // wrapped: - 2^64 = - (2^32 - 1)
    if (t2 >= P64) t2 -= P64;
    return t2;
}
WH_HD uint64_t g_mul(uint64_t a, uint64_t b) {
    const u128 z = (u128)a * b;
    return g_reduce128((uint64_t)z, (uint64_t)(z >> 64));
}
WH_HD uint64_t g_pow(uint64_t x, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = g_mul(r, x);
        x = g_mul(x, x);
        e >>= 1;
    }
    return r;
}
WH_HD uint64_t g_inv(uint64_t x) { return g_pow(x, P64 - 2); }     // 0 maps to 0, as FieldParameters::inv (fp.rs:196-213)

}  // namespace wh
