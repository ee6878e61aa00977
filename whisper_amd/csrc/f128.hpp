// Field128 arithmetic for gfx950 device code and for the host-side client/report generator.
//
// GF(p), p = 2^128 - 7*2^66 + 1 (reference: batched-flpcp-rs/src/fp.rs:391-421).  The reference
// keeps elements in Montgomery form; here an element IS its canonical residue (two 64-bit
// limbs, little endian = the wire encoding of field.rs:598-607), so report bytes are used as
// they arrive and results are written without conversion.  Reduction uses the shape of p:
//     2^128 = 28*2^64 - 1   (mod p)
//     2^192 = 783*2^64 - 28 (mod p)
// Same field, same elements, hence bit-identical canonical results.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define WH_HD __host__ __device__ __forceinline__
#else
#define WH_HD inline
#endif

namespace wh {

typedef unsigned __int128 u128;

struct F128 {
    uint64_t lo, hi;
};

constexpr uint64_t P_LO = 1ULL;
constexpr uint64_t P_HI = 0xffffffffffffffe4ULL;  // 2^64 - 28

WH_HD u128 f_p() { return ((u128)P_HI << 64) | P_LO; }
WH_HD u128 f_u(const F128 &a) { return ((u128)a.hi << 64) | a.lo; }
WH_HD F128 f_make(u128 v) { return F128{(uint64_t)v, (uint64_t)(v >> 64)}; }
WH_HD F128 f_zero() { return F128{0, 0}; }
WH_HD F128 f_one() { return F128{1, 0}; }
WH_HD F128 f_small(uint64_t v) { return F128{v, 0}; }
WH_HD bool f_is_zero(const F128 &a) { return (a.lo | a.hi) == 0; }
WH_HD bool f_eq(const F128 &a, const F128 &b) { return a.lo == b.lo && a.hi == b.hi; }
// canonical-range test used by decoders and by the rejection sampler (v < p)
WH_HD bool f_in_range(uint64_t lo, uint64_t hi) { return hi < P_HI || (hi == P_HI && lo < P_LO); }

WH_HD F128 f_add(const F128 &a, const F128 &b) {
    u128 x = f_u(a), y = f_u(b);
    u128 s = x + y;
    bool carry = s < x;
    if (carry || s >= f_p()) s -= f_p();
    return f_make(s);
}
WH_HD F128 f_sub(const F128 &a, const F128 &b) {
    u128 x = f_u(a), y = f_u(b);
    u128 d = x - y;
    if (x < y) d += f_p();
    return f_make(d);
}
WH_HD F128 f_neg(const F128 &a) { return f_is_zero(a) ? a : f_make(f_p() - f_u(a)); }

// reduce z3:z2:z1:z0 (256 bits) to the canonical residue
WH_HD F128 f_reduce256(uint64_t z0, uint64_t z1, uint64_t z2, uint64_t z3) {
    // value = (z1 + 28 z2 + 783 z3) 2^64 + (z0 - z2 - 28 z3)
    u128 A = (u128)z1 + (u128)z2 * 28u + (u128)z3 * 783u;  // < 2^74
    uint64_t a_lo = (uint64_t)A, a_hi = (uint64_t)(A >> 64);  // a_hi < 2^10
    // a_hi 2^128 = a_hi (28 2^64 - 1)
    u128 A1 = (u128)a_lo + (u128)a_hi * 28u;
    uint64_t c = (uint64_t)(A1 >> 64);  // 0 or 1
    uint64_t a2 = (uint64_t)A1 + c * 28u;  // no further carry: if c then (uint64)A1 < 2^15
    u128 neg = (u128)z2 + (u128)z3 * 28u + a_hi + c;  // < 2^69
    u128 S = ((u128)a2 << 64) + z0;
    if (S >= neg) S -= neg; else S = f_p() - (neg - S);
    if (S >= f_p()) S -= f_p();
    return f_make(S);
}

WH_HD F128 f_mul(const F128 &a, const F128 &b) {
    u128 p00 = (u128)a.lo * b.lo;
    u128 p01 = (u128)a.lo * b.hi;
    u128 p10 = (u128)a.hi * b.lo;
    u128 p11 = (u128)a.hi * b.hi;
    uint64_t z0 = (uint64_t)p00;
    u128 t = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
    uint64_t z1 = (uint64_t)t;
    t = (t >> 64) + (p01 >> 64) + (p10 >> 64) + (uint64_t)p11;
    uint64_t z2 = (uint64_t)t;
    uint64_t z3 = (uint64_t)((t >> 64) + (p11 >> 64));
    return f_reduce256(z0, z1, z2, z3);
}
WH_HD F128 f_sqr(const F128 &a) { return f_mul(a, a); }

// a * 2 (used by the bit-vector decoder: doubling needs no multiplier)
WH_HD F128 f_dbl(const F128 &a) { return f_add(a, a); }

WH_HD F128 f_pow(F128 x, u128 e) {
    F128 r = f_one();
    while (e) {
        if (e & 1) r = f_mul(r, x);
        x = f_sqr(x);
        e >>= 1;
    }
    return r;
}
// x^(2^k)
WH_HD F128 f_sqr_n(F128 x, unsigned k) {
    for (unsigned i = 0; i < k; i++) x = f_sqr(x);
    return x;
}
// x^-1 = x^(p-2); callers guarantee x != 0 where it matters (0 maps to 0)
WH_HD F128 f_inv(const F128 &x) {
    // p - 2 = 2^128 - 7*2^66 - 1: left-to-right square-and-multiply
    u128 e = f_p() - 2;
    F128 r = f_one();
    for (int i = 127; i >= 0; i--) {
        r = f_sqr(r);
        if ((e >> i) & 1) r = f_mul(r, x);
    }
    return r;
}

}  // namespace wh
