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
    uint64_t lo = a.lo + b.lo;
    uint64_t c = lo < a.lo;
    uint64_t hi = a.hi + b.hi;
    uint64_t c1 = hi < a.hi;
    hi += c;
    uint64_t carry = c1 | (hi < c);
    // a + b < 2p, so one subtraction of p = (2^64 - 28) 2^64 + 1 is enough
    if (carry || hi > P_HI || (hi == P_HI && lo >= P_LO)) {
        uint64_t bw = lo == 0;
        lo -= 1;
        hi = hi + 28u - bw;
    }
    return F128{lo, hi};
}
WH_HD F128 f_sub(const F128 &a, const F128 &b) {
    uint64_t lo = a.lo - b.lo;
    uint64_t b1 = a.lo < b.lo;
    uint64_t hi = a.hi - b.hi;
    uint64_t b2 = a.hi < b.hi;
    uint64_t borrow = b2 | (hi < b1);
    hi -= b1;
    if (borrow) {
        lo += 1;
        hi = hi - 28u + (lo == 0);
    }
    return F128{lo, hi};
}
WH_HD F128 f_neg(const F128 &a) { return f_sub(F128{0, 0}, a); }

// Reduce a 256-bit value given as eight 32-bit limbs (little endian) to the canonical residue.
// Written on 32/64-bit integers only so that hipcc lowers it to v_mad_u64_u32 and 64-bit adds
// (no 128-bit temporaries, no carry chains through VCC).
//   value = (Z0 - Z2 - 28 Z3) + 2^64 (Z1 + 28 Z2 + 783 Z3),   Zk = z[2k] + 2^32 z[2k+1]
WH_HD F128 f_reduce_limbs(uint32_t z0, uint32_t z1, uint32_t z2, uint32_t z3, uint32_t z4, uint32_t z5,
                          uint32_t z6, uint32_t z7) {
    // A = Z1 + 28 Z2 + 783 Z3  -> limbs a0, a1 and a small top part a2 < 2^10
    uint64_t t = (uint64_t)z4 * 28u + z2;
    t = (uint64_t)z6 * 783u + t;
    uint32_t a0 = (uint32_t)t;
    t = (uint64_t)z5 * 28u + (t >> 32);
    t += z3;
    t = (uint64_t)z7 * 783u + t;
    uint32_t a1 = (uint32_t)t, a2 = (uint32_t)(t >> 32);
    // a2 2^128 = a2 (28 2^64 - 1): fold into A (may carry once more) and into the negative part
    uint64_t A = ((uint64_t)a1 << 32) | a0;
    uint64_t A1 = A + (uint64_t)a2 * 28u;
    uint32_t c2 = A1 < A;
    A1 += c2 * 28u;                         // no further carry: after a wrap A1 < 2^15
    // neg = Z2 + 28 Z3 + a2 + c2  (< 2^69): limbs n_lo (64 bits) and n2
    t = (uint64_t)z6 * 28u + z4;
    t += (uint64_t)a2 + c2;
    uint32_t n0 = (uint32_t)t;
    t = (uint64_t)z7 * 28u + (t >> 32);
    t += z5;
    uint64_t n_lo = ((uint64_t)(uint32_t)t << 32) | n0;
    uint64_t n2 = t >> 32;
    // R = (A1 2^64 + Z0) - neg  (mod 2^128), then one conditional +p or -p
    uint64_t s_lo = ((uint64_t)z1 << 32) | z0;
    uint64_t r_lo = s_lo - n_lo;
    uint64_t b1 = s_lo < n_lo;
    uint64_t sub_hi = n2 + b1;
    uint64_t r_hi = A1 - sub_hi;
    bool borrow = A1 < sub_hi;
    if (borrow) {
        // + p = + (2^128 - 28 2^64 + 1)
        uint64_t lo = r_lo + 1;
        uint64_t cy = lo == 0;
        r_hi = r_hi - 28u + cy;
        r_lo = lo;
    } else if (r_hi > P_HI || (r_hi == P_HI && r_lo >= P_LO)) {
        uint64_t bw = r_lo == 0;
        r_lo -= 1;
        r_hi = r_hi + 28u - bw;
    }
    return F128{r_lo, r_hi};
}
WH_HD F128 f_reduce256(uint64_t z0, uint64_t z1, uint64_t z2, uint64_t z3) {
    return f_reduce_limbs((uint32_t)z0, (uint32_t)(z0 >> 32), (uint32_t)z1, (uint32_t)(z1 >> 32), (uint32_t)z2,
                          (uint32_t)(z2 >> 32), (uint32_t)z3, (uint32_t)(z3 >> 32));
}

// Schoolbook 4 x 4 limbs, row by row: every step is t = a_i b_j + z + carry, which cannot
// overflow 64 bits ((2^32-1)^2 + 2 (2^32-1) = 2^64 - 1).
WH_HD F128 f_mul(const F128 &x, const F128 &y) {
    const uint32_t a0 = (uint32_t)x.lo, a1 = (uint32_t)(x.lo >> 32), a2 = (uint32_t)x.hi, a3 = (uint32_t)(x.hi >> 32);
    const uint32_t b0 = (uint32_t)y.lo, b1 = (uint32_t)(y.lo >> 32), b2 = (uint32_t)y.hi, b3 = (uint32_t)(y.hi >> 32);
    uint64_t t;
    uint32_t z0, z1, z2, z3, z4, z5, z6, z7;
    t = (uint64_t)a0 * b0;                         z0 = (uint32_t)t;
    t = (uint64_t)a0 * b1 + (t >> 32);             z1 = (uint32_t)t;
    t = (uint64_t)a0 * b2 + (t >> 32);             z2 = (uint32_t)t;
    t = (uint64_t)a0 * b3 + (t >> 32);             z3 = (uint32_t)t; z4 = (uint32_t)(t >> 32);
#define WH_ROW(ai, zA, zB, zC, zD, zE)                                                         \
    t = (uint64_t)ai * b0 + zA;                            zA = (uint32_t)t;                     \
    t = (uint64_t)ai * b1 + zB + (t >> 32);                zB = (uint32_t)t;                     \
    t = (uint64_t)ai * b2 + zC + (t >> 32);                zC = (uint32_t)t;                     \
    t = (uint64_t)ai * b3 + zD + (t >> 32);                zD = (uint32_t)t; zE = (uint32_t)(t >> 32);
    WH_ROW(a1, z1, z2, z3, z4, z5)
    WH_ROW(a2, z2, z3, z4, z5, z6)
    WH_ROW(a3, z3, z4, z5, z6, z7)
#undef WH_ROW
    return f_reduce_limbs(z0, z1, z2, z3, z4, z5, z6, z7);
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
    // p - 2 = 2^128 - 7*2^66 - 1 = (2^59 - 1) 2^69 + (2^66 - 1): two runs of ones, each by the doubling chain
    // x^(2^(a+b) - 1) = (x^(2^a - 1))^(2^b) x^(2^b - 1); 134 squarings and 11 multiplications
    const F128 t2 = f_mul(f_sqr(x), x);                  // 2^2 - 1
    const F128 t3 = f_mul(f_sqr(t2), x);                 // 2^3 - 1
    const F128 t6 = f_mul(f_sqr_n(t3, 3), t3);
    const F128 t7 = f_mul(f_sqr(t6), x);
    const F128 t14 = f_mul(f_sqr_n(t7, 7), t7);
    const F128 t28 = f_mul(f_sqr_n(t14, 14), t14);
    const F128 t29 = f_mul(f_sqr(t28), x);
    const F128 t58 = f_mul(f_sqr_n(t29, 29), t29);
    const F128 t59 = f_mul(f_sqr(t58), x);               // x^(2^59 - 1)
    const F128 t66 = f_mul(f_sqr_n(t59, 7), t7);         // x^(2^66 - 1)
    return f_mul(f_sqr_n(t59, 69), t66);
}

// ---- lazy dot products ------------------------------------------------------------------
// sum_t x_t * w_t without reducing per term: both operands are cut into five 26-bit limbs and
// the 25 partial products of a term are added into nine 64-bit column sums (one
// v_mad_u64_u32 each, no carries).  A product is < 2^52 and a column receives at most five per
// term, so LAZY_MAX_TERMS terms fit in 64 bits; f_lazy_reduce then yields the same canonical
// element as the term-by-term f_add(f_mul()) chain.
constexpr uint32_t M26 = (1u << 26) - 1;
constexpr uint32_t LAZY_MAX_TERMS = 512;
struct Lazy {
    uint64_t c[9];
};
WH_HD void f_split26(const F128 &x, uint32_t (&o)[5]) {
    o[0] = (uint32_t)x.lo & M26;
    o[1] = (uint32_t)(x.lo >> 26) & M26;
    o[2] = (uint32_t)((x.hi << 12) | (x.lo >> 52)) & M26;
    o[3] = (uint32_t)(x.hi >> 14) & M26;
    o[4] = (uint32_t)(x.hi >> 40);
}
WH_HD void f_lazy_zero(Lazy &a) {
#pragma unroll
    for (int k = 0; k < 9; k++) a.c[k] = 0;
}
// acc += a * b as one v_mad_u64_u32.  Written as an asm statement because the optimiser
// otherwise sums the products of a column first and adds the running sum with a separate
// 64-bit add (one extra instruction per column and term).
WH_HD void mad_acc(uint64_t &acc, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t carry_out;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry_out) : "v"(a), "v"(b));
#else
    acc += (uint64_t)a * b;
#endif
}
WH_HD void f_lazy_mac(Lazy &a, const uint32_t (&x)[5], const uint32_t (&w)[5]) {
#if defined(__HIP_DEVICE_COMPILE__)
    // all 25 products in ONE asm statement: hipcc puts an s_nop between two asm statements that follow each other
    // directly (175 of them per four elements in the query kernel's loop), none inside a statement.  Consecutive
    // instructions never touch the same column.
    asm("v_mad_u64_u32 %0, vcc, %9, %14, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %9, %15, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %9, %16, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %9, %17, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %9, %18, %4\n\t"
        "v_mad_u64_u32 %1, vcc, %10, %14, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %10, %15, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %10, %16, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %10, %17, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %10, %18, %5\n\t"
        "v_mad_u64_u32 %2, vcc, %11, %14, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %11, %15, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %11, %16, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %11, %17, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %11, %18, %6\n\t"
        "v_mad_u64_u32 %3, vcc, %12, %14, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %12, %15, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %12, %16, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %12, %17, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %12, %18, %7\n\t"
        "v_mad_u64_u32 %4, vcc, %13, %14, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %13, %15, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %13, %16, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %13, %17, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %13, %18, %8"
        : "+v"(a.c[0]), "+v"(a.c[1]), "+v"(a.c[2]), "+v"(a.c[3]), "+v"(a.c[4]), "+v"(a.c[5]), "+v"(a.c[6]), "+v"(a.c[7]), "+v"(a.c[8])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4])
        : "vcc");
#else
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) mad_acc(a.c[i + j], x[i], w[j]);
#endif
}
WH_HD F128 f_lazy_reduce(const Lazy &a) {
    // carry the column sums into 26-bit digits l[0..10] (value < 2^271)
    uint64_t l[11];
    uint64_t t = a.c[0];
    l[0] = t & M26;
#pragma unroll
    for (int k = 1; k < 9; k++) {
        t = a.c[k] + (t >> 26);
        l[k] = t & M26;
    }
    t >>= 26;
    l[9] = t & M26;
    l[10] = t >> 26;
    // repack at 64-bit boundaries: digit k starts at bit 26 k
    const uint64_t w0 = l[0] | (l[1] << 26) | (l[2] << 52);
    const uint64_t w1 = (l[2] >> 12) | (l[3] << 14) | (l[4] << 40);
    const uint64_t w2 = (l[4] >> 24) | (l[5] << 2) | (l[6] << 28) | (l[7] << 54);
    const uint64_t w3 = (l[7] >> 10) | (l[8] << 16) | (l[9] << 42);
    const uint64_t w4 = (l[9] >> 22) | (l[10] << 4);
    // 2^256 = 21896 * 2^64 - 783 (mod p)
    const F128 c256 = F128{0xfffffffffffffcf1ULL, 0x5587ULL};
    return f_add(f_reduce256(w0, w1, w2, w3), f_mul(F128{w4, 0}, c256));
}
// sum of 128-bit values x_t * s_t with small multipliers (s_t < 2^31, sum of s_t < 2^32):
// four 64-bit column sums over the 32-bit words of x
struct LazySmall {
    uint64_t c[4];
};
WH_HD void f_small_zero(LazySmall &a) { a.c[0] = a.c[1] = a.c[2] = a.c[3] = 0; }
WH_HD void f_small_mac(LazySmall &a, const F128 &x, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WH_SMALL_MAC_PLAIN)
    // one multiply-add per word: left to itself hipcc turns a power-of-two s into a 64-bit shift and a 64-bit add (two
    // first-slot instructions instead of one)
    asm("v_mad_u64_u32 %0, vcc, %4, %8, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %5, %8, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %6, %8, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %7, %8, %3"
        : "+v"(a.c[0]), "+v"(a.c[1]), "+v"(a.c[2]), "+v"(a.c[3])
        : "v"((uint32_t)x.lo), "v"((uint32_t)(x.lo >> 32)), "v"((uint32_t)x.hi), "v"((uint32_t)(x.hi >> 32)), "v"(s)
        : "vcc");
    return;
#endif
    a.c[0] += (uint64_t)(uint32_t)x.lo * s;
    a.c[1] += (uint64_t)(uint32_t)(x.lo >> 32) * s;
    a.c[2] += (uint64_t)(uint32_t)x.hi * s;
    a.c[3] += (uint64_t)(uint32_t)(x.hi >> 32) * s;
}
WH_HD F128 f_small_reduce(const LazySmall &a) {
    uint64_t t = a.c[0];
    const uint32_t z0 = (uint32_t)t;
    t = a.c[1] + (t >> 32);
    const uint32_t z1 = (uint32_t)t;
    t = a.c[2] + (t >> 32);
    const uint32_t z2 = (uint32_t)t;
    t = a.c[3] + (t >> 32);
    return f_reduce_limbs(z0, z1, z2, (uint32_t)t, (uint32_t)(t >> 32), 0, 0, 0);
}

}  // namespace wh
