// SHAKE128 sponge for gfx950 device code (one sponge per lane, state in VGPRs) and host code.
//
// Reference behaviour being replaced: XofShake128 (batched-flpcp-rs/src/vdaf/xof.rs:197-218,
// sha3 0.10.8 crate) and the rejection sampler Prng::get (prng.rs:73-99).
//
// Layout decisions for CDNA4:
//  * the 25-lane state lives in 50 VGPRs per work-item; every state index is a compile-time
//    constant (runtime-indexed register arrays would go to scratch memory);
//  * the hot absorb path takes a whole 168-byte rate block from memory as twenty-one 8-byte
//    loads at any byte alignment (gfx950 global loads are byte-addressable), XORs it in with
//    static indices and permutes;
//  * partial blocks (prefixes, segment seams, tails) take a slow path that selects the state
//    word with a wave-uniform switch (the position is the same in every lane of a wave because
//    a launch only mixes reports of one shape).
#pragma once
#include "f128.hpp"

namespace wh {

constexpr unsigned SHAKE128_RATE = 168;
constexpr unsigned RATE_WORDS = 21;

#if defined(__HIPCC__)
__device__ __constant__ const uint64_t KECCAK_RC_DEV[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#endif
static const uint64_t KECCAK_RC_HOST[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

// 64-bit rotate by a compile-time amount.  On the device it is two v_alignbit_b32 (a 64-bit
// funnel over the register pair); left to itself hipcc emits 64-bit shifts + ORs.
WH_HD uint64_t rotl64(uint64_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32), nlo, nhi;
    if (n == 32) { nlo = hi; nhi = lo; }
    else if (n < 32) {
        nlo = __builtin_amdgcn_alignbit(lo, hi, 32 - n);
        nhi = __builtin_amdgcn_alignbit(hi, lo, 32 - n);
    } else {
        nlo = __builtin_amdgcn_alignbit(hi, lo, 64 - n);
        nhi = __builtin_amdgcn_alignbit(lo, hi, 64 - n);
    }
    return ((uint64_t)nhi << 32) | nlo;
#else
    return (x << n) | (x >> (64 - n));
#endif
}

// One Keccak-f[1600] round, all indices static.
#define WH_KECCAK_ROUND(a, rc)                                                                   \
    do {                                                                                         \
        uint64_t c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20];                                       \
        uint64_t c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21];                                       \
        uint64_t c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22];                                       \
        uint64_t c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23];                                       \
        uint64_t c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];                                       \
        uint64_t d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1);      \
        uint64_t d3 = c2 ^ rotl64(c4, 1), d4 = c3 ^ rotl64(c0, 1);                               \
        uint64_t b0 = a[0] ^ d0;                                                                 \
        uint64_t b1 = rotl64(a[6] ^ d1, 44), b2 = rotl64(a[12] ^ d2, 43);                        \
        uint64_t b3 = rotl64(a[18] ^ d3, 21), b4 = rotl64(a[24] ^ d4, 14);                       \
        uint64_t b5 = rotl64(a[3] ^ d3, 28), b6 = rotl64(a[9] ^ d4, 20);                         \
        uint64_t b7 = rotl64(a[10] ^ d0, 3), b8 = rotl64(a[16] ^ d1, 45);                        \
        uint64_t b9 = rotl64(a[22] ^ d2, 61);                                                    \
        uint64_t b10 = rotl64(a[1] ^ d1, 1), b11 = rotl64(a[7] ^ d2, 6);                         \
        uint64_t b12 = rotl64(a[13] ^ d3, 25), b13 = rotl64(a[19] ^ d4, 8);                      \
        uint64_t b14 = rotl64(a[20] ^ d0, 18);                                                   \
        uint64_t b15 = rotl64(a[4] ^ d4, 27), b16 = rotl64(a[5] ^ d0, 36);                       \
        uint64_t b17 = rotl64(a[11] ^ d1, 10), b18 = rotl64(a[17] ^ d2, 15);                     \
        uint64_t b19 = rotl64(a[23] ^ d3, 56);                                                   \
        uint64_t b20 = rotl64(a[2] ^ d2, 62), b21 = rotl64(a[8] ^ d3, 55);                       \
        uint64_t b22 = rotl64(a[14] ^ d4, 39), b23 = rotl64(a[15] ^ d0, 41);                     \
        uint64_t b24 = rotl64(a[21] ^ d1, 2);                                                    \
        a[0] = b0 ^ (~b1 & b2) ^ (rc);                                                           \
        a[1] = b1 ^ (~b2 & b3); a[2] = b2 ^ (~b3 & b4); a[3] = b3 ^ (~b4 & b0); a[4] = b4 ^ (~b0 & b1); \
        a[5] = b5 ^ (~b6 & b7); a[6] = b6 ^ (~b7 & b8); a[7] = b7 ^ (~b8 & b9); a[8] = b8 ^ (~b9 & b5); a[9] = b9 ^ (~b5 & b6); \
        a[10] = b10 ^ (~b11 & b12); a[11] = b11 ^ (~b12 & b13); a[12] = b12 ^ (~b13 & b14); a[13] = b13 ^ (~b14 & b10); a[14] = b14 ^ (~b10 & b11); \
        a[15] = b15 ^ (~b16 & b17); a[16] = b16 ^ (~b17 & b18); a[17] = b17 ^ (~b18 & b19); a[18] = b18 ^ (~b19 & b15); a[19] = b19 ^ (~b15 & b16); \
        a[20] = b20 ^ (~b21 & b22); a[21] = b21 ^ (~b22 & b23); a[22] = b22 ^ (~b23 & b24); a[23] = b23 ^ (~b24 & b20); a[24] = b24 ^ (~b20 & b21); \
    } while (0)

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 has a full-rate three-input bitwise instruction (v_bitop3_b32, truth table in the
// immediate: a = 0xf0, b = 0xcc, c = 0xaa).  Theta's five-way parities and its application
// become xor3 (0x96), chi becomes a ^ (~b & c) (0xd2): 122 full-rate + 58 half-rate (v_alignbit)
// instructions a round instead of 152 + 108.
__device__ __forceinline__ uint64_t xor3_64(uint64_t a, uint64_t b, uint64_t c) {
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0x96);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0x96);
    return lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t chi_64(uint64_t a, uint64_t b, uint64_t c) {
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0xd2);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0xd2);
    return lo | ((uint64_t)hi << 32);
}
// Issue model measured on gfx950 (tools/inst_rate_gen.py, profiles/r2_inst_rate.txt): a SIMD starts at most two VALU
// instructions per four cycles, from two different wavefronts.  The first slot goes to the highest-priority (then oldest)
// ready wavefront and takes any instruction; the second slot only takes a full-rate one (VOP1/VOP2 class and v_bitop3), never
// v_alignbit and the other "half-rate" VOP3 integer operations.  So a rotation only overlaps with another wavefront's
// bitwise work when the rotating wavefront is the one picked first: the round keeps its rotations in two runs (theta's five
// rotl 1, rho's 24) and raises the wavefront's priority to 3 for exactly those runs; outside them it runs at FPRIO (0, or the
// rank 0..2 a kernel gives its wavefronts: keccak_f1600_ranked).  Measured per round and SIMD with 3 / 4 / 8
// resident wavefronts: 706 / 679 / 646 cycles before, 482 / 462 / 428 with the hooks; the floor of this model is
// 180 instructions x 2 cycles = 360.
#define WH_KECCAK_PRIO_H __builtin_amdgcn_s_setprio(3)
#define WH_KECCAK_PRIO_F __builtin_amdgcn_s_setprio(FPRIO)
#define WH_SB_ __builtin_amdgcn_sched_barrier(0)
#define WH_KECCAK_ROUND_DEV(a, rc)                                                               \
    do {                                                                                         \
        const uint64_t c0 = xor3_64(xor3_64(a[0], a[5], a[10]), a[15], a[20]);                   \
        const uint64_t c1 = xor3_64(xor3_64(a[1], a[6], a[11]), a[16], a[21]);                   \
        const uint64_t c2 = xor3_64(xor3_64(a[2], a[7], a[12]), a[17], a[22]);                   \
        const uint64_t c3 = xor3_64(xor3_64(a[3], a[8], a[13]), a[18], a[23]);                   \
        const uint64_t c4 = xor3_64(xor3_64(a[4], a[9], a[14]), a[19], a[24]);                   \
        WH_SB_; WH_KECCAK_PRIO_H; WH_SB_;                                                        \
        const uint64_t r0 = rotl64(c0, 1), r1 = rotl64(c1, 1), r2 = rotl64(c2, 1);               \
        const uint64_t r3 = rotl64(c3, 1), r4 = rotl64(c4, 1);                                   \
        WH_SB_; WH_KECCAK_PRIO_F; WH_SB_;                                                        \
        /* a ^ d_x with d_x = c_(x-1) ^ rot(c_(x+1), 1), in one xor3; t_ in pi order */          \
        uint64_t t_[25];                                                                         \
        t_[0] = xor3_64(a[0], c4, r1); t_[1] = xor3_64(a[6], c0, r2); t_[2] = xor3_64(a[12], c1, r3);       \
        t_[3] = xor3_64(a[18], c2, r4); t_[4] = xor3_64(a[24], c3, r0); t_[5] = xor3_64(a[3], c2, r4);      \
        t_[6] = xor3_64(a[9], c3, r0); t_[7] = xor3_64(a[10], c4, r1); t_[8] = xor3_64(a[16], c0, r2);      \
        t_[9] = xor3_64(a[22], c1, r3); t_[10] = xor3_64(a[1], c0, r2); t_[11] = xor3_64(a[7], c1, r3);     \
        t_[12] = xor3_64(a[13], c2, r4); t_[13] = xor3_64(a[19], c3, r0); t_[14] = xor3_64(a[20], c4, r1);  \
        t_[15] = xor3_64(a[4], c3, r0); t_[16] = xor3_64(a[5], c4, r1); t_[17] = xor3_64(a[11], c0, r2);    \
        t_[18] = xor3_64(a[17], c1, r3); t_[19] = xor3_64(a[23], c2, r4); t_[20] = xor3_64(a[2], c1, r3);   \
        t_[21] = xor3_64(a[8], c2, r4); t_[22] = xor3_64(a[14], c3, r0); t_[23] = xor3_64(a[15], c4, r1);   \
        t_[24] = xor3_64(a[21], c0, r2);                                                         \
        WH_SB_; WH_KECCAK_PRIO_H; WH_SB_;                                                        \
        t_[1] = rotl64(t_[1], 44); t_[2] = rotl64(t_[2], 43); t_[3] = rotl64(t_[3], 21); t_[4] = rotl64(t_[4], 14);     \
        t_[5] = rotl64(t_[5], 28); t_[6] = rotl64(t_[6], 20); t_[7] = rotl64(t_[7], 3); t_[8] = rotl64(t_[8], 45);      \
        t_[9] = rotl64(t_[9], 61); t_[10] = rotl64(t_[10], 1); t_[11] = rotl64(t_[11], 6); t_[12] = rotl64(t_[12], 25); \
        t_[13] = rotl64(t_[13], 8); t_[14] = rotl64(t_[14], 18); t_[15] = rotl64(t_[15], 27); t_[16] = rotl64(t_[16], 36); \
        t_[17] = rotl64(t_[17], 10); t_[18] = rotl64(t_[18], 15); t_[19] = rotl64(t_[19], 56); t_[20] = rotl64(t_[20], 62); \
        t_[21] = rotl64(t_[21], 55); t_[22] = rotl64(t_[22], 39); t_[23] = rotl64(t_[23], 41); t_[24] = rotl64(t_[24], 2); \
        WH_SB_; WH_KECCAK_PRIO_F; WH_SB_;                                                        \
        a[0] = chi_64(t_[0], t_[1], t_[2]) ^ (rc);                                               \
        a[1] = chi_64(t_[1], t_[2], t_[3]); a[2] = chi_64(t_[2], t_[3], t_[4]); a[3] = chi_64(t_[3], t_[4], t_[0]); a[4] = chi_64(t_[4], t_[0], t_[1]);      \
        a[5] = chi_64(t_[5], t_[6], t_[7]); a[6] = chi_64(t_[6], t_[7], t_[8]); a[7] = chi_64(t_[7], t_[8], t_[9]); a[8] = chi_64(t_[8], t_[9], t_[5]); a[9] = chi_64(t_[9], t_[5], t_[6]); \
        a[10] = chi_64(t_[10], t_[11], t_[12]); a[11] = chi_64(t_[11], t_[12], t_[13]); a[12] = chi_64(t_[12], t_[13], t_[14]); a[13] = chi_64(t_[13], t_[14], t_[10]); a[14] = chi_64(t_[14], t_[10], t_[11]); \
        a[15] = chi_64(t_[15], t_[16], t_[17]); a[16] = chi_64(t_[16], t_[17], t_[18]); a[17] = chi_64(t_[17], t_[18], t_[19]); a[18] = chi_64(t_[18], t_[19], t_[15]); a[19] = chi_64(t_[19], t_[15], t_[16]); \
        a[20] = chi_64(t_[20], t_[21], t_[22]); a[21] = chi_64(t_[21], t_[22], t_[23]); a[22] = chi_64(t_[22], t_[23], t_[24]); a[23] = chi_64(t_[23], t_[24], t_[20]); a[24] = chi_64(t_[24], t_[20], t_[21]); \
    } while (0)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#ifndef WH_KECCAK_UNROLL
#define WH_KECCAK_UNROLL 1
#endif
template <int FPRIO>
__device__ __forceinline__ void keccak_f1600_p(uint64_t (&a)[25]) {
#pragma unroll WH_KECCAK_UNROLL
    for (int r = 0; r < 24; r++) WH_KECCAK_ROUND_DEV(a, KECCAK_RC_DEV[r]);
}
#endif
WH_HD void keccak_f1600(uint64_t (&a)[25]) {
#if defined(__HIP_DEVICE_COMPILE__)
    keccak_f1600_p<0>(a);
#else
    for (int r = 0; r < 24; r++) WH_KECCAK_ROUND(a, KECCAK_RC_HOST[r]);
#endif
}
// rank: wave-uniform, 0..2 -- the priority outside the rotation runs (three copies of the round loop, 1.5 KB each)
WH_HD void keccak_f1600_ranked(uint64_t (&a)[25], uint32_t rank) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (rank == 0) keccak_f1600_p<0>(a);
    else if (rank == 1) keccak_f1600_p<1>(a);
    else keccak_f1600_p<2>(a);
#else
    (void)rank;
    keccak_f1600(a);
#endif
}

#if defined(__HIPCC__)
// Equal shares of a SIMD's issue slots for long-lived sponge wavefronts.  A SIMD picks the oldest ready wavefront first: of
// the sponge wavefronts it holds, two run at nearly full speed and the others get what is left.  A wavefront can instead
// take its priority outside the rotation runs from the shader clock, which the wavefronts of a SIMD read alike: epochs of
// 2^WH_PRIO_EPOCH_LOG2 cycles, rank = (wave slot + epoch) mod 3, so that over three epochs each has had every rank.
#ifndef WH_PRIO_EPOCH_LOG2
#define WH_PRIO_EPOCH_LOG2 21
#endif
__device__ __forceinline__ uint32_t wave_slot() { return __builtin_amdgcn_s_getreg((3 << 11) | 4); }   // HW_ID.wave_id
__device__ __forceinline__ uint32_t rotate_prio(uint32_t slot) {
#ifndef WH_MEAS_NO_ROTATE
    const uint32_t epoch = (uint32_t)(__builtin_amdgcn_s_memtime() >> WH_PRIO_EPOCH_LOG2);
    return __builtin_amdgcn_readfirstlane((slot + epoch) % 3);
#else
    return 0;
#endif
}
#endif

// unaligned little-endian loads/stores (byte-addressable on gfx950 and on x86)
WH_HD uint64_t ld64(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
WH_HD F128 ld128(const uint8_t *p) { F128 v; __builtin_memcpy(&v, p, 16); return v; }
WH_HD void st128(uint8_t *p, const F128 &v) { __builtin_memcpy(p, &v, 16); }

// The same through a pointer KNOWN to be in global memory (report bytes, workspace rows).  A generic pointer makes
// flat_load / flat_store, which count in lgkmcnt as well as vmcnt: the scalar load of a round constant inside the
// permutation then waits (s_waitcnt lgkmcnt(0)) for every block load put in flight ahead of it and for the stores of the
// block before.  global_load / global_store count in vmcnt alone.
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint64_t wh_u64_u1 __attribute__((aligned(1)));
typedef uint32_t wh_u32x4 __attribute__((ext_vector_type(4)));
typedef wh_u32x4 wh_u32x4_u1 __attribute__((aligned(1)));
__device__ __forceinline__ uint64_t ld64g(const uint8_t *p) { return *(const __attribute__((address_space(1))) wh_u64_u1 *)p; }
__device__ __forceinline__ uint8_t ld8g(const uint8_t *p) { return *(const __attribute__((address_space(1))) uint8_t *)p; }
__device__ __forceinline__ F128 ld128g(const uint8_t *p) {
    const wh_u32x4 t = *(const __attribute__((address_space(1))) wh_u32x4_u1 *)p;
    return F128{t.x | ((uint64_t)t.y << 32), t.z | ((uint64_t)t.w << 32)};
}
// the same with the non-temporal hint: a stream that is read (written) once and must not push other lines out of L2
__device__ __forceinline__ F128 ld128g_nt(const uint8_t *p) {
    const wh_u32x4 t = __builtin_nontemporal_load((const __attribute__((address_space(1))) wh_u32x4_u1 *)p);
    return F128{t.x | ((uint64_t)t.y << 32), t.z | ((uint64_t)t.w << 32)};
}
__device__ __forceinline__ void st128g_nt(uint8_t *p, const F128 &v) {
    const wh_u32x4 t = {(uint32_t)v.lo, (uint32_t)(v.lo >> 32), (uint32_t)v.hi, (uint32_t)(v.hi >> 32)};
    __builtin_nontemporal_store(t, (__attribute__((address_space(1))) wh_u32x4_u1 *)p);
}
__device__ __forceinline__ void st128g(uint8_t *p, const F128 &v) {
    const wh_u32x4 t = {(uint32_t)v.lo, (uint32_t)(v.lo >> 32), (uint32_t)v.hi, (uint32_t)(v.hi >> 32)};
    *(__attribute__((address_space(1))) wh_u32x4_u1 *)p = t;
}
#else
WH_HD F128 ld128g_nt(const uint8_t *p) { return ld128(p); }
WH_HD void st128g_nt(uint8_t *p, const F128 &v) { st128(p, v); }
WH_HD uint64_t ld64g(const uint8_t *p) { return ld64(p); }
WH_HD uint8_t ld8g(const uint8_t *p) { return *p; }
WH_HD F128 ld128g(const uint8_t *p) { return ld128(p); }
WH_HD void st128g(uint8_t *p, const F128 &v) { st128(p, v); }
#endif

// G: the byte streams this sponge absorbs and emits live in global memory (device kernels over reports and workspace rows)
// PRIO (device): the wavefront's priority outside the rotation runs of the permutation -- 0..2 fixed, or -1: rotating
// with the shader clock (rotate_prio; `slot` must hold wave_slot())
template <bool G, int PRIO = 0>
struct SpongeT {
    uint64_t st[25];
    uint32_t pos;  // bytes absorbed into (or squeezed from) the current rate block
    uint32_t slot;

    WH_HD void permute() {
#if defined(__HIP_DEVICE_COMPILE__)
        if (PRIO < 0) keccak_f1600_ranked(st, rotate_prio(slot));
        else keccak_f1600_p<(PRIO < 0 ? 0 : PRIO)>(st);
#else
        keccak_f1600(st);
#endif
    }

    WH_HD void reset() {
#pragma unroll
        for (int i = 0; i < 25; i++) st[i] = 0;
        pos = 0;
    }

    // st[idx] ^= w with idx only known at run time.  On the device idx MUST be wave-uniform: the readfirstlane makes the
    // switch a scalar branch tree instead of 21 exec-masked bodies (which cost the chain kernel 70 spilled registers).
    // So every active lane of a wavefront has to absorb segments of the same LENGTHS (the pointers may differ): a sponge
    // whose segment lengths depend on the report -- the joint-rand seed over "the parts before the own one, the own part,
    // the parts after it" -- puts lanes of different aggregator ids out of phase, and this function then XORs one lane's
    // word index for all (round 2's un-batched path did, in batches that mixed ids).  The seed sponges therefore absorb
    // part by part, 16 bytes from a per-lane pointer (kernels.hip / chain.hip: seed_parts).
    WH_HD void xor_word(uint32_t idx, uint64_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
        idx = __builtin_amdgcn_readfirstlane(idx);
#endif
        switch (idx) {
#define C_(i) case i: st[i] ^= w; break;
            C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10)
            C_(11) C_(12) C_(13) C_(14) C_(15) C_(16) C_(17) C_(18) C_(19) C_(20)
#undef C_
        }
    }
    WH_HD uint64_t get_word(uint32_t idx) const {
#if defined(__HIP_DEVICE_COMPILE__)
        idx = __builtin_amdgcn_readfirstlane(idx);
#endif
        switch (idx) {
#define C_(i) case i: return st[i];
            C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10)
            C_(11) C_(12) C_(13) C_(14) C_(15) C_(16) C_(17) C_(18) C_(19) C_(20)
#undef C_
        }
        return 0;
    }

    WH_HD void absorb_byte(uint8_t b) {
        xor_word(pos >> 3, (uint64_t)b << (8 * (pos & 7)));
        if (++pos == SHAKE128_RATE) { permute(); pos = 0; }
    }

    // XofShake128::init (xof.rs:200-209): [len(dst)=8] || dst || seed, 25 bytes, written
    // straight into the zero state.  dst = {7, 0, id_be32, usage_be16} (vdaf.rs:339-346).
    // The seed is passed as its two little-endian words.
    WH_HD void init_xof(uint32_t alg_id, uint32_t usage, uint64_t s0, uint64_t s1) {
        reset();
        uint64_t w0 = 8ull | (7ull << 8) | (0ull << 16) | ((uint64_t)((alg_id >> 24) & 0xff) << 24) |
                      ((uint64_t)((alg_id >> 16) & 0xff) << 32) | ((uint64_t)((alg_id >> 8) & 0xff) << 40) |
                      ((uint64_t)(alg_id & 0xff) << 48) | ((uint64_t)((usage >> 8) & 0xff) << 56);
        st[0] = w0;
        st[1] = (uint64_t)(usage & 0xff) | (s0 << 8);
        st[2] = (s0 >> 56) | (s1 << 8);
        st[3] = s1 >> 56;
        pos = 25;
    }
    WH_HD void init_xof(uint32_t alg_id, uint32_t usage, const uint8_t *seed16) {
        init_xof(alg_id, usage, ld64(seed16), ld64(seed16 + 8));
    }

    // absorb the 8 bytes of w (little endian) at any position inside the block
    WH_HD void absorb_u64(uint64_t w) {
        if (pos + 8 <= SHAKE128_RATE) {
            uint32_t sh = (pos & 7) * 8, idx = pos >> 3;
            xor_word(idx, w << sh);
            if (sh) xor_word(idx + 1, w >> (64 - sh));
            pos += 8;
            if (pos == SHAKE128_RATE) { permute(); pos = 0; }
        } else {
#pragma unroll 1
            for (int i = 0; i < 8; i++) absorb_byte((uint8_t)(w >> (8 * i)));
        }
    }
    WH_HD void absorb_f(const F128 &v) { absorb_u64(v.lo); absorb_u64(v.hi); }

    // Absorb up to three byte segments (any alignment, any lengths) back to back.  The state
    // is permuted in exactly one place: steps only XOR and advance `pos`, and a full block
    // (pos == RATE) is permuted at the top of the loop.  Whole rate blocks take the static-index
    // path; the stretch before the first block boundary, seams and tails go word / byte wise.
    // PREFETCH: loads of the next whole block are issued before the permutation (42 more VGPRs).
    template <bool PREFETCH = true>
    WH_HD void absorb3(const uint8_t *p, uint32_t len, const uint8_t *p2, uint32_t len2, const uint8_t *p3,
                       uint32_t len3) {
        for (;;) {
            if (pos == SHAKE128_RATE) {
                // a whole block follows: put its loads in flight under the permutation
                const bool pre = PREFETCH && len >= SHAKE128_RATE;
                uint64_t w[RATE_WORDS];
                if (pre) {
#pragma unroll
                    for (int i = 0; i < (int)RATE_WORDS; i++) w[i] = G ? ld64g(p + 8 * i) : ld64(p + 8 * i);
                }
                permute();
                pos = 0;
                if (pre) {
#pragma unroll
                    for (int i = 0; i < (int)RATE_WORDS; i++) st[i] ^= w[i];
                    p += SHAKE128_RATE; len -= SHAKE128_RATE; pos = SHAKE128_RATE;
                    continue;
                }
            }
            if (!len) {
                if (!len2 && !len3) break;
                p = p2; len = len2; p2 = p3; len2 = len3; len3 = 0;
                continue;
            }
            if (pos == 0 && len >= SHAKE128_RATE) {
#pragma unroll
                for (int i = 0; i < (int)RATE_WORDS; i++) st[i] ^= G ? ld64g(p + 8 * i) : ld64(p + 8 * i);
                p += SHAKE128_RATE; len -= SHAKE128_RATE; pos = SHAKE128_RATE;
            } else if (len >= 8 && pos + 8 <= SHAKE128_RATE) {
                uint64_t w = G ? ld64g(p) : ld64(p);
                uint32_t sh = (pos & 7) * 8, idx = pos >> 3;
                xor_word(idx, w << sh);
                if (sh) xor_word(idx + 1, w >> (64 - sh));
                p += 8; len -= 8; pos += 8;
            } else {
                xor_word(pos >> 3, (uint64_t)(G ? ld8g(p) : *p) << (8 * (pos & 7)));
                p++; len--; pos++;
            }
        }
    }
    WH_HD void absorb(const uint8_t *p, uint32_t len) { absorb3(p, len, nullptr, 0, nullptr, 0); }

    // SHAKE128 padding (0x1f ... 0x80) and the permutation that yields the first output block
    WH_HD void finalize() {
        xor_word(pos >> 3, (uint64_t)0x1f << (8 * (pos & 7)));
        st[20] ^= 0x8000000000000000ULL;
        permute();
        pos = 0;
    }
    // Xof::into_seed (xof.rs:127-132): first 16 output bytes
    WH_HD F128 into_seed() { finalize(); return F128{st[0], st[1]}; }
};
typedef SpongeT<false> Sponge;
typedef SpongeT<true> SpongeG;

// Prng::get over a finalized sponge (prng.rs:73-99): consecutive 16-byte chunks of the output
// stream, little endian, chunks >= p skipped; writes `n` elements (16 B each) at `out`.
// Two rate blocks hold exactly 21 chunks, so the loop below is the whole stream walk.
// `bound` is p everywhere except under the test hook wh_*_set_test_sampler_bound.
template <bool G, int PRIO>
WH_HD void xof_emit(SpongeT<G, PRIO> &s, uint8_t *out, uint32_t n, const F128 &bound) {
    uint32_t left = n;
#define WH_EMIT(lo_, hi_)                                                                      \
    do {                                                                                         \
        uint64_t l_ = (lo_), h_ = (hi_);                                                         \
        if (left && (h_ < bound.hi || (h_ == bound.hi && l_ < bound.lo))) { if (G) st128g(out, F128{l_, h_}); else st128(out, F128{l_, h_}); out += 16; left--; } \
    } while (0)
    for (;;) {
        // even block: chunks (0,1) .. (18,19); word 20 is the low half of a straddling chunk
#pragma unroll
        for (int i = 0; i < 10; i++) WH_EMIT(s.st[2 * i], s.st[2 * i + 1]);
        if (!left) break;
        uint64_t carry = s.st[20];
        s.permute();
        WH_EMIT(carry, s.st[0]);
#pragma unroll
        for (int i = 0; i < 10; i++) WH_EMIT(s.st[2 * i + 1], s.st[2 * i + 2]);
        if (!left) break;
        s.permute();
    }
#undef WH_EMIT
}

}  // namespace wh
