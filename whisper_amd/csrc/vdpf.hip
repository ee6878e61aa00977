// Heavy-hitter key chains on gfx950: BCSketch::eval_and_check (hhcore/src/protocol.rs:78-176) for a batch of
// bincode-encoded KeyChain<u32>, i.e. VDPFKey::eval_all (hhcore/src/vdpf.rs:123-214) of the main key and of the
// 257 one-level support keys, the consistency folding, and the SHA-256 proof string.
//
//   k_hh_main     one workgroup per key chain: tree expansion (PrgSeed::expand_direction, common/src/prg.rs:133-174),
//                 leaf hashes (mmo_hash2to4, :202-241), leaf conversion into Bucket<u32> (convert -> refill8,
//                 :180-198, 368-403) written to `dat`, the per-position sum over all leaves, the PRF testing id
//   k_hh_support  one lane per support key
//   k_hh_chain    sixteen lanes per key chain: pi <- pi ^ H'(pi ^ corrected leaf hash), leaf by leaf (vdpf.rs:197-200;
//                 mmo_hash4to4, prg.rs:599-631), then SHA-256
//   k_hh_final    one lane per key chain: folding, the two vector digests and the digest of the 16640-character
//                 proof string (protocol.rs:121-175)
//
// AES-128 has no instruction on CDNA4: one 256-entry table (Te0) sits in LDS 32 times over, word index
// entry*32 + (lane & 31), so that the 32 lanes of a ds_read_b32 group always hit 32 different banks; the other
// three tables are rotations.  160 table reads + 16 for the last round per block: the kernels are bound by LDS
// reads (128 B/clk/CU), not by HBM.
#include "hh_dev.hpp"
#include "sha256_dev.hpp"

namespace wh {
namespace {

constexpr int32_t HH_ERR_KEY_FORMAT = -9;  // WH_ERR_KEY_FORMAT
constexpr uint32_t T_WORDS = 256 * 32;
constexpr uint32_t HDR_BYTES = 1472;       // header of a key chain up to the support keys, depth <= 11

struct B128 { uint32_t w[4]; };

__device__ __forceinline__ uint32_t rotl(uint32_t v, int n) { return __builtin_rotateleft32(v, n); }

// AES-128 with NT tables in LDS.  NT = 1: Te0, three rotates per column.  NT = 2: Te0 and Te0 <<< 16, one rotate per
// column.  NT = 4: Te0 <<< 8t, none.  rk = 44 round-key words (uniform).
//
// Measured on MI355X (tools/aes_microbench.hip, profiles/r1_aes_microbench.txt): a wavefront's ds_read_b32 and its VALU
// instructions do not overlap -- every extra VALU instruction per table read costs its full issue time -- so the cost
// of a table read is the issue time of the instructions that form its address plus the read itself: v_bfe + v_lshl_add
// 4.2 + 4.2 cycles, SDWA shift + add 4.2 + 2.3, v_perm_b32 4.2; the read 2.6; a three-input XOR 2.3.
//   NT = 1 (small kernels): 32 KiB, word index entry*32 + lane%32, address = SDWA byte shift + add.
//   NT = 2, 4: an entry row is 256 bytes -- table (t & 1) in its lower / upper 128 bytes, 32 lane copies each -- and
//   tables 2, 3 sit 64 KiB higher, so the whole address is ONE v_perm_b32: byte 0 = this lane's offset, byte 1 = the
//   state byte, byte 2 = the 64 KiB region, and the half-row is the instruction's immediate offset.  The tables must
//   start at LDS address 0 (the kernel has no other static LDS).
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
struct TabRef { uint32_t a, b; };  // NT = 1: byte address of this lane's copy of entry 0.  NT = 2, 4: v_perm source words
__device__ __forceinline__ TabRef tab_ref(const uint32_t *tables, uint32_t tid) {
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t *)tables + 4 * (tid & 31);
    return TabRef{base, base | 0x10000u};
}
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
template <int NT> __device__ __forceinline__ uint32_t te(const TabRef &R, int t, uint32_t x, int sh) {
    static_assert(NT == 1 || NT == 2 || NT == 4, "table count");
    if (NT == 1) {
        // byte select and shift in one (half-rate) SDWA instruction; v_bfe_u32 + v_lshl_add_u32 would be two
        uint32_t idx;
        if (sh == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(idx) : "v"(7u), "v"(x));
        else if (sh == 8) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(idx) : "v"(7u), "v"(x));
        else if (sh == 16) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(idx) : "v"(7u), "v"(x));
        else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(idx) : "v"(7u), "v"(x));
        return *(lds_cu32 *)(uintptr_t)(R.a + idx);
    }
    // selector bytes, low to high: lane word byte 0, state byte sh/8 (source 0 = bytes 4..7), lane word byte 2, zero
    const uint32_t sel = 0x0c020000u | ((4u + sh / 8) << 8);
    const int tt = NT == 4 ? t : t >> 1;   // NT = 2 holds tables 0 and 2 in the two half-rows of region 0
    const uint32_t addr = __builtin_amdgcn_perm(x, tt < 2 ? R.a : R.b, sel);
    return *(lds_cu32 *)(uintptr_t)(addr + (tt & 1) * 128);
}
template <int NT> __device__ __forceinline__ uint32_t aes_col(const TabRef &R, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t k) {
    if (NT == 4) return xor3(xor3(te<4>(R, 0, a, 0), te<4>(R, 1, b, 8), te<4>(R, 2, c, 16)), te<4>(R, 3, d, 24), k);
    if (NT == 2) return xor3(te<2>(R, 0, a, 0), te<2>(R, 2, c, 16), k) ^ rotl(te<2>(R, 0, b, 8) ^ te<2>(R, 2, d, 24), 8);
    return xor3(xor3(te<1>(R, 0, a, 0), rotl(te<1>(R, 0, b, 8), 8), rotl(te<1>(R, 0, c, 16), 16)), rotl(te<1>(R, 0, d, 24), 24), k);
}
// last round: SubBytes + ShiftRows only.  S[x] sits in bytes 1, 2 of table 0; 2, 3 of table 1; 0, 3 of table 2; 0, 1 of table 3
template <int NT> __device__ __forceinline__ uint32_t aes_last(const TabRef &R, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t k) {
    if (NT == 4) return ((te<4>(R, 2, a, 0) & 0xff) | (te<4>(R, 3, b, 8) & 0xff00) | (te<4>(R, 0, c, 16) & 0xff0000) | (te<4>(R, 1, d, 24) & 0xff000000)) ^ k;
    if (NT == 2) return ((te<2>(R, 2, a, 0) & 0xff) | (te<2>(R, 0, b, 8) & 0xff00) | (te<2>(R, 0, c, 16) & 0xff0000) | (te<2>(R, 2, d, 24) & 0xff000000)) ^ k;
    return (((te<1>(R, 0, a, 0) >> 8) & 0xff) | (te<1>(R, 0, b, 8) & 0xff00) | (te<1>(R, 0, c, 16) & 0xff0000) | ((te<1>(R, 0, d, 24) << 8) & 0xff000000)) ^ k;
}
template <int NT = 1> __device__ __forceinline__ B128 aes_enc(const TabRef &R, const uint32_t *__restrict__ rk, const B128 &in) {
    uint32_t s0 = in.w[0] ^ rk[0], s1 = in.w[1] ^ rk[1], s2 = in.w[2] ^ rk[2], s3 = in.w[3] ^ rk[3];
#pragma unroll
    for (int r = 1; r < 10; r++) {
        const uint32_t t0 = aes_col<NT>(R, s0, s1, s2, s3, rk[4 * r]);
        const uint32_t t1 = aes_col<NT>(R, s1, s2, s3, s0, rk[4 * r + 1]);
        const uint32_t t2 = aes_col<NT>(R, s2, s3, s0, s1, rk[4 * r + 2]);
        const uint32_t t3 = aes_col<NT>(R, s3, s0, s1, s2, rk[4 * r + 3]);
        s0 = t0; s1 = t1; s2 = t2; s3 = t3;
    }
    B128 o;
    o.w[0] = aes_last<NT>(R, s0, s1, s2, s3, rk[40]);
    o.w[1] = aes_last<NT>(R, s1, s2, s3, s0, rk[41]);
    o.w[2] = aes_last<NT>(R, s2, s3, s0, s1, rk[42]);
    o.w[3] = aes_last<NT>(R, s3, s0, s1, s2, rk[43]);
    return o;
}
// Matyas-Meyer-Oseas (prg.rs:311-331)
template <int NT = 1> __device__ __forceinline__ B128 mmo(const TabRef &T, const uint32_t *__restrict__ rk, const B128 &in) {
    B128 o = aes_enc<NT>(T, rk, in);
    o.w[0] ^= in.w[0]; o.w[1] ^= in.w[1]; o.w[2] ^= in.w[2]; o.w[3] ^= in.w[3];
    return o;
}
// the sixteen table reads of one block's round, in the order aes_col consumes them, and the four columns from them
template <int NT> __device__ __forceinline__ void aes_round_reads(const TabRef &R, const uint32_t (&s)[4], uint32_t (&t)[16]) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        t[4 * c] = te<NT>(R, 0, s[c], 0);
        t[4 * c + 1] = te<NT>(R, NT == 2 ? 0 : 1, s[(c + 1) & 3], 8);
        t[4 * c + 2] = te<NT>(R, 2, s[(c + 2) & 3], 16);
        t[4 * c + 3] = te<NT>(R, NT == 2 ? 2 : 3, s[(c + 3) & 3], 24);
    }
}
template <int NT> __device__ __forceinline__ void aes_round_mix(const uint32_t (&t)[16], const uint32_t *__restrict__ rk4, uint32_t (&s)[4]) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (NT == 4) s[c] = xor3(xor3(t[4 * c], t[4 * c + 1], t[4 * c + 2]), t[4 * c + 3], rk4[c]);
        else s[c] = xor3(t[4 * c], t[4 * c + 2], rk4[c]) ^ rotl(t[4 * c + 1] ^ t[4 * c + 3], 8);
    }
}
// two independent blocks, round by round, so that both blocks' table reads are in flight together
//
// Left to itself the compiler consumes every two or three reads at once (141 s_waitcnt lgkmcnt(0) in the two-block loop).
// With two tables (2 048 buckets and more) a fixed order pays, 9.85 -> 8.8 ms: the sixteen reads of one block, then that
// block's XORs and rotates under the sixteen reads of the other.  With four tables the same order costs 4 % (7.2 -> 7.5 ms;
// all 32 reads ahead of all XORs: 7.7), so that layout keeps the compiler's order.
template <int NT> __device__ __forceinline__ void mmo2(const TabRef &R, const uint32_t *__restrict__ rk, const B128 &ia, const B128 &ib, B128 &oa, B128 &ob) {
#ifndef WH_HH_FIXED_ORDER_NT4
#define WH_HH_FIXED_ORDER_NT4 0
#endif
    if constexpr (NT == 2 || (NT == 4 && WH_HH_FIXED_ORDER_NT4)) {
        uint32_t a[4], b[4], ta[16], tb[16];
#pragma unroll
        for (int k = 0; k < 4; k++) { a[k] = ia.w[k] ^ rk[k]; b[k] = ib.w[k] ^ rk[k]; }
        aes_round_reads<NT>(R, a, ta);
#pragma unroll
        for (int r = 1; r < 10; r++) {
            __builtin_amdgcn_sched_barrier(0);
            aes_round_mix<NT>(ta, rk + 4 * r, a);
            aes_round_reads<NT>(R, b, tb);
            __builtin_amdgcn_sched_barrier(0);
            aes_round_mix<NT>(tb, rk + 4 * r, b);
            if (r < 9) aes_round_reads<NT>(R, a, ta);
        }
        oa.w[0] = aes_last<NT>(R, a[0], a[1], a[2], a[3], rk[40]) ^ ia.w[0]; ob.w[0] = aes_last<NT>(R, b[0], b[1], b[2], b[3], rk[40]) ^ ib.w[0];
        oa.w[1] = aes_last<NT>(R, a[1], a[2], a[3], a[0], rk[41]) ^ ia.w[1]; ob.w[1] = aes_last<NT>(R, b[1], b[2], b[3], b[0], rk[41]) ^ ib.w[1];
        oa.w[2] = aes_last<NT>(R, a[2], a[3], a[0], a[1], rk[42]) ^ ia.w[2]; ob.w[2] = aes_last<NT>(R, b[2], b[3], b[0], b[1], rk[42]) ^ ib.w[2];
        oa.w[3] = aes_last<NT>(R, a[3], a[0], a[1], a[2], rk[43]) ^ ia.w[3]; ob.w[3] = aes_last<NT>(R, b[3], b[0], b[1], b[2], rk[43]) ^ ib.w[3];
        return;
    }
    uint32_t a0 = ia.w[0] ^ rk[0], a1 = ia.w[1] ^ rk[1], a2 = ia.w[2] ^ rk[2], a3 = ia.w[3] ^ rk[3];
    uint32_t b0 = ib.w[0] ^ rk[0], b1 = ib.w[1] ^ rk[1], b2 = ib.w[2] ^ rk[2], b3 = ib.w[3] ^ rk[3];
#pragma unroll
    for (int r = 1; r < 10; r++) {
        const uint32_t t0 = aes_col<NT>(R, a0, a1, a2, a3, rk[4 * r]), u0 = aes_col<NT>(R, b0, b1, b2, b3, rk[4 * r]);
        const uint32_t t1 = aes_col<NT>(R, a1, a2, a3, a0, rk[4 * r + 1]), u1 = aes_col<NT>(R, b1, b2, b3, b0, rk[4 * r + 1]);
        const uint32_t t2 = aes_col<NT>(R, a2, a3, a0, a1, rk[4 * r + 2]), u2 = aes_col<NT>(R, b2, b3, b0, b1, rk[4 * r + 2]);
        const uint32_t t3 = aes_col<NT>(R, a3, a0, a1, a2, rk[4 * r + 3]), u3 = aes_col<NT>(R, b3, b0, b1, b2, rk[4 * r + 3]);
        a0 = t0; a1 = t1; a2 = t2; a3 = t3; b0 = u0; b1 = u1; b2 = u2; b3 = u3;
    }
    oa.w[0] = aes_last<NT>(R, a0, a1, a2, a3, rk[40]) ^ ia.w[0]; ob.w[0] = aes_last<NT>(R, b0, b1, b2, b3, rk[40]) ^ ib.w[0];
    oa.w[1] = aes_last<NT>(R, a1, a2, a3, a0, rk[41]) ^ ia.w[1]; ob.w[1] = aes_last<NT>(R, b1, b2, b3, b0, rk[41]) ^ ib.w[1];
    oa.w[2] = aes_last<NT>(R, a2, a3, a0, a1, rk[42]) ^ ia.w[2]; ob.w[2] = aes_last<NT>(R, b2, b3, b0, b1, rk[42]) ^ ib.w[2];
    oa.w[3] = aes_last<NT>(R, a3, a0, a1, a2, rk[43]) ^ ia.w[3]; ob.w[3] = aes_last<NT>(R, b3, b0, b1, b2, rk[43]) ^ ib.w[3];
}
// inc_be (prg.rs:449-452): the counter's upper 64-bit lane
__device__ __forceinline__ void ctr_add(B128 &c, uint32_t k) {
    const uint64_t hi = (((uint64_t)c.w[3] << 32) | c.w[2]) + k;
    c.w[2] = (uint32_t)hi; c.w[3] = (uint32_t)(hi >> 32);
}
template <int NT = 1> __device__ __forceinline__ void fill_table(uint32_t *T, const HhTables *__restrict__ tab, uint32_t tid, uint32_t nthreads) {
    for (uint32_t k = tid; k < T_WORDS; k += nthreads) {
        const uint32_t e = k >> 5, rep = k & 31, v = tab->te0[e];
        if (NT == 1) T[k] = v;
#pragma unroll
        for (int t = 0; t < (NT == 1 ? 0 : NT); t++)   // row of 64 words per entry, half-row (t & 1), region t >> 1
            T[(t >> 1) * 16384 + e * 64 + (t & 1) * 32 + rep] = rotl(v, NT == 2 ? 16 * t : 8 * t);
    }
}

// little-endian loads from byte pointers of any alignment
template <class P> __device__ __forceinline__ uint32_t ldb32(P p) {
    return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}
template <class P> __device__ __forceinline__ uint64_t ldb64(P p) { return (uint64_t)ldb32(p) | (uint64_t)ldb32(p + 4) << 32; }
template <class P> __device__ __forceinline__ B128 ldb128(P p) {
    B128 b;
#pragma unroll
    for (int k = 0; k < 4; k++) b.w[k] = ldb32(p + 4 * k);
    return b;
}

// eight lower-case hex characters of v, most significant nibble first, as two big-endian message words
__device__ __forceinline__ void hex_words(uint32_t v, uint32_t &w0, uint32_t &w1) {
    uint32_t o[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t nib = (v >> (28 - 4 * k)) & 15;
        o[k >> 2] = (o[k >> 2] << 8) | (nib < 10 ? 48 + nib : 87 + nib);
    }
    w0 = o[0]; w1 = o[1];
}
// SHA-256 of 64 bytes given as 16 little-endian words, written as the 64 hex characters in message-block form
__device__ __forceinline__ void sha_of_pi_hex(const uint32_t (&pi)[16], uint32_t *__restrict__ out16) {
    uint32_t h[8], w[16];
    sha_init(h);
#pragma unroll 1
    for (int blk = 0; blk < 2; blk++) {
#pragma unroll
        for (int k = 0; k < 16; k++) w[k] = blk == 0 ? __builtin_bswap32(pi[k]) : (k == 0 ? 0x80000000u : (k == 15 ? 512u : 0u));
        sha_compress(h, w);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) hex_words(h[k], out16[2 * k], out16[2 * k + 1]);
}

// ------------------------------------------------------------------------------------------ main key
// Persistent workgroups of 1024 lanes, one per CU, each walking over key chains blockIdx.x, + gridDim.x, ...  The AES
// tables take NT * 32 KiB of the CU's 160 KiB (four tables up to 512 buckets, two above).  There is no barrier per key
// chain; the sixteen wavefronts are a PRODUCER and fifteen CONSUMERS that meet through counters in LDS:
//
//   producer   copies and checks the header of the workgroup's c-th key chain, computes its testing id and expands the
//              narrow top of its tree, levels 0 .. depth - SP (a chain of AES latencies that would leave the CU idle),
//              into one of three (header, node) buffers, two key chains ahead of the consumers;
//   consumers  take work units from a counter.  Stage st holds the units of two key chains: the leaves of key chain
//              st - 1 -- leaf hashes (64 leaves a unit), conversion into buckets (two leaves a unit, lane = counter, so
//              a leaf is one 1 KiB coalesced store), sign_data and ctr (64 leaves) -- and, placed before the last
//              conversion units so that they are finished when the stage is, the wide bottom of the tree of key chain
//              st: SP levels under 64 nodes per unit, all in registers, into the other of two leaf buffers.
//              A wavefront goes on to the next stage as soon as the counter runs out; the last one to leave a stage
//              writes the sums over the leaves and hands the buffers back.
//
// Measured (profiles/r3_hh_queue.txt): the barrier version left the LDS pipe 61-66 % busy; taking the producer's work away
// altogether gained 12 % on it, a work counter alone 10 % at 512 buckets and -2 % at 2 048 (the producer, slowed by a
// now always busy LDS pipe, became the critical path: 74 AES latencies per tree there).
constexpr uint32_t MAIN_THREADS = 1024, CONSUMERS = MAIN_THREADS - 64, HH_SPLIT = 2;
__host__ __device__ constexpr uint32_t hh_split(uint32_t depth) { return depth < HH_SPLIT ? depth : HH_SPLIT; }

// header copy, checks, testing id and the tree's levels 0 .. levels of key chain i into (hdr, nodes, cbits): level l has 2^l
// nodes at nodes[0 .. 2^l); one wavefront, no workgroup barrier inside
template <int NT>
__device__ __forceinline__ bool hh_prepare(const TabRef &T, const HhTables *__restrict__ tab, const uint8_t *kb, uint32_t depth,
                                            uint32_t levels, uint32_t lane, uint8_t *hdr, uint4 *seeds, uint8_t *cbits, uint8_t *id_out) {
    const HhLayout L(depth);
    const uint32_t *rk0 = tab->rk[0];
    for (uint32_t k = lane; k < L.sup; k += 64) hdr[k] = kb[k];
    // what bincode, the indexing in eval_all and the asserts of eval_and_check would abort on: every support key's
    // header and the main key's
    bool ok = true;
    for (uint32_t sk = lane; sk < HH_SUPPORT; sk += 64) {
        const uint8_t *p = kb + L.sup + HH_SUPPORT_LEN * sk;
        ok = ok && p[0] <= 1 && ldb64(p + 17) == 1 && p[41] <= 1 && p[42] <= 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == 0) {
        ok = ok && hdr[L.key_id] <= 1 && ldb64(hdr + L.cw_len) == depth && ldb64(hdr + L.data_len) == 256 &&
             ldb64(hdr + L.sup_len) == HH_SUPPORT;
        for (uint32_t l = 0; l < depth; l++) ok = ok && hdr[L.cw + 18 * l + 16] <= 1 && hdr[L.cw + 18 * l + 17] <= 1;
        // testing id = AES_vk(id.to_le_bytes()), read little endian (server-hh/src/main.rs:258-259, prg.rs:650-655)
        const B128 e = aes_enc<NT>(T, tab->rk[2], ldb128(hdr));
        uint32_t *o = reinterpret_cast<uint32_t *>(id_out);
        o[0] = e.w[0]; o[1] = e.w[1]; o[2] = e.w[2]; o[3] = e.w[3];
        const B128 root = ldb128(hdr + L.root);
        seeds[0] = make_uint4(root.w[0], root.w[1], root.w[2], root.w[3]);
        cbits[0] = hdr[L.key_id];
    }
    ok = __all(ok);
    if (!ok) return false;
    // the children of node p go to 2p and 2p+1, in place, from the last 64-parent pass down so that children never land
    // on parents still to be read
    for (uint32_t l = 0; l < levels; l++) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const B128 cw = ldb128(hdr + L.cw + 18 * l);
        const uint32_t cb0 = hdr[L.cw + 18 * l + 16], cb1 = hdr[L.cw + 18 * l + 17];
        const uint32_t P = 1u << l;
#pragma unroll 1
        for (int c = (int)((P - 1) >> 6); c >= 0; c--) {
            const uint32_t p = (uint32_t)c * 64 + lane;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (p < P) {
                const uint4 s = seeds[p];
                const uint32_t bit = cbits[p];
                // expand_direction(true, true): control bits from the seed's two low bits, which are then cleared
                uint32_t bl = (s.x & 1) == 0, br = (s.x & 2) == 0;
                B128 c0 = {{s.x & ~3u, s.y, s.z, s.w}};
                B128 left = mmo<NT>(T, rk0, c0);
                ctr_add(c0, 1);
                B128 right = mmo<NT>(T, rk0, c0);
                if (bit) {
#pragma unroll
                    for (int k = 0; k < 4; k++) { left.w[k] ^= cw.w[k]; right.w[k] ^= cw.w[k]; }
                    bl ^= cb0; br ^= cb1;
                }
                seeds[2 * p] = make_uint4(left.w[0], left.w[1], left.w[2], left.w[3]);
                seeds[2 * p + 1] = make_uint4(right.w[0], right.w[1], right.w[2], right.w[3]);
                cbits[2 * p] = (uint8_t)bl; cbits[2 * p + 1] = (uint8_t)br;
            }
        }
    }
    return true;
}

// Where the bucket vectors go.  dat: every key chain's Vec<Bucket<u32>> (NULL: not materialised).  sketch: atomic adds
// straight into a sketch, key chain g (its index in the caller's array) into sketch number g / group_size, flip = 1
// subtracting -- the way the few key chains the group test rejected are taken out of a sum again.  Measured on MI355X
// (profiles/r2_hh_fused_atomics.txt): one atomic per counter sustains 0.34 TB/s of added bytes, three times slower than
// writing the vectors and summing them (k_hh_sum_u32), with device-scope atomics and with per-XCD partial sketches
// alike; so whole batches are summed from materialised vectors (api.hip: hh_eval) and this path serves the subtraction.
// In such a launch (tpc == NULL) only the buckets are computed.
struct HhSketchOut {
    uint32_t *sketch;          // NULL: no accumulation
    uint64_t group_size, first;
    const uint32_t *kidx;      // launch slot -> key chain index (NULL: first + slot)
    uint32_t flip;
};
__device__ __forceinline__ void sk_add(uint32_t *p, uint32_t v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// counters in LDS, all written by lane 0 of a wavefront
enum : uint32_t {
    Q_WORK = 0,        // [2] next work unit of stage st, at st & 1
    Q_LEFT_N = 2,      // [2] consumers that have left stage st
    Q_TREE = 4,        // [2] tree units finished for the leaf buffer
    Q_READY = 6,       // key chains the producer has published
    Q_LEFT = 7,        // stages every consumer has left
    Q_WORDS = 8
};
__device__ __forceinline__ void q_wait(const uint32_t *p, uint32_t v) {
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(4);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int NT>
__global__ __launch_bounds__(1024) void k_hh_main(const uint8_t *__restrict__ keys, uint64_t stride, uint32_t n, uint32_t depth,
                                                  const HhTables *__restrict__ tab, uint32_t *__restrict__ dat,
                                                  uint4 *__restrict__ tpc, uint32_t *__restrict__ buckg,
                                                  uint8_t *__restrict__ ids, int32_t *__restrict__ status, HhSketchOut so) {
    extern __shared__ uint32_t lds[];
    const uint32_t N = 1u << depth, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t SP = hh_split(depth), NP = N >> SP;                    // nodes of level depth - SP: the producer's last
    uint32_t *Tall = lds;
    uint4 *seeds_b = reinterpret_cast<uint4 *>(lds + NT * T_WORDS);      // [2][N] leaves
    uint4 *nodes_b = seeds_b + 2 * N;                                     // [3][NP] the producer's tree
    uint8_t *hdr_b = reinterpret_cast<uint8_t *>(nodes_b + 3 * NP);      // [3][HDR_BYTES]
    uint32_t *red_b = reinterpret_cast<uint32_t *>(hdr_b + 3 * HDR_BYTES);  // [2][260] sums over the leaves
    uint32_t *okf = red_b + 520;                                          // [3] (+1)
    uint32_t *qs = okf + 4;                                               // [Q_WORDS]
    uint8_t *id_lds = reinterpret_cast<uint8_t *>(qs + Q_WORDS);          // [16] testing id of a subtract launch (nowhere to put it)
    uint8_t *cbits_b = id_lds + 16;                                       // [3][NP] control bits of the producer's nodes
    const HhLayout L(depth);
    const bool producer = wv == MAIN_THREADS / 64 - 1;

    fill_table<NT>(Tall, tab, tid, MAIN_THREADS);
    if (tid < 520) red_b[tid] = 0;
    if (tid < Q_WORDS) qs[tid] = 0;
    __syncthreads();
    const TabRef T = tab_ref(Tall, tid);
    const uint32_t *rk0 = tab->rk[0];
    const uint32_t G = gridDim.x;
    const uint64_t words = (uint64_t)N * HH_BKT;
    // `keys` starts at the launch's first key chain (kidx == NULL) or is the caller's whole array (kidx != NULL)
    auto key_slot = [&](uint32_t k) -> uint64_t { return so.kidx ? so.kidx[k] : k; };
    auto key_index = [&](uint32_t k) -> uint64_t { return so.kidx ? so.kidx[k] : so.first + k; };
    uint32_t *sk_base = so.sketch;
    const uint32_t C = (n - blockIdx.x + G - 1) / G;      // key chains of this workgroup (grid <= n)
    auto post = [&](uint32_t *p, uint32_t v) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    if (producer) {
#pragma unroll 1
        for (uint32_t c = 0; c < C; c++) {
            // buffer c % 3 held key chain c - 3, whose leaves are the business of stage c - 2
            if (c >= 3) q_wait(&qs[Q_LEFT], c - 1);
            const uint32_t k = blockIdx.x + c * G, h = c % 3;
            const bool ok = hh_prepare<NT>(T, tab, keys + key_slot(k) * stride, depth, depth - SP, lane, hdr_b + h * HDR_BYTES,
                                           nodes_b + h * NP, cbits_b + h * NP, ids ? ids + 16ull * k : id_lds);
            if (lane == 0) { okf[h] = ok; if (status) status[k] = ok ? 0 : HH_ERR_KEY_FORMAT; }
            post(&qs[Q_READY], c + 1);
        }
        return;
    }

    const uint32_t TU = (NP + 63) >> 6;                   // tree units of a key chain
    const uint32_t U2 = (N + 63) >> 6, U1 = tpc ? U2 : 0, H = (N + 1) >> 1;
    // the tree units go before the last HT conversion units: late enough for the producer, early enough to be done
    const uint32_t HT = H < 4 * (CONSUMERS / 64) ? H / 2 : 3 * (CONSUMERS / 64), H1 = H - HT;
#pragma unroll 1
    for (uint32_t st = 0; st <= C; st++) {
        const uint32_t b = st & 1;
        // the counters at b belonged to stage st - 2
        if (st >= 2) q_wait(&qs[Q_LEFT], st - 1);
        if (st >= 1) q_wait(&qs[Q_READY], st);            // (its tree units were taken a stage ago, perhaps not yet started)
        const bool has_tree = st < C;                     // key chain st: buffers st % 3, leaves into st & 1
        const uint32_t c = st - 1, hb = c % 3, lb = c & 1;  // key chain of the leaves (st >= 1)
        const uint32_t k = blockIdx.x + c * G;
        const bool leaves = st >= 1 && okf[st >= 1 ? hb : 0];
        uint4 *seeds = seeds_b + lb * N;
        const uint8_t *hdr = hdr_b + hb * HDR_BYTES;
        uint32_t *red = red_b + lb * 260;
        const uint64_t leaf0 = (uint64_t)k * N;
        if (st >= 1 && !leaves && dat) {
            uint32_t *d = dat + leaf0 * HH_BKT;
            for (uint64_t q = tid; q < words; q += CONSUMERS) d[q] = 0;
        }
        const uint32_t nA = leaves ? U1 : 0, nB = nA + (leaves ? H1 : 0), nC = nB + (has_tree ? TU : 0), nD = nC + (leaves ? HT : 0),
                       nE = nD + (leaves ? U2 : 0);
        const bool negate = leaves && ((hdr[L.key_id] != 0) != (so.flip != 0));
        uint32_t *sk = leaves && sk_base ? sk_base + (key_index(k) / so.group_size) * words : nullptr;
        uint32_t oc0 = 0, oc1 = 0;
        B128 oc = {{0, 0, 0, 0}};
        if (leaves) {
            q_wait(&qs[Q_TREE + lb], TU);                 // (done a stage ago, all but the first time)
            oc0 = ldb32(hdr + L.oc + 1024); oc1 = ldb32(hdr + L.oc + 1028);
            oc = ldb128(hdr + L.oc + 16 * lane);
        }
        uint32_t acc[4] = {0, 0, 0, 0}, t0 = 0, t1 = 0;
        // the next unit's number is asked for before this one's work and arrives under it
        auto grab = [&]() { uint32_t v = 0; if (lane == 0) v = atomicAdd(&qs[Q_WORK + b], 1u); return v; };
        // ---- per-leaf pieces that occupy one lane each:
        //  hash: tilde_pi = mmo_hash2to4(seed, x) with x = IntModN::zero() at EVERY leaf (IntModN::one() has val 0,
        //    common/src/group.rs:448-453, so `x.add(&GIn::one())` never moves), corrected by cs when lsb(seed) = 1;
        //  else: sign_data and ctr of the converted bucket = the first 8 bytes of counter seed+64.
        auto leaf_unit = [&](uint32_t x, bool hash) {
            if (x >= N) return;
            const uint4 sd = seeds[x];
            const uint32_t m = 0u - (sd.x & 1);
            B128 cc = {{sd.x, sd.y, sd.z, sd.w}};
            if (hash) {
                const B128 s1 = mmo<NT>(T, rk0, cc);
                const B128 s3 = mmo<NT>(T, rk0, s1);
                const uint8_t *cs = hdr + L.cs;
                uint4 *o = tpc + (leaf0 + x) * 4;
                B128 q = ldb128(cs);
                o[0] = make_uint4(tab->cx[0][0] ^ (q.w[0] & m), tab->cx[0][1] ^ (q.w[1] & m), tab->cx[0][2] ^ (q.w[2] & m), tab->cx[0][3] ^ (q.w[3] & m));
                q = ldb128(cs + 16);
                o[1] = make_uint4(s1.w[0] ^ (q.w[0] & m), s1.w[1] ^ (q.w[1] & m), s1.w[2] ^ (q.w[2] & m), s1.w[3] ^ (q.w[3] & m));
                q = ldb128(cs + 32);
                o[2] = make_uint4(tab->cx[0][4] ^ (q.w[0] & m), tab->cx[0][5] ^ (q.w[1] & m), tab->cx[0][6] ^ (q.w[2] & m), tab->cx[0][7] ^ (q.w[3] & m));
                q = ldb128(cs + 48);
                o[3] = make_uint4(s3.w[0] ^ (q.w[0] & m), s3.w[1] ^ (q.w[1] & m), s3.w[2] ^ (q.w[2] & m), s3.w[3] ^ (q.w[3] & m));
            } else {
                ctr_add(cc, 64);
                const B128 o = mmo<NT>(T, rk0, cc);
                uint32_t a = o.w[0] + (oc0 & m), bb = o.w[1] + (oc1 & m);
                if (negate) { a = 0u - a; bb = 0u - bb; }
                if (dat) *reinterpret_cast<uint2 *>(dat + (leaf0 + x) * HH_BKT + 256) = make_uint2(a, bb);
                if (sk) { sk_add(sk + (uint64_t)x * HH_BKT + 256, a); sk_add(sk + (uint64_t)x * HH_BKT + 257, bb); }
                t0 += a; t1 += bb;
            }
        };
        // ---- convert::<Bucket<u32>>: 1032 bytes = counters seed+0 .. seed+64 (the last half used, above); leaves 2v and
        // 2v + 1 together: 32 table reads in flight per round instead of 16 keep the LDS queue fed; the sum over leaves
        // stays in registers
        auto pair_unit = [&](uint32_t v) {
            const uint32_t x = 2 * v, x1 = x + 1 < N ? x + 1 : x;    // N = 1: the leaf is computed twice, stored once
            const uint4 sa = seeds[x], sb = seeds[x1];
            B128 ca = {{sa.x, sa.y, sa.z, sa.w}}, cb = {{sb.x, sb.y, sb.z, sb.w}};
            ctr_add(ca, lane); ctr_add(cb, lane);
            B128 oa, ob;
            mmo2<NT>(T, rk0, ca, cb, oa, ob);
            const uint32_t ma = 0u - (sa.x & 1), mb = 0u - (sb.x & 1);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                oa.w[q] += oc.w[q] & ma; ob.w[q] += oc.w[q] & mb;
                if (negate) { oa.w[q] = 0u - oa.w[q]; ob.w[q] = 0u - ob.w[q]; }
                acc[q] += oa.w[q];
                if (x1 != x) acc[q] += ob.w[q];
            }
            if (dat) {
                uint2 *d = reinterpret_cast<uint2 *>(dat + (leaf0 + x) * HH_BKT + 4 * lane);  // 8-byte aligned only
                d[0] = make_uint2(oa.w[0], oa.w[1]);
                d[1] = make_uint2(oa.w[2], oa.w[3]);
                if (x1 != x) {
                    d = reinterpret_cast<uint2 *>(dat + (leaf0 + x1) * HH_BKT + 4 * lane);
                    d[0] = make_uint2(ob.w[0], ob.w[1]);
                    d[1] = make_uint2(ob.w[2], ob.w[3]);
                }
            }
            if (sk) {
                uint32_t *pa = sk + (uint64_t)x * HH_BKT + 4 * lane, *pb = sk + (uint64_t)x1 * HH_BKT + 4 * lane;
#pragma unroll
                for (int q = 0; q < 4; q++) sk_add(pa + q, oa.w[q]);
                if (x1 != x) {
#pragma unroll
                    for (int q = 0; q < 4; q++) sk_add(pb + q, ob.w[q]);
                }
            }
        };
        // ---- the last SP levels of key chain st's tree under 64 of the producer's nodes: lane = node, its 2^SP leaves in
        // registers (PrgSeed::expand_direction, common/src/prg.rs:133-174; the two children of a node are one mmo2)
        auto tree_unit = [&](uint32_t j) {
            const uint32_t h1 = st % 3;
            const uint8_t *hdr1 = hdr_b + h1 * HDR_BYTES;
            const uint32_t p = 64 * j + lane;
            if (okf[h1] && p < NP) {
                B128 nd[1u << HH_SPLIT];
                uint32_t tb[1u << HH_SPLIT];
                const uint4 s0 = nodes_b[h1 * NP + p];
                nd[0] = B128{{s0.x, s0.y, s0.z, s0.w}};
                tb[0] = cbits_b[h1 * NP + p];
#pragma unroll
                for (uint32_t l = 0; l < HH_SPLIT; l++) {
                    if (l < SP) {
                        const uint8_t *cwp = hdr1 + L.cw + 18 * (depth - SP + l);
                        const B128 cw = ldb128(cwp);
                        const uint32_t cb0 = cwp[16], cb1 = cwp[17];
#pragma unroll
                        for (int q = (1 << l) - 1; q >= 0; q--) {
                            const B128 s = nd[q];
                            const uint32_t bit = tb[q];
                            uint32_t bl = (s.w[0] & 1) == 0, br = (s.w[0] & 2) == 0;
                            B128 c0 = {{s.w[0] & ~3u, s.w[1], s.w[2], s.w[3]}}, c1 = c0;
                            ctr_add(c1, 1);
                            B128 left, right;
                            mmo2<NT>(T, rk0, c0, c1, left, right);
                            const uint32_t mk = 0u - bit;
#pragma unroll
                            for (int w = 0; w < 4; w++) { left.w[w] ^= cw.w[w] & mk; right.w[w] ^= cw.w[w] & mk; }
                            bl ^= cb0 & bit; br ^= cb1 & bit;
                            nd[2 * q] = left; nd[2 * q + 1] = right;
                            tb[2 * q] = bl; tb[2 * q + 1] = br;
                        }
                    }
                }
                uint4 *out = seeds_b + (st & 1) * N + ((uint64_t)p << SP);
#pragma unroll
                for (uint32_t q = 0; q < (1u << HH_SPLIT); q++)
                    if (q < (1u << SP)) out[q] = make_uint4(nd[q].w[0], nd[q].w[1], nd[q].w[2], nd[q].w[3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) atomicAdd(&qs[Q_TREE + (st & 1)], 1u);
        };

        uint32_t u = __builtin_amdgcn_readfirstlane(grab());
#pragma unroll 1
        while (u < nA) { const uint32_t un = grab(); leaf_unit(64 * u + lane, true); u = __builtin_amdgcn_readfirstlane(un); }
#pragma unroll 1
        while (u < nB) { const uint32_t un = grab(); pair_unit(u - nA); u = __builtin_amdgcn_readfirstlane(un); }
        if (u < nC) {
            // the producer has published key chain st, and the leaf buffer's last readers (stage st - 1) have left
            q_wait(&qs[Q_READY], st + 1);
            if (st >= 2) q_wait(&qs[Q_LEFT], st);
#pragma unroll 1
            while (u < nC) { const uint32_t un = grab(); tree_unit(u - nB); u = __builtin_amdgcn_readfirstlane(un); }
        }
#pragma unroll 1
        while (u < nD) { const uint32_t un = grab(); pair_unit(H1 + u - nC); u = __builtin_amdgcn_readfirstlane(un); }
#pragma unroll 1
        while (u < nE) { const uint32_t un = grab(); leaf_unit(64 * (u - nD) + lane, false); u = __builtin_amdgcn_readfirstlane(un); }
        if (leaves) {
            if (t0 | t1) { atomicAdd(&red[256], t0); atomicAdd(&red[257], t1); }
#pragma unroll
            for (int q = 0; q < 4; q++) atomicAdd(&red[4 * lane + q], acc[q]);
        }
        // leave the stage; the last wavefront out writes the sums and hands the buffers back
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        uint32_t dn = 0;
        if (lane == 0) dn = atomicAdd(&qs[Q_LEFT_N + b], 1u);
        dn = __builtin_amdgcn_readfirstlane(dn);
        if (dn == CONSUMERS / 64 - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (st >= 1) {
                for (uint32_t q = lane; q < HH_BKT; q += 64) { if (buckg) buckg[(uint64_t)k * HH_BKT + q] = red[q]; red[q] = 0; }
                if (lane == 0) qs[Q_TREE + lb] = 0;
            }
            if (lane == 0) { qs[Q_WORK + b] = 0; qs[Q_LEFT_N + b] = 0; }
            post(&qs[Q_LEFT], st + 1);
        }
    }
}

// the same from a pointer known to be global memory: one unaligned load instruction (gfx950 global loads are byte-addressable)
// instead of four or sixteen byte loads
typedef uint32_t hh_u32_u1 __attribute__((aligned(1)));
typedef uint32_t hh_u32x4 __attribute__((ext_vector_type(4)));
typedef hh_u32x4 hh_u32x4_u1 __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t ldg32u(const uint8_t *p) { return *(const __attribute__((address_space(1))) hh_u32_u1 *)p; }
__device__ __forceinline__ B128 ldg128u(const uint8_t *p) {
    const hh_u32x4 t = *(const __attribute__((address_space(1))) hh_u32x4_u1 *)p;
    return B128{{t.x, t.y, t.z, t.w}};
}

// ------------------------------------------------------------------------------------------ support keys
// VDPFKey<u32>::eval_all::<bool>: one level, two leaves, hashed index x = false, true
constexpr uint32_t SUP_THREADS = 1024, TAIL_NT = 4;
__global__ __launch_bounds__(1024) void k_hh_support(const uint8_t *__restrict__ keys, uint64_t stride, uint32_t n, uint32_t depth,
                                                    const HhTables *__restrict__ tab, uint32_t *__restrict__ proofblk,
                                                    uint32_t *__restrict__ supd, const int32_t *__restrict__ status) {
    extern __shared__ uint32_t Tall[];   // the four AES tables, 128 KiB
    const uint32_t tid = threadIdx.x;
    fill_table<TAIL_NT>(Tall, tab, tid, SUP_THREADS);
    __syncthreads();
    const TabRef T = tab_ref(Tall, tid);
    const HhLayout L(depth);
    // persistent workgroups (one per CU: the tables are filled once, not once per 1024 keys)
#pragma unroll 1
    for (uint64_t g = (uint64_t)blockIdx.x * SUP_THREADS + tid; g < (uint64_t)n * HH_SUPPORT; g += (uint64_t)gridDim.x * SUP_THREADS) {
    const uint64_t i = g / HH_SUPPORT;
    const uint32_t sup = (uint32_t)(g - i * HH_SUPPORT);
    const uint8_t *p = keys + i * stride + L.sup + (uint64_t)HH_SUPPORT_LEN * sup;
    if (status[i] != 0) continue;  // rejected by k_hh_main (it checks every support key's header too)
    // the 111 bytes of a support key as seven unaligned loads (byte-wise they were 101 load instructions per lane)
    const uint32_t kid = p[0];
    const B128 root = ldg128u(p + 1);
    const B128 cw = ldg128u(p + 25);
    uint32_t cs[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const B128 t = ldg128u(p + 43 + 16 * q);
#pragma unroll
        for (int k = 0; k < 4; k++) cs[4 * q + k] = t.w[k];
    }
    const uint32_t oc = ldg32u(p + 107);
    const uint32_t *rk0 = tab->rk[0], *rk1 = tab->rk[1];

    B128 c0 = {{root.w[0] & ~3u, root.w[1], root.w[2], root.w[3]}};
    B128 leaf[2];
    leaf[0] = mmo<TAIL_NT>(T, rk0, c0);
    ctr_add(c0, 1);
    leaf[1] = mmo<TAIL_NT>(T, rk0, c0);
    uint32_t pi[16], d[2];
#pragma unroll
    for (int k = 0; k < 16; k++) pi[k] = cs[k];
#pragma unroll 1
    for (int x = 0; x < 2; x++) {
        B128 s = leaf[0];
        if (x) s = leaf[1];
        if (kid) { s.w[0] ^= cw.w[0]; s.w[1] ^= cw.w[1]; s.w[2] ^= cw.w[2]; s.w[3] ^= cw.w[3]; }  // state.bit = key_id at the root
        const B128 s1 = mmo<TAIL_NT>(T, rk0, s);
        const B128 s3 = mmo<TAIL_NT>(T, rk0, s1);
        const uint32_t m = 0u - (s.w[0] & 1);
        // convert::<u32>: next_u32 -> refill -> the first four bytes of MMO_0(seed), which is s1
        uint32_t word = s1.w[0] + (oc & m);
        if (kid) word = 0u - word;
        d[x] = word;
        B128 hin[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            hin[0].w[k] = pi[k] ^ tab->cx[x][k] ^ (cs[k] & m);
            hin[1].w[k] = pi[4 + k] ^ s1.w[k] ^ (cs[4 + k] & m);
            hin[2].w[k] = pi[8 + k] ^ tab->cx[x][4 + k] ^ (cs[8 + k] & m);
            hin[3].w[k] = pi[12 + k] ^ s3.w[k] ^ (cs[12 + k] & m);
        }
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const B128 ho = mmo<TAIL_NT>(T, rk1, hin[b]);
#pragma unroll
            for (int k = 0; k < 4; k++) pi[4 * b + k] ^= ho.w[k];
        }
    }
    supd[g * 2] = d[0];
    supd[g * 2 + 1] = d[1];
    sha_of_pi_hex(pi, proofblk + (i * (HH_SUPPORT + 1) + 1 + sup) * 16);
    }
}

// ------------------------------------------------------------------------------------------ proof chain of the main key
// pi <- pi ^ H'(pi ^ tilde_pi_x) for x = 0 .. N-1.  H' (mmo_hash4to4) hashes its four 16-byte blocks independently, so
// the chain is four independent block chains, and a step is one AES on the critical path: the kernel is bound by the
// latency of that AES, not by throughput.  Sixteen lanes per key chain: lane 4 b + q carries column q of block b, a round is
// four table reads per lane instead of sixteen, and the three other columns of the block come from the neighbours of the
// quad by DPP (quad_perm rotations, no LDS).  Each lane keeps its own eleven round-key words.
__device__ __forceinline__ uint32_t quad_rot(uint32_t x, int by) {          // x of lane (q + by) % 4 of this quad
    if (by == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x39, 0xf, 0xf, false);   // quad_perm:[1,2,3,0]
    if (by == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4e, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x93, 0xf, 0xf, false);                // quad_perm:[3,0,1,2]
}
__device__ __forceinline__ uint32_t mmo_quad(const TabRef &R, const uint32_t (&rkq)[11], uint32_t in) {
    uint32_t s = in ^ rkq[0];
#pragma unroll
    for (int r = 1; r < 10; r++) {
        const uint32_t s1 = quad_rot(s, 1), s2 = quad_rot(s, 2), s3 = quad_rot(s, 3);
        s = xor3(xor3(te<4>(R, 0, s, 0), te<4>(R, 1, s1, 8), te<4>(R, 2, s2, 16)), te<4>(R, 3, s3, 24), rkq[r]);
    }
    const uint32_t s1 = quad_rot(s, 1), s2 = quad_rot(s, 2), s3 = quad_rot(s, 3);
    s = ((te<4>(R, 2, s, 0) & 0xff) | (te<4>(R, 3, s1, 8) & 0xff00) | (te<4>(R, 0, s2, 16) & 0xff0000) | (te<4>(R, 1, s3, 24) & 0xff000000)) ^ rkq[10];
    return s ^ in;
}
__global__ __launch_bounds__(1024) void k_hh_chain(const uint8_t *__restrict__ keys, uint64_t stride, uint32_t n, uint32_t depth,
                                                   const HhTables *__restrict__ tab, const uint4 *__restrict__ tpc,
                                                   uint32_t *__restrict__ proofblk, const int32_t *__restrict__ status) {
    extern __shared__ uint32_t Tall[];   // the four AES tables, 128 KiB
    const uint32_t tid = threadIdx.x;
    fill_table<TAIL_NT>(Tall, tab, tid, blockDim.x);
    __syncthreads();
    const TabRef T = tab_ref(Tall, tid);
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + tid;
    const uint64_t i = g >> 4;
    const uint32_t w = (uint32_t)g & 15, q = w & 3;          // word w of the 64-byte proof: block w / 4, column q
    const bool live = i < n && status[i] == 0;               // uniform over the 16 lanes of a key chain
    const HhLayout L(depth);
    const uint32_t N = 1u << depth;
    uint32_t rkq[11];
#pragma unroll
    for (int r = 0; r < 11; r++) rkq[r] = tab->rk[1][4 * r + q];
    uint32_t pi = 0;
    if (live) pi = ldb32(keys + i * stride + L.cs + 4 * w);
    // every lane runs the loop (DPP needs its quad active); a dead key chain reads row 0 of key chain 0 and is never stored
    const uint32_t *tp = reinterpret_cast<const uint32_t *>(tpc) + (live ? i : 0) * N * 16 + w;
    uint32_t nx = tp[0];
#pragma unroll 1
    for (uint32_t x = 0; x < N; x++) {
        const uint32_t hin = pi ^ nx;
        if (x + 1 < N) nx = tp[(x + 1) * 16];
        pi ^= mmo_quad(T, rkq, hin);
    }
    // gather the sixteen words in the first lane of the group
    uint32_t all[16];
#pragma unroll
    for (int k = 0; k < 16; k++) all[k] = __shfl(pi, (int)((tid & 48) + k), 64);
    if (live && w == 0) sha_of_pi_hex(all, proofblk + i * (HH_SUPPORT + 1) * 16);
}

// ------------------------------------------------------------------------------------------ folding and the tag
// Three SHA-256 messages per key chain:
//   A  bytes of the folded bucket (258 u32, little endian)                 17 blocks
//   B  bytes of the support sums (257 u32)                                 17 blocks
//   C  the proof string: hex(main) | 257 x hex(support) | hex(A) | hex(B)  260 blocks + padding
// A lane pair per key chain: the odd lane hashes A and B while the even lane is on the first 34 blocks of C (one copy of
// the compression, the message words chosen per lane), hands the two digests over, and the even lane finishes C: 261
// compressions in sequence instead of 295.
__global__ __launch_bounds__(64) void k_hh_final(const uint8_t *__restrict__ keys, uint64_t stride, uint32_t n, uint32_t depth,
                                                 const uint32_t *__restrict__ proofblk, const uint32_t *__restrict__ supd,
                                                 const uint32_t *__restrict__ buckg, uint8_t *__restrict__ tags,
                                                 const int32_t *__restrict__ status) {
    const uint64_t t = (uint64_t)blockIdx.x * 64 + threadIdx.x, i = t >> 1;
    const bool odd = t & 1;
    if (i >= n) return;
    uint32_t *tag = reinterpret_cast<uint32_t *>(tags + 64 * i);
    if (status[i] != 0) {
        if (!odd) {
#pragma unroll
            for (int k = 0; k < 16; k++) tag[k] = 0;
        }
        return;
    }
    const HhLayout L(depth);
    const bool kid = keys[i * stride + L.key_id] != 0;
    const uint32_t *bk = buckg + i * HH_BKT, *sd = supd + i * HH_SUPPORT * 2, *pb = proofblk + i * (HH_SUPPORT + 1) * 16;
    uint32_t h[8], hexA[16], hexB[16], w[16];
    sha_init(h);
    // message C: 258 ready-made blocks (four 16-byte loads each, the next block's in flight under the compression),
    // then hex(A), hex(B) and the padding block
    const uint4 *pq = reinterpret_cast<const uint4 *>(pb);
    uint4 nx[4] = {pq[0], pq[1], pq[2], pq[3]};
    auto c_block = [&](uint32_t b) {
#pragma unroll
        for (int q = 0; q < 4; q++) { w[4 * q] = nx[q].x; w[4 * q + 1] = nx[q].y; w[4 * q + 2] = nx[q].z; w[4 * q + 3] = nx[q].w; }
        if (b + 1 < HH_SUPPORT + 1) {
#pragma unroll
            for (int q = 0; q < 4; q++) nx[q] = pq[4 * (b + 1) + q];
        }
    };
#pragma unroll 1
    for (uint32_t blk = 0; blk < 34; blk++) {
        if (odd) {
            const bool isA = blk < 17;
            const uint32_t b = isA ? blk : blk - 17, words = isA ? HH_BKT : HH_SUPPORT;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t e = 16 * b + k;
                uint32_t v = 0;
                if (e < words) {
                    const uint32_t d0 = e < HH_SUPPORT ? sd[2 * e] : 0, d1 = e < HH_SUPPORT ? sd[2 * e + 1] : 0;
                    if (isA) {
                        // buck -= (right - left) per support key; key 1 also takes 1 off ctr and negates (protocol.rs:99-126)
                        v = bk[e] - (d1 - d0);
                        if (kid) { if (e == HH_BKT - 1) v -= 1; v = 0u - v; }
                    } else {
                        v = d0 + d1;                      // must be 1 across the two servers (protocol.rs:104-108, 143-148)
                        if (kid) v = 0u - (v - 1);
                    }
                    v = __builtin_bswap32(v);             // little-endian bytes of the u32 as a big-endian message word
                } else if (e == words) v = 0x80000000u;
                else if (e == 271) v = words * 32;        // message length in bits
                w[k] = v;
            }
        } else {
            c_block(blk);
        }
        sha_compress(h, w);
        if (odd && (blk == 16 || blk == 33)) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t a, b2;
                hex_words(h[k], a, b2);
                if (blk == 16) { hexA[2 * k] = a; hexA[2 * k + 1] = b2; } else { hexB[2 * k] = a; hexB[2 * k + 1] = b2; }
            }
            sha_init(h);
        }
    }
    // the digests cross to the even lane (the pair sits in one quad: a DPP swap of neighbours)
#pragma unroll
    for (int k = 0; k < 16; k++) {
        hexA[k] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hexA[k], 0xB1, 0xF, 0xF, true);     // quad_perm [1, 0, 3, 2]
        hexB[k] = (uint32_t)__builtin_amdgcn_mov_dpp((int)hexB[k], 0xB1, 0xF, 0xF, true);
    }
    if (odd) return;
#pragma unroll 1
    for (uint32_t b = 34; b < HH_SUPPORT + 4; b++) {
        if (b < HH_SUPPORT + 1) {
            c_block(b);
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++)
                w[k] = b == HH_SUPPORT + 1 ? hexA[k] : (b == HH_SUPPORT + 2 ? hexB[k] : (k == 0 ? 0x80000000u : (k == 15 ? 64u * (HH_SUPPORT + 3) * 8 : 0u)));
        }
        sha_compress(h, w);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t a, b;
        hex_words(h[k], a, b);
        tag[2 * k] = __builtin_bswap32(a);
        tag[2 * k + 1] = __builtin_bswap32(b);
    }
}

}  // namespace

static int hh_main_tables(uint32_t depth) { return depth <= 9 ? 4 : 2; }
size_t hh_main_lds_bytes(uint32_t depth) {
    const size_t N = (size_t)1 << depth;
    const size_t NP = N >> hh_split(depth);
    return 4 * (size_t)T_WORDS * hh_main_tables(depth) + 2 * 16 * N + 3 * 16 * NP + 3 * HDR_BYTES + 4 * (524 + Q_WORDS) + 16 + 3 * NP + 16;
}

int launch_hh_main(const uint8_t *d_keys, uint64_t stride, uint32_t n, uint32_t depth, const HhTables *d_tab, const HhWork &w,
                   uint32_t *d_dat, uint8_t *d_ids, int32_t *d_status, const HhSketch &sk, hipStream_t stream) {
    const size_t lds = hh_main_lds_bytes(depth);
    auto kmain = hh_main_tables(depth) == 4 ? k_hh_main<4> : k_hh_main<2>;
    const HhSketchOut so{sk.sketch, sk.group_size ? sk.group_size : 1, sk.first, sk.kidx, sk.flip};
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kmain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t grid = n < (uint32_t)cus ? n : (uint32_t)cus;   // one persistent workgroup per CU
    hipLaunchKernelGGL(kmain, dim3(grid), dim3(MAIN_THREADS), lds, stream, d_keys, stride, n, depth, d_tab, d_dat, sk.buckets_only ? nullptr : w.tpc,
                       sk.buckets_only ? nullptr : w.buckg, d_ids, d_status, so);
    return (int)hipGetLastError();
}
static const size_t TAIL_LDS = 4 * (size_t)T_WORDS * TAIL_NT;
int launch_hh_support(const uint8_t *d_keys, uint64_t stride, uint32_t n, uint32_t depth, const HhTables *d_tab, const HhWork &w,
                      const int32_t *d_status, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_hh_support), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TAIL_LDS);
    if (e != hipSuccess) return (int)e;
    const uint64_t sup_threads = (uint64_t)n * HH_SUPPORT;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    uint64_t groups = (sup_threads + SUP_THREADS - 1) / SUP_THREADS;
    if (groups > (uint64_t)cus) groups = cus;
    if (!groups) groups = 1;
    hipLaunchKernelGGL(k_hh_support, dim3((uint32_t)groups), dim3(SUP_THREADS), TAIL_LDS, stream, d_keys, stride, n,
                       depth, d_tab, w.proofblk, w.supd, d_status);
    return (int)hipGetLastError();
}
int launch_hh_chain(const uint8_t *d_keys, uint64_t stride, uint32_t n, uint32_t depth, const HhTables *d_tab, const HhWork &w,
                    const int32_t *d_status, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_hh_chain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TAIL_LDS);
    if (e != hipSuccess) return (int)e;
    // one round of workgroups over the chip: as many lanes per workgroup as that takes (the tables are 128 KiB, one workgroup per CU)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint64_t lanes = 16ull * n;
    uint32_t threads = (uint32_t)(((lanes + cus - 1) / cus + 255) / 256 * 256);
    threads = threads < 256 ? 256 : threads > 1024 ? 1024 : threads;
    hipLaunchKernelGGL(k_hh_chain, dim3((uint32_t)((lanes + threads - 1) / threads)), dim3(threads), TAIL_LDS, stream, d_keys, stride, n, depth, d_tab, w.tpc,
                       w.proofblk, d_status);
    return (int)hipGetLastError();
}
int launch_hh_final(const uint8_t *d_keys, uint64_t stride, uint32_t n, uint32_t depth, const HhWork &w, uint8_t *d_tags,
                    const int32_t *d_status, hipStream_t stream) {
    hipLaunchKernelGGL(k_hh_final, dim3((n + 31) / 32), dim3(64), 0, stream, d_keys, stride, n, depth, w.proofblk, w.supd, w.buckg, d_tags,
                       d_status);
    return (int)hipGetLastError();
}

}  // namespace wh
