// Device-visible description of one Prio3 instance and of the per-batch workspace.
// Sizes follow batched-flpcp-rs/src/flp/types.rs (SumVec 478-677, Histogram 318-471,
// Average 213-313) and flp.rs:749-758; the report layout follows VdafBatchedKey::encode
// (vdaf.rs:148-162).
#pragma once
#include "f128.hpp"

namespace wh {

enum : uint32_t { CIRCUIT_SUMVEC = 0, CIRCUIT_HISTOGRAM = 1, CIRCUIT_AVERAGE = 2 };

// status codes (include/whisper_gfx950.h)
enum : int32_t {
    ST_OK = 0, ST_ERR_PARAM = -1, ST_ERR_SHORT = -2, ST_ERR_MODULUS = -3, ST_ERR_AGG_ID = -4,
    ST_ERR_NUM_QUERIES = -5, ST_ERR_QUERY_RAND = -6, ST_ERR_DECIDE = -7, ST_ERR_JOINT_RAND = -8
};

// Prio3 usage constants (vdaf/prio3.rs:69-78)
enum : uint32_t {
    DST_MEAS = 1, DST_PROOF = 2, DST_JOINT_RAND = 3, DST_PROVE_RAND = 4, DST_QUERY_RAND = 5,
    DST_JR_SEED = 6, DST_JR_PART = 7, DST_QR_PART = 8, DST_RLC = 9, DST_HASH = 10
};

struct Prio3Dev {
    uint32_t circuit, bits, len, chunk;
    uint32_t input_len, proof_len, verifier_len, output_len;
    uint32_t jr_len, calls, m, log2m, arity, gpoly_len;
    uint32_t alg_id;
    uint32_t leader_len, helper_len;  // encoded report sizes
    uint32_t n_weights;               // calls + 1
    uint32_t num_aggregators;
    uint32_t mode;                    // 0: whisper's batched reports; 1: draft-07 VdafKey (prepare_init); 2: bare FLP query
    uint32_t ver_stride;              // elements per row of the verifier output (verifier_len, + 1 with the joint-rand part)
    F128 half;                        // num_shares^-1 = 2^-1
    F128 m_inv;                       // m^-1
    F128 sampler_bound;               // p (test hook: a smaller bound makes the sampler's skip branch run)
    const F128 *omega;                // omega_m^j, j = 0..m-1 (device)
    const F128 *s_tab;                // sum_{j=1..calls} omega_m^{ij}, i = 0..gpoly_len-1 (device)
};

// Per-report descriptor written by the parse kernel.
struct RepDesc {
    const uint8_t *base;    // first byte of the encoded report
    const uint8_t *meas;    // input_len elements (report bytes for a Leader, expanded row for a Helper)
    const uint8_t *proof1;  // proof_len elements, run 1
    const uint8_t *proof2;  // proof_len elements, run 2
    const uint8_t *tail;    // public_share_second | num_queries | pp0 | pp1 | qr_blind | nonce
    const uint8_t *jr_blind;
    const uint8_t *seeds;   // Helper: measurement-share seed; the proof-share seeds follow at P.mode's offsets
    uint32_t role;          // agg_id
    uint32_t hslot;         // helper slot in the expansion rows
};

// Per-report scalars handed from the hash kernels to the query kernel.
struct RepScal {
    F128 jr[4];     // joint_rand_fused (2 * jr_len used)
    F128 qr[2];     // query_rand_fused
    F128 hash;      // hash of the public parts as a field element
    F128 own;       // this aggregator's joint-rand / query-rand part while its seed is derived
    uint32_t rand_diff[2];
    uint32_t pad[2];
};

// Per-report powers and sums for the ParallelSum query kernel, computed once per report by the
// lane that derives the randomness (a wave-uniform value costs a whole wavefront there).
struct RepPow {
    F128 L0, h_sumL;    // L_0(r); h * sum_{j>=1} L_j(r)
    F128 rho_lo[8];     // rho^(t+1), t = 0..7
    F128 rho_hi[8];     // rho^(8t)
    F128 rho64[5];      // rho^(64t), t = 0..4
    F128 r_lo[8];       // r^t
    F128 r_hi[8];       // r^(8t)
    F128 r64;           // r^64
};

struct Workspace {
    RepDesc *desc;       // [cap]
    RepScal *scal;       // [cap]
    F128 *weights;       // [cap][n_weights]   barycentric L_j(r) for run 2
    uint32_t *wlimbs;    // [cap][calls][12]   ParallelSum circuits: 26-bit limbs of L_j and of
                         //                    M_j = L_j rho^(chunk (j-1)), j = 1..calls (5 + 5 + 2 pad)
    RepPow *pows;        // [cap]
    F128 *rlc;           // [cap][verifier_len + 5]
    uint8_t *hrows;      // [cap][(input_len + 2 proof_len) * 16] expanded Helper shares
    uint32_t *hlist;     // [cap] report index of each Helper slot
    uint32_t *llist;     // [cap] report indices of the Leader-role reports
    uint32_t *dirty;     // [cap] Helper reports whose share stream had a rejected chunk (k_meas), redone the general way
    uint32_t *counters;  // [0] = number of Helpers, [1] = number of Leaders (well-formed reports only), [2] = dirty
    uint64_t hrow_bytes;
};

}  // namespace wh
