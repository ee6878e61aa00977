// Shared between the heavy-hitter kernels (vdpf.hip) and the C-ABI layer (api.hip).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace wh {

// Bucket<u32> = 256 data + sign_data + ctr (hhcore/src/bucket.rs:6-25)
constexpr uint32_t HH_BKT = 258;
constexpr uint32_t HH_SUPPORT = 257;      // STRING_SIZE + 1 support keys (protocol.rs:95)
constexpr uint32_t HH_SUPPORT_LEN = 111;  // bincode VDPFKey<u32>, one level
constexpr uint32_t HH_MAX_DEPTH = 11;     // num_buckets <= 2048 (server-hh/src/main.rs:428-436)

// byte offsets inside one bincode KeyChain<u32> (serialize/src/lib.rs:72-80; hhcore/src/protocol.rs:10-15,
// vdpf.rs:22-29, dpf.rs:5-14, bucket.rs:5-10); d = levels of the main key
struct HhLayout {
    uint32_t key_id, root, cw_len, cw, cs, data_len, oc, sup_len, sup, total;
    __host__ __device__ explicit HhLayout(uint32_t d) {
        key_id = 16; root = 17; cw_len = 33; cw = 41; cs = 41 + 18 * d; data_len = cs + 64; oc = data_len + 8;
        sup_len = oc + 4 * HH_BKT; sup = sup_len + 8; total = sup + HH_SUPPORT * HH_SUPPORT_LEN;
    }
};

// constant tables in HBM, read through scalar loads
struct HhTables {
    uint32_t te0[256];   // AES table: bytes (2s, s, s, 3s)
    uint32_t rk[3][44];  // round keys: [0] key 0^16 (FixedKeyPrgStream), [1] key [1;16] (HasherStream), [2] verify key
    uint32_t cx[2][8];   // hashed index x in {0,1}: MMO_0([x,0,...]) then MMO_0 of that (mmo_hash2to4 blocks 0 and 2)
};

// per-sub-batch scratch in HBM
struct HhWork {
    uint4 *tpc;          // [cap][2^d][4]  corrected leaf hashes (tilde pi) of the main keys
    uint32_t *proofblk;  // [cap][258][16] SHA-256 message blocks: hex of the main proof, then of each support proof
    uint32_t *supd;      // [cap][257][2]  the two outputs of every support key
    uint32_t *buckg;     // [cap][258]     sum of the main key's outputs over all leaves
};

// atomic sketch accumulation of k_hh_main (vdpf.hip: HhSketchOut); sketch == NULL: none
struct HhSketch {
    uint32_t *sketch = nullptr;     // [sketches][2^depth * 258]
    uint64_t group_size = 1;        // key chain g adds into sketch g / group_size
    uint64_t first = 0;             // index of the launch's first key chain in the caller's array (kidx == NULL)
    const uint32_t *kidx = nullptr; // launch slot -> key chain index (d_keys then is the whole array)
    uint32_t flip = 0;              // 1: subtract
    bool buckets_only = false;      // a subtract launch: no leaf hashes, no per-key sums
};
// the four launches of one sub-batch, in this order on one stream; each returns a hipError_t as int.  d_keys is the
// launch's first key chain (kidx == NULL) or the whole array (kidx != NULL); d_dat, d_ids, d_status may be NULL.
int launch_hh_main(const uint8_t *d_keys, uint64_t stride, uint32_t n, uint32_t depth, const HhTables *d_tab, const HhWork &w,
                   uint32_t *d_dat, uint8_t *d_ids, int32_t *d_status, const HhSketch &sk, hipStream_t stream);

int launch_hh_support(const uint8_t *d_keys, uint64_t stride, uint32_t n, uint32_t depth, const HhTables *d_tab, const HhWork &w,
                      const int32_t *d_status, hipStream_t stream);
int launch_hh_chain(const uint8_t *d_keys, uint64_t stride, uint32_t n, uint32_t depth, const HhTables *d_tab, const HhWork &w,
                    const int32_t *d_status, hipStream_t stream);
int launch_hh_final(const uint8_t *d_keys, uint64_t stride, uint32_t n, uint32_t depth, const HhWork &w, uint8_t *d_tags,
                    const int32_t *d_status, hipStream_t stream);
size_t hh_main_lds_bytes(uint32_t depth);

}  // namespace wh
