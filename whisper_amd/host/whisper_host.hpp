// Host side of the drop-in, in C++ (the reference's host language, Rust, has no toolchain in this image):
// the server-side control flow around the C ABI of include/whisper_gfx950.h, with the reference's names.
//
//   ClientProofTag, general_binary_split_test   common/src/grouptest.rs:22-34, 123-371
//   run_vdaf_prepare                            server-batch-prio3/src/main.rs:65-125
//   run_hh_expand_and_aggregate                 server-hh/src/main.rs:53-110, 255-282 (one repetition)
//   get_sign_and_bkt, bytes_to_bucket, recover  hhcore/src/lib.rs:15-24, utils.rs:43-66, bucket.rs:112-120,
//                                               countsketch.rs:30-46
//   Peer::exchange_message                      bridge/src/mpc_conn.rs (TCP there; an in-process rendezvous here,
//                                               as the reference's own test uses two tasks on localhost)
//
// Nothing here computes on reports: tags, testing ids and the aggregate come from the GPU library.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/whisper_gfx950.h"
#include "../csrc/keccak.hpp"

namespace whisper_host {

typedef unsigned __int128 u128;

// grouptest.rs:22-34, ClientProofTag<P>.  TAG = 16: `tag` is the encoding of Prio3BatchedOutputShare<Field128> and batch
// proofs are 16 bytes of SHAKE128 (tag_size 16, server-batch-prio3/src/main.rs:99); TAG = 64: `tag` is the heavy-hitter
// proof String (64 hex characters) and batch proofs are SHA-256 (tag_size 32, server-hh/src/main.rs:61-68).
template <size_t TAG> struct ClientProofTagT {
    u128 testing_id;
    uint8_t tag[TAG];
    bool operator==(const ClientProofTagT &o) const { return testing_id == o.testing_id && !memcmp(tag, o.tag, TAG); }
    bool operator!=(const ClientProofTagT &o) const { return !(*this == o); }
};
template <size_t TAG> constexpr size_t digest_len() { return TAG == 16 ? 16 : 32; }

// grouptest.rs:100-105
template <size_t TAG> struct GroupTestMessageT {
    u128 lower = 0, upper = 0;
    bool has_singleton = false, has_batch = false;
    ClientProofTagT<TAG> singleton{};
    uint8_t batch_proof[32] = {0};
};
template <size_t TAG> struct RoundMessageT {   // what one exchange_message carries
    std::vector<GroupTestMessageT<TAG>> msgs;
    std::vector<u128> corrections;
    uint8_t first[32] = {0};
};

// Two parties meeting in memory: exchange_message blocks until the other side has called it too.
template <size_t TAG> class RendezvousT {
    typedef RoundMessageT<TAG> RoundMessage;
    std::mutex mu;
    std::condition_variable cv;
    RoundMessage slot[2][2];               // [round parity][party]: the late reader of round g must not see round g + 1
    int present = 0, generation = 0;
public:
    RoundMessage exchange(int who, const RoundMessage &mine) {
        std::unique_lock<std::mutex> lk(mu);
        const int gen = generation;
        slot[gen & 1][who] = mine;
        if (++present == 2) { present = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
        return slot[gen & 1][1 - who];
    }
};
template <size_t TAG> struct PeerT {
    RendezvousT<TAG> *r;
    int who;                               // 0 = Alice, 1 = Bob
    RoundMessageT<TAG> exchange_message(const RoundMessageT<TAG> &m) const { return r->exchange(who, m); }
};
typedef ClientProofTagT<16> ClientProofTag;
typedef RendezvousT<16> Rendezvous;
typedef PeerT<16> Peer;

// SHA-256 (FIPS 180-4) for the heavy-hitter batch proofs and the sketch's bucket hash; `sha2 = "0.10"` in the reference
struct Sha256 {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t buf[64];
    uint64_t total = 0;
    static uint32_t rr(uint32_t v, int n) { return (v >> n) | (v << (32 - n)); }
    void block(const uint8_t *b) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
            0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
            0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
            0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
            0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
            0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
            0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64], s[8];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)b[4 * i] << 24 | (uint32_t)b[4 * i + 1] << 16 | (uint32_t)b[4 * i + 2] << 8 | b[4 * i + 3];
        for (int i = 16; i < 64; i++)
            w[i] = w[i - 16] + w[i - 7] + (rr(w[i - 15], 7) ^ rr(w[i - 15], 18) ^ (w[i - 15] >> 3)) + (rr(w[i - 2], 17) ^ rr(w[i - 2], 19) ^ (w[i - 2] >> 10));
        memcpy(s, h, 32);
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = s[7] + (rr(s[4], 6) ^ rr(s[4], 11) ^ rr(s[4], 25)) + ((s[4] & s[5]) ^ (~s[4] & s[6])) + K[i] + w[i];
            const uint32_t t2 = (rr(s[0], 2) ^ rr(s[0], 13) ^ rr(s[0], 22)) + ((s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]));
            for (int k = 7; k > 0; k--) s[k] = s[k - 1];
            s[4] += t1; s[0] = t1 + t2;
        }
        for (int i = 0; i < 8; i++) h[i] += s[i];
    }
    void update(const uint8_t *p, size_t n) {
        for (size_t i = 0; i < n; i++) {
            buf[total++ & 63] = p[i];
            if ((total & 63) == 0) block(buf);
        }
    }
    void finish(uint8_t out[32]) {
        const uint64_t bits = total * 8;
        const uint8_t pad = 0x80, zero = 0;
        update(&pad, 1);
        while ((total & 63) != 56) update(&zero, 1);
        uint8_t len[8];
        for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(len, 8);
        for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
    }
};

// grouptest.rs:43-99: hash of verify_key || tags while testing_id <= upper -- SHAKE128 to 16 bytes (batch_proof_shake128)
// for 16-byte tags, SHA-256 (batch_proof_sha256) otherwise
template <size_t TAG>
inline size_t batch_proof(const ClientProofTagT<TAG> *proofs, size_t n, const uint8_t verify_key[16], u128 upper, uint8_t out[32]) {
    size_t split = n;
    if (TAG == 16) {
        wh::Sponge s;
        s.reset();
        s.absorb(verify_key, 16);
        for (size_t i = 0; i < n; i++) {
            if (proofs[i].testing_id > upper) { split = i; break; }
            s.absorb(proofs[i].tag, TAG);
        }
        s.finalize();
        memset(out, 0, 32);
        memcpy(out, s.st, 16);
    } else {
        Sha256 s;
        s.update(verify_key, 16);
        for (size_t i = 0; i < n; i++) {
            if (proofs[i].testing_id > upper) { split = i; break; }
            s.update(proofs[i].tag, TAG);
        }
        s.finish(out);
    }
    return split;
}

template <size_t TAG>
inline size_t make_message(const ClientProofTagT<TAG> *sl, size_t n, u128 lower, u128 upper, const uint8_t vk[16], GroupTestMessageT<TAG> &m) {
    m = GroupTestMessageT<TAG>{};
    m.lower = lower; m.upper = upper;
    uint8_t pf[32];
    const size_t idx = batch_proof(sl, n, vk, upper, pf);
    if (idx == 1) { m.has_singleton = true; m.singleton = sl[0]; }
    else if (idx > 1) { m.has_batch = true; memcpy(m.batch_proof, pf, 32); }
    return idx;
}

// grouptest.rs:123-371: one party's run; `proofs` sorted by testing id.  Returns the ids this party rejects.
template <size_t TAG>
inline std::set<u128> general_binary_split_test(const std::vector<ClientProofTagT<TAG>> &proofs, const uint8_t verify_key[16],
                                                const PeerT<TAG> &peer, size_t d, size_t *rounds = nullptr) {
    typedef ClientProofTagT<TAG> ClientProofTag;
    typedef GroupTestMessageT<TAG> GroupTestMessage;
    typedef RoundMessageT<TAG> RoundMessage;
    struct Slice { const ClientProofTag *p; size_t n; };
    std::set<u128> result;
    size_t nrounds = 1;
    const u128 U128_MAX = ~(u128)0;
    RoundMessage mine, theirs;
    batch_proof(proofs.data(), proofs.size(), verify_key, U128_MAX, mine.first);
    theirs = peer.exchange_message(mine);
    if (!memcmp(mine.first, theirs.first, 32)) { if (rounds) *rounds = nrounds; return result; }
    d = std::max<size_t>(d, 2);
    const u128 q = U128_MAX / d, r = U128_MAX % d;
    std::vector<GroupTestMessage> msgs;
    std::vector<Slice> slices;
    std::vector<u128> corrections;
    {
        const ClientProofTag *rest = proofs.data();
        size_t left = proofs.size();
        u128 lower = 0;
        for (size_t i = 0; i < d; i++) {
            const u128 upper = i < r ? (q + 1) * (i + 1) : r * (q + 1) + (i + 1 - r) * q;
            GroupTestMessage m;
            const size_t idx = make_message(rest, left, lower, upper, verify_key, m);
            msgs.push_back(m);
            slices.push_back({rest, left});
            rest += idx; left -= idx;
            lower = upper;
        }
    }
    // The reference loops while it has messages or corrections of its own (grouptest.rs:216); with the
    // rendezvous both parties must agree on the number of rounds, so a round also follows whenever a
    // correction may be on its way in either direction (one side sent a singleton, the other a batch).
    bool more = true;
    while (more) {
        nrounds++;
        mine = RoundMessage{};
        mine.msgs = msgs; mine.corrections = corrections;
        theirs = peer.exchange_message(mine);
        for (u128 c : theirs.corrections) result.insert(c);
        if (theirs.msgs.size() != msgs.size()) throw std::runtime_error("group test: the peers disagree on the open ranges");
        std::vector<GroupTestMessage> nm;
        std::vector<Slice> ns;
        std::vector<u128> nc;
        bool correction_possible = false;
        for (size_t t = 0; t < msgs.size(); t++) {
            const GroupTestMessage &a = msgs[t], &b = theirs.msgs[t];
            const Slice sl = slices[t];
            if (a.lower != b.lower || a.upper != b.upper) throw std::runtime_error("group test: range mismatch");
            const bool a_single = a.has_singleton && !a.has_batch, a_batch = !a.has_singleton && a.has_batch;
            const bool b_single = b.has_singleton && !b.has_batch, b_batch = !b.has_singleton && b.has_batch;
            const bool b_none = !b.has_singleton && !b.has_batch;
            correction_possible |= (a_single && b_batch) || (a_batch && b_single);
            if (a_single) {
                if (b_none) result.insert(a.singleton.testing_id);
                else if (b_single && b.singleton != a.singleton) result.insert(a.singleton.testing_id);
            } else if (a_batch) {
                if (b_single) {
                    for (size_t i = 0; i < sl.n; i++) {
                        if (b.singleton.testing_id == sl.p[i].testing_id) {
                            if (!memcmp(b.singleton.tag, sl.p[i].tag, TAG)) continue;
                            nc.push_back(sl.p[i].testing_id);
                        }
                        result.insert(sl.p[i].testing_id);
                    }
                } else if (b_none) {
                    for (size_t i = 0; i < sl.n; i++) result.insert(sl.p[i].testing_id);
                } else if (b_batch && memcmp(a.batch_proof, b.batch_proof, 32)) {
                    const u128 mid = a.lower + ((a.upper - a.lower) >> 1);
                    GroupTestMessage left, right;
                    const size_t mid_idx = make_message(sl.p, sl.n, a.lower, mid, verify_key, left);
                    make_message(sl.p + mid_idx, sl.n - mid_idx, mid, a.upper, verify_key, right);
                    nm.push_back(left); nm.push_back(right);
                    ns.push_back({sl.p, mid_idx}); ns.push_back({sl.p + mid_idx, sl.n - mid_idx});
                }
            }
        }
        msgs.swap(nm); slices.swap(ns); corrections.swap(nc);
        more = !msgs.empty() || correction_possible;
    }
    if (rounds) *rounds = nrounds;
    return result;
}

struct PrepareOutcome {
    std::vector<uint8_t> aggregate;        // AggregateShare<Field128>::get_encoded(), output_len * 16 bytes
    size_t clients_passed = 0;
    std::set<u128> bad_set;
};

// server-batch-prio3/src/main.rs:65-125 with the map over prepare_encoded_key and the merge loop on the GPU.
// The reference unwrap()s every decode / query error; here such a report throws, as the caller asked for.
inline PrepareOutcome run_vdaf_prepare(wh_ctx *ctx, const wh_prio3_sizes &sizes, const uint8_t verify_key[16],
                                       const std::vector<std::vector<uint8_t>> &vdaf_keys, const Peer &peer,
                                       size_t num_bad_clients) {
    const size_t n = vdaf_keys.size();
    std::vector<uint64_t> offsets(n + 1, 0);
    for (size_t i = 0; i < n; i++) offsets[i + 1] = offsets[i] + vdaf_keys[i].size();
    std::vector<uint8_t> blob(offsets[n] + 1);
    for (size_t i = 0; i < n; i++) memcpy(blob.data() + offsets[i], vdaf_keys[i].data(), vdaf_keys[i].size());
    std::vector<uint8_t> tags(16 * n), ids(16 * n);
    std::vector<int32_t> status(n);
    int32_t rc = wh_prio3_prepare_batch(ctx, verify_key, blob.data(), offsets.data(), n, tags.data(), ids.data(), status.data());
    if (rc) throw std::runtime_error(std::string("wh_prio3_prepare_batch: ") + wh_last_error(ctx));
    for (size_t i = 0; i < n; i++)
        if (status[i]) throw std::runtime_error("report " + std::to_string(i) + " does not decode / verify (status " + std::to_string(status[i]) + ")");
    // sort by testing id (main.rs:93), remembering where each report sits in the batch
    std::vector<uint32_t> order(n);
    auto id_of = [&](size_t i) { u128 v; memcpy(&v, ids.data() + 16 * i, 16); return v; };
    rc = wh_sort_by_testing_id(ctx, ids.data(), n, order.data());
    if (rc) throw std::runtime_error(std::string("wh_sort_by_testing_id: ") + wh_last_error(ctx));
    std::vector<ClientProofTag> proof_tags(n);
    for (size_t k = 0; k < n; k++) {
        proof_tags[k].testing_id = id_of(order[k]);
        memcpy(proof_tags[k].tag, tags.data() + 16 * order[k], 16);
    }
    PrepareOutcome out;
    out.bad_set = general_binary_split_test(proof_tags, verify_key, peer, num_bad_clients);
    // merge of the out shares whose testing id is not in the bad set (main.rs:108-118)
    std::vector<uint8_t> keep(n);
    for (size_t i = 0; i < n; i++) {
        keep[i] = out.bad_set.count(id_of(i)) ? 0 : 1;
        out.clients_passed += keep[i];
    }
    out.aggregate.resize(16 * (size_t)sizes.output_len);
    rc = wh_prio3_aggregate(ctx, keep.data(), out.aggregate.data());
    if (rc) throw std::runtime_error(std::string("wh_prio3_aggregate: ") + wh_last_error(ctx));
    return out;
}

// ---- heavy hitters ----
typedef ClientProofTagT<64> HhProofTag;
constexpr size_t BUCKET_WORDS = 258;       // STRING_SIZE + 2 (hhcore/src/bucket.rs:12-20)

// hhcore/src/lib.rs:15-24
inline void get_sign_and_bkt(const uint8_t *v, size_t len, size_t num_buckets, uint16_t rep_count, bool &sign, uint16_t &bkt) {
    Sha256 s;
    s.update(v, len);
    const uint8_t rc[2] = {(uint8_t)rep_count, (uint8_t)(rep_count >> 8)};
    s.update(rc, 2);
    uint8_t h[32];
    s.finish(h);
    uint64_t x = 0;
    for (int i = 0; i < 8; i++) x |= (uint64_t)h[3 + i] << (8 * i);
    sign = h[0] & 1;
    bkt = (uint16_t)(x % num_buckets);
}
// hhcore/src/utils.rs:43-66: counters +-1 per bit (bits of a byte LSB first), flipped when sign is false
inline void bytes_to_bucket(const uint8_t s[32], bool sign, uint32_t out[BUCKET_WORDS]) {
    const uint32_t one = sign ? 1u : 0xffffffffu, minus = sign ? 0xffffffffu : 1u;
    for (int i = 0; i < 256; i++) out[i] = ((s[i >> 3] >> (i & 7)) & 1) ? one : minus;
    out[256] = one;
    out[257] = 1;
}
// CountSketch::recover (countsketch.rs:30-46) with Bucket::recover (bucket.rs:112-120); u32 is "positive" below 2^16
inline std::set<std::vector<uint8_t>> recover(const uint32_t *sketch, size_t num_buckets, uint32_t threshold, uint16_t rep_count) {
    std::set<std::vector<uint8_t>> out;
    for (size_t i = 0; i < num_buckets; i++) {
        const uint32_t *b = sketch + i * BUCKET_WORDS;
        if (!(b[257] > threshold)) continue;
        const bool pos = b[256] < (1u << 16);
        std::vector<uint8_t> str(32, 0);
        for (int k = 0; k < 256; k++) {
            const bool bit = (b[k] < (1u << 16)) == pos;
            str[k >> 3] |= (uint8_t)bit << (k & 7);
        }
        bool sign; uint16_t idx;
        get_sign_and_bkt(str.data(), 32, num_buckets, rep_count, sign, idx);
        if (idx == i) out.insert(str);
    }
    return out;
}

struct HhOutcome {
    std::vector<uint32_t> sketch;          // num_buckets * 258 counters: the kept clients' bucket vectors summed
    std::set<u128> bad_set;
};
// One repetition of batch_collect_and_aggregate_per_run + aggregate_hhs (server-hh/src/main.rs:53-110, 255-282): key chains
// as the bincode bytes received, expansion and sums on the GPU, the group test in between on the host.
inline HhOutcome run_hh_expand_and_aggregate(wh_ctx *ctx, const uint8_t verify_key[16], const std::vector<std::vector<uint8_t>> &key_chains,
                                             uint32_t depth, const PeerT<64> &peer, size_t num_bad_clients) {
    const size_t n = key_chains.size(), stride = wh_hh_key_chain_len(depth);
    std::vector<uint8_t> keys(n * stride + 1), tags(64 * n), ids(16 * n);
    for (size_t i = 0; i < n; i++) {
        if (key_chains[i].size() != stride) throw std::runtime_error("key chain " + std::to_string(i) + " has the wrong length");
        memcpy(keys.data() + i * stride, key_chains[i].data(), stride);
    }
    std::vector<int32_t> status(n);
    int32_t rc = wh_hh_eval_and_check_batch(ctx, verify_key, keys.data(), stride, n, depth, tags.data(), ids.data(), status.data());
    if (rc) throw std::runtime_error(std::string("wh_hh_eval_and_check_batch: ") + wh_last_error(ctx));
    for (size_t i = 0; i < n; i++)
        if (status[i]) throw std::runtime_error("key chain " + std::to_string(i) + " does not decode (status " + std::to_string(status[i]) + ")");
    std::vector<uint32_t> order(n);
    auto id_of = [&](size_t i) { u128 v; memcpy(&v, ids.data() + 16 * i, 16); return v; };
    rc = wh_sort_by_testing_id(ctx, ids.data(), n, order.data());                                     // main.rs:266
    if (rc) throw std::runtime_error(std::string("wh_sort_by_testing_id: ") + wh_last_error(ctx));
    std::vector<HhProofTag> proofs(n);
    for (size_t k = 0; k < n; k++) {
        proofs[k].testing_id = id_of(order[k]);
        memcpy(proofs[k].tag, tags.data() + 64 * order[k], 64);
    }
    HhOutcome out;
    out.bad_set = general_binary_split_test(proofs, verify_key, peer, num_bad_clients);
    std::vector<uint8_t> keep(n);
    for (size_t i = 0; i < n; i++) keep[i] = out.bad_set.count(id_of(i)) ? 0 : 1;
    out.sketch.resize(((size_t)BUCKET_WORDS) << depth);
    rc = wh_hh_aggregate(ctx, keep.data(), out.sketch.data());
    if (rc) throw std::runtime_error(std::string("wh_hh_aggregate: ") + wh_last_error(ctx));
    return out;
}

}  // namespace whisper_host
