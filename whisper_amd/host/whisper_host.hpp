// Host side of the drop-in, in C++ (the reference's host language, Rust, has no toolchain in this image):
// the server-side control flow around the C ABI of include/whisper_gfx950.h, with the reference's names.
//
//   ClientProofTag, general_binary_split_test   common/src/grouptest.rs:22-34, 123-371
//   run_vdaf_prepare                            server-batch-prio3/src/main.rs:65-125
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

// grouptest.rs:22-34.  `tag` is the 16-byte encoding of Prio3BatchedOutputShare<Field128>.
struct ClientProofTag {
    u128 testing_id;
    uint8_t tag[16];
    bool operator==(const ClientProofTag &o) const { return testing_id == o.testing_id && !memcmp(tag, o.tag, 16); }
    bool operator!=(const ClientProofTag &o) const { return !(*this == o); }
};

// grouptest.rs:100-105
struct GroupTestMessage {
    u128 lower = 0, upper = 0;
    bool has_singleton = false, has_batch = false;
    ClientProofTag singleton{};
    uint8_t batch_proof[16] = {0};
};
struct RoundMessage {                      // what one exchange_message carries
    std::vector<GroupTestMessage> msgs;
    std::vector<u128> corrections;
    uint8_t first[16] = {0};
};

// Two parties meeting in memory: exchange_message blocks until the other side has called it too.
class Rendezvous {
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
struct Peer {
    Rendezvous *r;
    int who;                               // 0 = Alice, 1 = Bob
    RoundMessage exchange_message(const RoundMessage &m) const { return r->exchange(who, m); }
};

// grouptest.rs:82-99 (batch_proof_shake128): SHAKE128(verify_key || tags while testing_id <= upper) -> 16 bytes
inline size_t batch_proof(const ClientProofTag *proofs, size_t n, const uint8_t verify_key[16], u128 upper, uint8_t out[16]) {
    wh::Sponge s;
    s.reset();
    s.absorb(verify_key, 16);
    size_t split = n;
    for (size_t i = 0; i < n; i++) {
        if (proofs[i].testing_id > upper) { split = i; break; }
        s.absorb(proofs[i].tag, 16);
    }
    s.finalize();
    memcpy(out, s.st, 16);
    return split;
}

inline size_t make_message(const ClientProofTag *sl, size_t n, u128 lower, u128 upper, const uint8_t vk[16], GroupTestMessage &m) {
    m = GroupTestMessage{};
    m.lower = lower; m.upper = upper;
    uint8_t pf[16];
    const size_t idx = batch_proof(sl, n, vk, upper, pf);
    if (idx == 1) { m.has_singleton = true; m.singleton = sl[0]; }
    else if (idx > 1) { m.has_batch = true; memcpy(m.batch_proof, pf, 16); }
    return idx;
}

// grouptest.rs:123-371: one party's run; `proofs` sorted by testing id.  Returns the ids this party rejects.
inline std::set<u128> general_binary_split_test(const std::vector<ClientProofTag> &proofs, const uint8_t verify_key[16],
                                                const Peer &peer, size_t d, size_t *rounds = nullptr) {
    struct Slice { const ClientProofTag *p; size_t n; };
    std::set<u128> result;
    size_t nrounds = 1;
    const u128 U128_MAX = ~(u128)0;
    RoundMessage mine, theirs;
    batch_proof(proofs.data(), proofs.size(), verify_key, U128_MAX, mine.first);
    theirs = peer.exchange_message(mine);
    if (!memcmp(mine.first, theirs.first, 16)) { if (rounds) *rounds = nrounds; return result; }
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
                            if (!memcmp(b.singleton.tag, sl.p[i].tag, 16)) continue;
                            nc.push_back(sl.p[i].testing_id);
                        }
                        result.insert(sl.p[i].testing_id);
                    }
                } else if (b_none) {
                    for (size_t i = 0; i < sl.n; i++) result.insert(sl.p[i].testing_id);
                } else if (b_batch && memcmp(a.batch_proof, b.batch_proof, 16)) {
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
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    auto id_of = [&](size_t i) { u128 v; memcpy(&v, ids.data() + 16 * i, 16); return v; };
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return id_of(a) < id_of(b); });
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

}  // namespace whisper_host
