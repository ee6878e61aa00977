// Alice and Bob in one process, each running the host mirror of one server-hh repetition (server-hh/src/main.rs:53-110,
// 255-282) over its own GPU context: clients vote for 32-byte strings through a count sketch (client-hh/src/main.rs:39-64),
// key chains from the CPU client (wh_hh_gen_key_chain), a few clients malformed the way client-hh makes them (the two
// servers get key chains of different strings, main.rs:84-118).  Checks what the reference's flow promises: both servers
// reject exactly the malformed clients (group test over the String tags, SHA-256 batch proofs), the two sketches add up to
// the honest clients' buckets, and the heavy string is read back.  Needs a GPU.
#include <cstdio>
#include <thread>
#include "../../whisper_amd/host/whisper_host.hpp"
using namespace whisper_host;

static uint64_t sm(uint64_t &x) { uint64_t z = (x += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

int main() {
    const uint32_t DEPTH = 5, NB = 1u << DEPTH, N = 120;
    const uint16_t REP = 2;
    const std::set<size_t> bad = {7, 50, 119};
    uint64_t seed = 99;
    // population of 20 strings; string 0 is heavy (40 % of the clients)
    std::vector<std::vector<uint8_t>> pop(20, std::vector<uint8_t>(32));
    for (auto &s : pop) for (auto &b : s) b = (uint8_t)sm(seed);
    const size_t stride = wh_hh_key_chain_len(DEPTH);
    std::vector<std::vector<uint8_t>> alice(N, std::vector<uint8_t>(stride)), bob(N, std::vector<uint8_t>(stride)), scratch(2, std::vector<uint8_t>(stride));
    std::vector<size_t> pick(N);
    std::vector<uint32_t> want((size_t)NB * BUCKET_WORDS, 0);
    for (size_t c = 0; c < N; c++) {
        pick[c] = (sm(seed) % 100 < 40) ? 0 : 1 + sm(seed) % 19;
        uint8_t id[16] = {0}, rs[32];
        memcpy(id, &c, sizeof c);
        auto gen = [&](size_t str, uint8_t *k0, uint8_t *k1) {
            bool sign; uint16_t idx; uint32_t bk[BUCKET_WORDS];
            get_sign_and_bkt(pop[str].data(), 32, NB, REP, sign, idx);
            bytes_to_bucket(pop[str].data(), sign, bk);
            for (auto &b : rs) b = (uint8_t)sm(seed);
            if (wh_hh_gen_key_chain(DEPTH, id, idx, bk, rs, k0, k1)) { printf("FAIL gen\n"); exit(1); }
            return std::make_pair(idx, std::vector<uint32_t>(bk, bk + BUCKET_WORDS));
        };
        auto mine = gen(pick[c], alice[c].data(), bob[c].data());
        if (bad.count(c)) gen((pick[c] + 3) % pop.size(), scratch[0].data(), bob[c].data());
        else for (size_t k = 0; k < BUCKET_WORDS; k++) want[(size_t)mine.first * BUCKET_WORDS + k] += mine.second[k];
    }
    uint8_t vk[16];
    for (int i = 0; i < 16; i++) vk[i] = (uint8_t)(200 + i);
    wh_ctx *ca = nullptr, *cb = nullptr;
    if (wh_hh_ctx_create(0, &ca) || wh_hh_ctx_create(0, &cb)) { printf("FAIL ctx (no GPU?)\n"); return 1; }
    RendezvousT<64> rv;
    HhOutcome oa, ob;
    std::string err;
    std::thread ta([&] { try { oa = run_hh_expand_and_aggregate(ca, vk, alice, DEPTH, PeerT<64>{&rv, 0}, bad.size()); } catch (std::exception &e) { err = e.what(); } });
    std::thread tb([&] { try { ob = run_hh_expand_and_aggregate(cb, vk, bob, DEPTH, PeerT<64>{&rv, 1}, bad.size()); } catch (std::exception &e) { err = e.what(); } });
    ta.join(); tb.join();
    if (!err.empty()) { printf("FAIL %s\n", err.c_str()); return 1; }
    if (oa.bad_set != ob.bad_set || oa.bad_set.size() != bad.size()) { printf("FAIL bad sets: %zu / %zu\n", oa.bad_set.size(), ob.bad_set.size()); return 1; }
    // my_sketch.insert(received_sketch) (main.rs:386-391): counters add mod 2^32
    std::vector<uint32_t> merged(want.size());
    for (size_t k = 0; k < merged.size(); k++) merged[k] = oa.sketch[k] + ob.sketch[k];
    if (merged != want) { printf("FAIL merged sketch differs from the honest clients' buckets\n"); return 1; }
    const auto hh = recover(merged.data(), NB, (uint32_t)(N * 0.25), REP);
    if (hh.size() != 1 || *hh.begin() != pop[0]) { printf("FAIL recover: %zu strings\n", hh.size()); return 1; }
    wh_ctx_destroy(ca); wh_ctx_destroy(cb);
    printf("OK: %zu malformed clients rejected on both servers, sketches merge to the honest votes, heavy hitter recovered\n", bad.size());
    return 0;
}
