// Alice and Bob in one process, each running the host mirror of server-batch-prio3's run_vdaf_prepare
// (main.rs:65-125) over its own GPU context: clients from the CPU generator (client-batch-prio3/src/main.rs:67-153),
// some of them tampered on Bob's copy only (nonce[0] += 1, main.rs:146-149).  Checks what the reference's flow
// promises: both servers reject exactly the tampered clients, and the two aggregate shares unshard to the sum of
// the honest clients' measurements.  Needs a GPU.
#include <cstdio>
#include <thread>
#include "../../whisper_amd/host/whisper_host.hpp"
using namespace whisper_host;

static uint64_t sm(uint64_t &x) { uint64_t z = (x += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

int main() {
    const uint32_t LEN = 16, BITS = 16, CHUNK = 8, N = 200;
    const std::set<size_t> bad = {3, 77, 150};
    wh_prio3_params p = {WH_SUMVEC, BITS, LEN, CHUNK, 2, 0};
    wh_prio3_sizes z;
    if (wh_prio3_sizes_of(&p, &z)) { printf("FAIL sizes\n"); return 1; }
    uint64_t seed = 42;
    std::vector<uint64_t> meas(N * LEN);
    for (auto &m : meas) m = sm(seed) & 0xffff;
    std::vector<uint8_t> nonces(16 * N), rands(144 * N), lr((size_t)z.leader_report_len * N), hr((size_t)z.helper_report_len * N);
    for (auto &b : nonces) b = (uint8_t)sm(seed);
    for (auto &b : rands) b = (uint8_t)sm(seed);
    if (wh_client_shard_many(&p, meas.data(), LEN, nonces.data(), rands.data(), 1000, N, 4, lr.data(), hr.data())) { printf("FAIL shard\n"); return 1; }
    // client i sends its Leader share to Alice and its Helper share to Bob when i is even, the other way round when odd
    // (client-batch-prio3 main.rs:114-115); the nonce is the last 16 bytes of a report
    std::vector<std::vector<uint8_t>> alice(N), bob(N);
    for (size_t i = 0; i < N; i++) {
        std::vector<uint8_t> l(lr.begin() + i * z.leader_report_len, lr.begin() + (i + 1) * z.leader_report_len);
        std::vector<uint8_t> h(hr.begin() + i * z.helper_report_len, hr.begin() + (i + 1) * z.helper_report_len);
        alice[i] = (i & 1) ? h : l;
        bob[i] = (i & 1) ? l : h;
        if (bad.count(i)) bob[i][bob[i].size() - 16] += 1;
    }
    uint8_t vk[16];
    for (int i = 0; i < 16; i++) vk[i] = (uint8_t)(16 + i);
    wh_ctx *ca = nullptr, *cb = nullptr;
    if (wh_ctx_create(0, &p, &ca) || wh_ctx_create(0, &p, &cb)) { printf("FAIL ctx (no GPU?)\n"); return 1; }
    Rendezvous rv;
    PrepareOutcome oa, ob;
    std::string err;
    std::thread ta([&] { try { oa = run_vdaf_prepare(ca, z, vk, alice, Peer{&rv, 0}, bad.size()); } catch (std::exception &e) { err = e.what(); } });
    std::thread tb([&] { try { ob = run_vdaf_prepare(cb, z, vk, bob, Peer{&rv, 1}, bad.size()); } catch (std::exception &e) { err = e.what(); } });
    ta.join(); tb.join();
    if (!err.empty()) { printf("FAIL %s\n", err.c_str()); return 1; }
    if (oa.bad_set != ob.bad_set || oa.bad_set.size() != bad.size() || oa.clients_passed != N - bad.size() || ob.clients_passed != N - bad.size()) {
        printf("FAIL bad sets: %zu / %zu, passed %zu / %zu\n", oa.bad_set.size(), ob.bad_set.size(), oa.clients_passed, ob.clients_passed);
        return 1;
    }
    // Collector::unshard (prio3.rs:2909-2921)
    for (uint32_t e = 0; e < LEN; e++) {
        wh::F128 a, b;
        memcpy(&a, oa.aggregate.data() + 16 * e, 16); memcpy(&b, ob.aggregate.data() + 16 * e, 16);
        const wh::F128 s = wh::f_add(a, b);
        uint64_t want = 0;
        for (size_t i = 0; i < N; i++) if (!bad.count(i)) want += meas[i * LEN + e];
        if (s.hi != 0 || s.lo != want) { printf("FAIL aggregate element %u: %llu vs %llu\n", e, (unsigned long long)s.lo, (unsigned long long)want); return 1; }
    }
    wh_ctx_destroy(ca); wh_ctx_destroy(cb);
    printf("OK: %zu of %u clients passed on both servers, aggregate matches\n", oa.clients_passed, N);
    return 0;
}
