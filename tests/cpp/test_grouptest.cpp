// The reference's own group-testing scenario (common/src/grouptest.rs:373-447): 300 tags, three corrupted and two
// deleted on Bob's side; Alice must reject all five ids, Bob the three he still holds.  CPU only.
#include <cstdio>
#include <thread>
#include "../../whisper_amd/host/whisper_host.hpp"
using namespace whisper_host;

static u128 splitmix(uint64_t &x) {
    auto next = [&] { uint64_t z = (x += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
    u128 v = next(); v = (v << 64) | next(); return v;
}
int main() {
    uint64_t seed = 7;
    std::vector<u128> ids(300);
    for (auto &v : ids) v = splitmix(seed);
    std::sort(ids.begin(), ids.end());
    std::vector<ClientProofTag> p0(300), p1;
    for (int i = 0; i < 300; i++) { p0[i].testing_id = ids[i]; memset(p0[i].tag, 0, 16); snprintf((char *)p0[i].tag, 16, "%d", i); }
    p1 = p0;
    for (int i : {20, 68, 100}) memcpy(p1[i].tag, "INCONSISTENCY", 14);
    p1.erase(p1.begin() + 200); p1.erase(p1.begin() + 4);
    uint8_t vk[16]; memset(vk, 1, 16);
    // (with many more ranges than defects the reference over-rejects: its first-round slices are open-ended,
    //  grouptest.rs:185-211 `cur_slices.push(slice_so_far)`; mirrored as is, so only the reference's own d is asserted)
    for (size_t d : {5u, 1u}) {
        Rendezvous rv;
        std::set<u128> bad0, bad1; size_t r0 = 0, r1 = 0;
        std::thread ta([&] { bad0 = general_binary_split_test(p0, vk, Peer{&rv, 0}, d, &r0); });
        std::thread tb([&] { bad1 = general_binary_split_test(p1, vk, Peer{&rv, 1}, d, &r1); });
        ta.join(); tb.join();
        std::set<u128> want1 = {ids[20], ids[68], ids[100]}, want0 = want1;
        want0.insert(ids[200]); want0.insert(ids[4]);
        if (bad0 != want0 || bad1 != want1 || r0 != r1) { printf("FAIL d=%zu: |bad0|=%zu |bad1|=%zu rounds %zu/%zu\n", d, bad0.size(), bad1.size(), r0, r1); return 1; }
        printf("d=%zu: ok, %zu rounds\n", d, r0);
    }
    {   // equal inputs: one round, nothing rejected
        Rendezvous rv; std::set<u128> a, b; size_t r0 = 0, r1 = 0;
        std::thread ta([&] { a = general_binary_split_test(p0, vk, Peer{&rv, 0}, 5, &r0); });
        std::thread tb([&] { b = general_binary_split_test(p0, vk, Peer{&rv, 1}, 5, &r1); });
        ta.join(); tb.join();
        if (!a.empty() || !b.empty() || r0 != 1 || r1 != 1) { printf("FAIL equal inputs\n"); return 1; }
    }
    printf("OK\n");
    return 0;
}
