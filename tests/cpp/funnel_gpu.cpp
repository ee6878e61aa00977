// The caller funnel of include/whisper_gfx950.h: 32 threads (the reference's 32 tokio tasks, server-batch-prio3/src/main.rs:
// 282-335, each with the reports of its own socket) share TWO contexts on one GPU.  Every task brackets
// prepare -> aggregate -> combine with wh_ctx_lock / wh_ctx_unlock, so the tasks queue per context; what each task gets back
// must equal what the same batch gives on a context of its own, single-threaded.  Also drives the multi-GPU combine behind
// the C ABI with one rank: without RCCL (a lone GPU) and through RCCL (ncclCommInitRank with world = 1).  Needs a GPU.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/whisper_gfx950.h"

static uint64_t sm(uint64_t &x) { uint64_t z = (x += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

struct Batch {
    std::vector<uint8_t> blob; std::vector<uint64_t> off; std::vector<uint8_t> keep;
    std::vector<uint8_t> tags, ids, agg; std::vector<int32_t> status;
};

int main() {
    const uint32_t LEN = 16, BITS = 16, CHUNK = 8, TASKS = 32, ROUNDS = 3;
    wh_prio3_params p = {WH_SUMVEC, BITS, LEN, CHUNK, 2, 0};
    wh_prio3_sizes z;
    if (wh_prio3_sizes_of(&p, &z)) { printf("FAIL sizes\n"); return 1; }
    uint8_t vk[16];
    for (int i = 0; i < 16; i++) vk[i] = (uint8_t)(3 * i + 1);
    // one batch per task, of its own size (20 .. 51 reports), roles alternating, one report dropped by the keep mask
    std::vector<Batch> B(TASKS);
    uint64_t seed = 7;
    for (uint32_t t = 0; t < TASKS; t++) {
        const uint32_t n = 20 + t;
        std::vector<uint64_t> meas((size_t)n * LEN);
        for (auto &m : meas) m = sm(seed) & 0xffff;
        std::vector<uint8_t> nonces(16 * n), rands(144 * n), lr((size_t)z.leader_report_len * n), hr((size_t)z.helper_report_len * n);
        for (auto &b : nonces) b = (uint8_t)sm(seed);
        for (auto &b : rands) b = (uint8_t)sm(seed);
        if (wh_client_shard_many(&p, meas.data(), LEN, nonces.data(), rands.data(), 1000 * t, n, 4, lr.data(), hr.data())) { printf("FAIL shard\n"); return 1; }
        Batch &b = B[t];
        b.off.push_back(0);
        for (uint32_t i = 0; i < n; i++) {
            const uint8_t *src = (i & 1) ? hr.data() + (size_t)i * z.helper_report_len : lr.data() + (size_t)i * z.leader_report_len;
            const size_t len = (i & 1) ? z.helper_report_len : z.leader_report_len;
            b.blob.insert(b.blob.end(), src, src + len);
            b.off.push_back(b.blob.size());
        }
        b.blob.push_back(0);
        b.keep.assign(n, 1); b.keep[t % n] = 0;
        b.tags.resize(16 * n); b.ids.resize(16 * n); b.status.resize(n); b.agg.resize(16 * z.output_len);
    }
    // reference results: every batch alone on a context of its own
    wh_ctx *solo = nullptr;
    if (wh_ctx_create(0, &p, &solo)) { printf("FAIL ctx (no GPU?)\n"); return 1; }
    for (auto &b : B) {
        const uint64_t n = b.keep.size();
        if (wh_prio3_prepare_batch(solo, vk, b.blob.data(), b.off.data(), n, b.tags.data(), b.ids.data(), b.status.data()) ||
            wh_prio3_aggregate(solo, b.keep.data(), b.agg.data())) { printf("FAIL solo: %s\n", wh_last_error(solo)); return 1; }
        for (auto s : b.status) if (s) { printf("FAIL solo status\n"); return 1; }
    }
    wh_ctx_destroy(solo);

    wh_ctx *ctx[2] = {nullptr, nullptr};
    if (wh_ctx_create(0, &p, &ctx[0]) || wh_ctx_create(0, &p, &ctx[1])) { printf("FAIL ctx\n"); return 1; }
    // context 0: a lone GPU (no RCCL); context 1: the same collectives through RCCL with one rank
    uint8_t id[WH_COMM_ID_BYTES];
    int32_t rc = wh_comm_unique_id(id);
    if (rc) { printf("FAIL wh_comm_unique_id %d\n", rc); return 1; }
    if (wh_comm_init(ctx[0], nullptr, 0, 1)) { printf("FAIL comm_init local: %s\n", wh_last_error(ctx[0])); return 1; }
    if ((rc = wh_comm_init(ctx[1], id, 0, 1))) { printf("FAIL comm_init rccl %d: %s\n", rc, wh_last_error(ctx[1])); return 1; }
    if (wh_comm_init(ctx[1], id, 0, 1) != WH_ERR_STATE) { printf("FAIL second comm_init accepted\n"); return 1; }

    std::atomic<int> failures{0};
    std::string first_error;
    std::mutex err_mu;
    auto fail = [&](const std::string &what) { failures++; std::lock_guard<std::mutex> g(err_mu); if (first_error.empty()) first_error = what; };
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < TASKS; t++)
        th.emplace_back([&, t] {
            const Batch &b = B[t];
            const uint64_t n = b.keep.size();
            std::vector<uint8_t> tags(16 * n), ids(16 * n), agg(16 * z.output_len);
            std::vector<int32_t> status(n);
            for (uint32_t r = 0; r < ROUNDS; r++) {
                wh_ctx *c = ctx[(t + r) & 1];
                // a task's sequence: nobody else may prepare on this context between its prepare and its aggregate
                wh_ctx_lock(c);
                int32_t e = wh_prio3_prepare_batch(c, vk, b.blob.data(), b.off.data(), n, tags.data(), ids.data(), status.data());
                std::this_thread::yield();
                if (!e) e = wh_prio3_allgather_reduce(c, b.keep.data(), agg.data());
                wh_ctx_unlock(c);
                if (e) { fail(std::string("call failed: ") + wh_last_error(c)); return; }
                if (tags != b.tags || ids != b.ids || status != b.status || agg != b.agg) { fail("task " + std::to_string(t) + " got another task's results"); return; }
                // an unbracketed single call is safe too (it takes the lock itself)
                if (wh_ctx_synchronize(c)) { fail("synchronize"); return; }
            }
        });
    for (auto &x : th) x.join();
    if (failures) { printf("FAIL %d tasks: %s\n", failures.load(), first_error.c_str()); return 1; }

    // heavy-hitter counters: in-place all-reduce (one rank: unchanged) through RCCL, on device memory of the caller
    // (the library exports no allocator: take a few words from a prepare output buffer via the test op instead)
    int32_t rk = -1, wd = -1;
    if (wh_comm_rank(ctx[1], &rk, &wd) || rk != 0 || wd != 1) { printf("FAIL comm_rank\n"); return 1; }
    if (wh_comm_destroy(ctx[0]) || wh_comm_destroy(ctx[1])) { printf("FAIL comm_destroy\n"); return 1; }
    std::vector<uint8_t> agg(16 * z.output_len);
    if (wh_prio3_allgather_reduce(ctx[0], nullptr, agg.data()) != WH_ERR_STATE) { printf("FAIL combine without comm accepted\n"); return 1; }
    wh_ctx_destroy(ctx[0]); wh_ctx_destroy(ctx[1]);
    printf("OK: %u threads x %u rounds over two contexts, every task got its own batch's tags, ids and aggregate (lone-GPU and RCCL combine)\n", TASKS, ROUNDS);
    return 0;
}
