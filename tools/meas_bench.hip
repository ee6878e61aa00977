// k_meas (whisper_amd/csrc/meas.hip) alone on a synthetic workspace: Leader-only, Helper-only and mixed launches, Keccak-f/s
// and the in-kernel clock (s_memtime / s_memrealtime stamps of a diagnostic build, -DWH_MEAS_CLOCKPROBE).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iwhisper_amd/csrc [-D...] tools/meas_bench.hip -o tools/meas_bench
//   tools/meas_bench [leaders helpers [input_len [reps]]]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifdef WH_MEAS_CLOCKPROBE
__device__ unsigned long long *g_probe;      // [wave][6]: memtime, memrealtime at entry and exit, HW_ID, XCC_ID
#endif
#include "meas.hip"
using namespace wh;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const uint32_t nl = argc > 1 ? atoi(argv[1]) : 62500, nh = argc > 2 ? atoi(argv[2]) : 62500;
    const uint32_t IL = argc > 3 ? atoi(argv[3]) : 16384, reps = argc > 4 ? atoi(argv[4]) : 3;
    const uint32_t n = nl + nh, npad = (n + 255) / 256 * 256, VL = 258;
    Prio3Dev P{};
    P.input_len = IL; P.verifier_len = VL; P.alg_id = 2; P.mode = 0; P.sampler_bound = F128{P_LO, P_HI};
    Workspace W{};
    const uint64_t row = 16ull * IL;
    uint8_t *d_data, *d_small;
    CK(hipMalloc(&d_data, (uint64_t)n * row + 256));
    const uint32_t tail_len = 40 + 32 * VL + 32, small_per = 128;
    CK(hipMalloc(&d_small, (uint64_t)n * small_per + tail_len + 64));
    CK(hipMemset(d_small, 0x5a, (uint64_t)n * small_per + tail_len + 64));
    // Leader shares: pseudo-random bytes (the clock the chip holds depends on the data)
    {
        std::vector<uint64_t> h(1 << 20);
        uint64_t x = 88172645463325252ull;
        for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x; }
        for (uint64_t off = 0; off < (uint64_t)nl * row; off += h.size() * 8) {
            const uint64_t len = std::min<uint64_t>(h.size() * 8, (uint64_t)nl * row - off);
            CK(hipMemcpy(d_data + off, h.data(), len, hipMemcpyHostToDevice));
        }
    }
    std::vector<RepDesc> desc(n);
    std::vector<uint32_t> llist(nl ? nl : 1), hlist(nh ? nh : 1);
    uint8_t *tail = d_small + (uint64_t)n * small_per;
    for (uint32_t i = 0; i < n; i++) {
        RepDesc &d = desc[i];
        uint8_t *sm = d_small + (uint64_t)i * small_per;
        d.base = sm; d.meas = d_data + (uint64_t)i * row; d.proof1 = d.proof2 = nullptr; d.tail = tail;
        d.jr_blind = sm + 64; d.seeds = sm + 80; d.role = i < nl ? 0 : 1; d.hslot = i < nl ? 0 : i - nl;
        if (i < nl) llist[i] = i; else hlist[i - nl] = i;
    }
    // distinct seeds and blinds per report
    {
        std::vector<uint8_t> sm((uint64_t)n * small_per);
        uint64_t x = 1234567;
        for (auto &b : sm) { x = x * 6364136223846793005ull + 1442695040888963407ull; b = (uint8_t)(x >> 56); }
        CK(hipMemcpy(d_small, sm.data(), sm.size(), hipMemcpyHostToDevice));
    }
    uint32_t counters[16] = {nh, nl, 0};
    int32_t *d_status;
    CK(hipMalloc(&W.desc, n * sizeof(RepDesc))); CK(hipMalloc(&W.scal, n * sizeof(RepScal)));
    CK(hipMalloc(&W.hlist, 4 * hlist.size())); CK(hipMalloc(&W.llist, 4 * llist.size())); CK(hipMalloc(&W.dirty, 4 * (n + 1)));
    CK(hipMalloc(&W.counters, 64)); CK(hipMalloc(&d_status, 4 * n));
    CK(hipMemcpy(W.desc, desc.data(), n * sizeof(RepDesc), hipMemcpyHostToDevice));
    CK(hipMemcpy(W.hlist, hlist.data(), 4 * hlist.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(W.llist, llist.data(), 4 * llist.size(), hipMemcpyHostToDevice));
    CK(hipMemset(d_status, 0, 4 * n));
    const uint32_t nwaves = (3 * npad + 64 + 255) / 256 * 4;
#ifdef WH_MEAS_CLOCKPROBE
    unsigned long long *d_probe;
    CK(hipMalloc(&d_probe, (uint64_t)nwaves * 48));
    CK(hipMemset(d_probe, 0, (uint64_t)nwaves * 48));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &d_probe, sizeof d_probe));
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double perms_per_lane = (25.0 + 17 + 16.0 * IL) / 168 + 1;
    const double lanes = nl + 2.0 * nh;
    for (uint32_t r = 0; r < reps; r++) {
        CK(hipMemcpy(W.counters, counters, 64, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        launch_meas(P, W, 0, npad, d_status, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        uint32_t c[16]; CK(hipMemcpy(c, W.counters, 64, hipMemcpyDeviceToHost));
        printf("leaders %u helpers %u input_len %u: %.3f ms  %.2f G Keccak-f/s  (dirty %u)", nl, nh, IL, ms, lanes * perms_per_lane / ms / 1e6, c[2]);
#ifdef WH_MEAS_CLOCKPROBE
        std::vector<unsigned long long> pr((uint64_t)nwaves * 6);
        CK(hipMemcpy(pr.data(), d_probe, pr.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> clk, cyc;
        for (uint32_t w = 0; w < nwaves; w++) {
            const unsigned long long *q = &pr[6ull * w];
            if (!q[1] || q[3] <= q[1]) continue;
            const double dt = (double)(q[3] - q[1]), dc = (double)(q[2] - q[0]);
            if (dt < 1000) continue;                    // waves that had nothing to do
            clk.push_back(dc / dt * 100e6 / 1e9); cyc.push_back(dc);
        }
        if (!clk.empty()) {
            std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
            printf("  in-kernel clock median %.3f GHz (min %.3f max %.3f, %zu waves), wave cycles median %.0f = %.0f per permutation",
                   clk[clk.size() / 2], clk.front(), clk.back(), clk.size(), cyc[cyc.size() / 2], cyc[cyc.size() / 2] / perms_per_lane);
        }
        {   // residency: wavefronts per SIMD (XCC, SE, SH, CU, SIMD of HW_ID), and when they started and ended
            std::vector<std::pair<unsigned long long, int>> ev;
            std::vector<uint32_t> per_simd(1 << 16, 0);
            unsigned long long t_first = ~0ull, t_last = 0;
            for (uint32_t w = 0; w < nwaves; w++) { const unsigned long long *q = &pr[6ull * w]; if (q[1] && q[3] > q[1] + 1000) { t_first = std::min(t_first, q[1]); t_last = std::max(t_last, q[3]); } }
            uint32_t late = 0, early_end = 0, cnt = 0;
            for (uint32_t w = 0; w < nwaves; w++) {
                const unsigned long long *q = &pr[6ull * w];
                if (!q[1] || q[3] <= q[1] + 1000) continue;
                cnt++;
                const uint32_t hw = (uint32_t)q[4], key = ((uint32_t)(q[5] & 0xf) << 12) | (((hw >> 8) & 0xff) << 4) | (((hw >> 4) & 3) << 2);
                per_simd[key >> 2]++;
                if (q[1] > t_first + 100000) late++;              // started more than 1 ms after the first
                if (q[3] + 100000 < t_last) early_end++;          // ended more than 1 ms before the last
            }
            uint32_t hist[16] = {0}, simds = 0;
            for (uint32_t v : per_simd) if (v) { hist[v < 15 ? v : 15]++; simds++; }
            printf("\n    span %.2f ms, %u waves on %u SIMDs; waves per SIMD:", (t_last - t_first) / 1e5, cnt, simds);
            for (int v = 1; v < 16; v++) if (hist[v]) printf(" %d:%u", v, hist[v]);
            printf("; started > 1 ms late: %u, ended > 1 ms early: %u", late, early_end);
            std::vector<double> ends; uint32_t slots[16] = {0};
            for (uint32_t w = 0; w < nwaves; w++) { const unsigned long long *q = &pr[6ull * w]; if (q[1] && q[3] > q[1] + 1000) { ends.push_back((q[3] - t_first) / 1e5); slots[q[4] & 15]++; } }
            std::sort(ends.begin(), ends.end());
            printf("\n    wave end times (ms after the first start): p5 %.2f p33 %.2f p50 %.2f p66 %.2f p90 %.2f max %.2f; wave slots used:", ends[ends.size() / 20], ends[ends.size() / 3],
                   ends[ends.size() / 2], ends[ends.size() * 2 / 3], ends[ends.size() * 9 / 10], ends.back());
            for (int v = 0; v < 16; v++) if (slots[v]) printf(" %d:%u", v, slots[v]);
        }
        CK(hipMemset(d_probe, 0, (uint64_t)nwaves * 48));
#endif
        printf("\n");
    }
    return 0;
}
