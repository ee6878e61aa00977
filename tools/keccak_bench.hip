// Keccak-f[1600] throughput on gfx950 with the state in VGPRs, one sponge per lane, at different
// occupancies.  hipcc --offload-arch=gfx950 -O3 -Iwhisper_amd/csrc tools/keccak_bench.hip -o tools/keccak_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
#include "keccak.hpp"
using namespace wh;
#define PERMS 512
// stamps[wave][4]: shader clock (s_memtime) and 100 MHz real-time counter (s_memrealtime) around the loop; the in-kernel
// clock is their ratio (MI355X_MICROARCH.md, DVFS give-back (6)).  NULL = no stamp executes.
template <int W>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) k(uint64_t *out, uint64_t seed, uint32_t active, unsigned long long *stamps) {
    if ((threadIdx.x & 63) >= active) return;
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = seed * (i + 1) + threadIdx.x + blockIdx.x * 977;
    const uint32_t wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (stamps && (threadIdx.x & 63) == 0) { stamps[4ull * wv] = __builtin_amdgcn_s_memtime(); stamps[4ull * wv + 1] = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll 1
    for (int p = 0; p < PERMS; p++) keccak_f1600(st);
    if (stamps && (threadIdx.x & 63) == 0) { stamps[4ull * wv + 2] = __builtin_amdgcn_s_memtime(); stamps[4ull * wv + 3] = __builtin_amdgcn_s_memrealtime(); }
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < 25; i++) r ^= st[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int W>
void run(uint64_t *out, uint32_t active = 64, int rounds = 4) {
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void *)k<W>);
    int blocks = 256 * W * rounds;   // `rounds` rounds of full occupancy
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, out, 3ull, active, nullptr); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, out, 3ull, active, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("waves/SIMD=%d active lanes=%u rounds=%d vgprs=%d scratch=%zu: %.3f ms  %.2f G perm/s", W, active, rounds, fa.numRegs,
           (size_t)fa.localSizeBytes, ms, (double)blocks * 4 * active * PERMS / ms / 1e6);
    // a stamped launch of the same grid: in-kernel clock and shader cycles per permutation and wavefront
    const uint32_t nw = blocks * 4;
    unsigned long long *d_st; hipMalloc(&d_st, 32ull * nw); hipMemset(d_st, 0, 32ull * nw);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, out, 3ull, active, d_st); hipDeviceSynchronize();
    std::vector<unsigned long long> h(4ull * nw); hipMemcpy(h.data(), d_st, 32ull * nw, hipMemcpyDeviceToHost); hipFree(d_st);
    std::vector<double> clk, cyc;
    for (uint32_t w = 0; w < nw; w++) if (h[4ull * w + 3] > h[4ull * w + 1]) {
        clk.push_back((double)(h[4ull * w + 2] - h[4ull * w]) / (double)(h[4ull * w + 3] - h[4ull * w + 1]) * 0.1);
        cyc.push_back((double)(h[4ull * w + 2] - h[4ull * w]) / PERMS);
    }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    if (!clk.empty()) printf("  | in-kernel clock %.3f GHz (median of %zu waves), %.0f shader cycles per permutation and wave", clk[clk.size() / 2], clk.size(), cyc[cyc.size() / 2]);
    printf("\n");
}
int main() {
    uint64_t *out; hipMalloc(&out, 64 << 20);
    // two seconds of back-to-back launches first: the clock the chip holds under this load, not the boost of an idle chip
    { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms = 0, tot = 0;
      while (tot < 2000) { hipEventRecord(a); hipLaunchKernelGGL(k<3>, dim3(256 * 3 * 4), dim3(256), 0, 0, out, 3ull, 64u, nullptr); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); tot += ms; } }
    run<1>(out, 64, 1); run<2>(out, 64, 1); run<2>(out, 32, 1); run<4>(out, 32, 1); run<4>(out, 16, 1); run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<5>(out); run<6>(out); run<8>(out);
    return 0;
}
