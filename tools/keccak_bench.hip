// Keccak-f[1600] throughput on gfx950 with the state in VGPRs, one sponge per lane, at different
// occupancies.  hipcc --offload-arch=gfx950 -O3 -Iwhisper_amd/csrc tools/keccak_bench.hip -o tools/keccak_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "keccak.hpp"
using namespace wh;
#define PERMS 512
template <int W>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) k(uint64_t *out, uint64_t seed, uint32_t active) {
    if ((threadIdx.x & 63) >= active) return;
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = seed * (i + 1) + threadIdx.x + blockIdx.x * 977;
#pragma unroll 1
    for (int p = 0; p < PERMS; p++) keccak_f1600(st);
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
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, out, 3ull, active); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, out, 3ull, active); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("waves/SIMD=%d active lanes=%u rounds=%d vgprs=%d scratch=%zu: %.3f ms  %.2f G perm/s\n", W, active, rounds, fa.numRegs,
           (size_t)fa.localSizeBytes, ms, (double)blocks * 4 * active * PERMS / ms / 1e6);
}
int main() {
    uint64_t *out; hipMalloc(&out, 64 << 20);
    run<1>(out, 64, 1); run<2>(out, 64, 1); run<2>(out, 32, 1); run<4>(out, 32, 1); run<4>(out, 16, 1); run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<5>(out); run<6>(out); run<8>(out);
    return 0;
}
