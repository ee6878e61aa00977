// HBM read rate for the query kernel's access pattern: one wavefront per report (or S waves per
// report), each streaming its report 1 KiB per load instruction with D loads in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_pattern.hip -o tools/stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));

template <int D>
__global__ void __launch_bounds__(256) k(const uint8_t *blob, uint64_t rep_stride, uint32_t rep_bytes, uint32_t misalign,
                                         uint32_t S, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint64_t gw = (uint64_t)blockIdx.x * waves_per_block + wave;      // global wave
    const uint64_t rep = gw / S;
    const uint32_t sub = gw % S;
    const uint8_t *base = blob + rep * rep_stride + misalign + 1024ull * sub + 16ull * lane;
    const uint32_t steps = rep_bytes / (1024 * S);
    u32x4 acc = {0, 0, 0, 0};
    u32x4 buf[D];
#pragma unroll
    for (int u = 0; u < D; u++) buf[u] = *(const __attribute__((address_space(1))) u32x4_u *)(base + 1024ull * S * u);
    for (uint32_t s = 0; s < steps; s += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            u32x4 v = buf[u];
            uint32_t nx = s + u + D < steps ? s + u + D : 0;
            buf[u] = *(const __attribute__((address_space(1))) u32x4_u *)(base + 1024ull * S * nx);
            acc ^= v;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[gw] = 1;
}

int main(int argc, char **argv) {
    const uint32_t rep_bytes = 262144;
    const uint64_t rep_stride = 295081;          // Leader report size of the bench workload
    const uint32_t n = 62500;
    uint8_t *blob; uint32_t *out;
    hipMalloc(&blob, rep_stride * n + 4096); hipMalloc(&out, 4u << 20);
    hipMemset(blob, 1, rep_stride * n + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, uint32_t S, uint32_t wpb, uint64_t stride, uint32_t mis) {
        uint32_t blocks = n * S / wpb;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * wpb), 0, 0, blob, stride, rep_bytes, mis, S, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s S=%u waves/block=%u stride=%llu misalign=%u: %.2f ms  %.2f TB/s\n", name, S, wpb, (unsigned long long)stride, mis, ms,
               (double)n * rep_bytes / ms / 1e9);
    };
    for (uint32_t mis : {49u, 0u}) {
        uint64_t stride = mis ? rep_stride : 294912;
        run("D=2", k<2>, 1, 4, stride, mis); run("D=4", k<4>, 1, 4, stride, mis); run("D=8", k<8>, 1, 4, stride, mis); run("D=16", k<16>, 1, 4, stride, mis);
        run("D=4", k<4>, 2, 2, stride, mis); run("D=8", k<8>, 2, 2, stride, mis);
        run("D=4", k<4>, 4, 4, stride, mis); run("D=8", k<8>, 4, 4, stride, mis);
        run("D=4", k<4>, 1, 1, stride, mis); run("D=8", k<8>, 1, 1, stride, mis);
    }
    return 0;
}
