// Dependent-issue latency of v_mad_u64_u32 on gfx950: NCH independent accumulator chains per lane,
// one wave per SIMD unless stated, cycles per instruction from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/mad_latency.hip -o tools/mad_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
template <int NCH, int KIND>
__global__ void __launch_bounds__(256) k(uint64_t *out, uint32_t seed, uint64_t *cyc) {
    uint64_t a[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) a[i] = seed * (i + 1) + threadIdx.x;
    uint32_t b = seed ^ 0x9e3779b9u, c = seed + blockIdx.x;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            if (KIND == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
            if (KIND == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % NCH]));
            if (KIND == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(*(uint32_t *)&a[i]) : "v"(b));
            if (KIND == 3) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(*(uint32_t *)&a[i]) : "v"(b));
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < NCH; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NCH, int KIND>
void run(const char *name, int threads, uint64_t *out, uint64_t *cyc) {
    hipLaunchKernelGGL((k<NCH, KIND>), dim3(1), dim3(threads), 0, 0, out, 12345u, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NCH, KIND>), dim3(1), dim3(threads), 0, 0, out, 12345u, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-16s chains=%d waves/SIMD=%d  %.1f ns/instr/wave  (counter ticks/instr %.2f)\n", name, NCH, threads / 256,
           ms * 1e6 / ((double)ITERS * NCH), (double)h / ((double)ITERS * NCH));
}
int main() {
    uint64_t *out, *cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    for (int threads : {256, 512, 1024}) {
        run<1, 0>("v_mad_u64_u32", threads, out, cyc); run<2, 0>("v_mad_u64_u32", threads, out, cyc);
        run<4, 0>("v_mad_u64_u32", threads, out, cyc); run<8, 0>("v_mad_u64_u32", threads, out, cyc);
        run<1, 1>("v_lshl_add_u64", threads, out, cyc); run<4, 1>("v_lshl_add_u64", threads, out, cyc); run<8, 1>("v_lshl_add_u64", threads, out, cyc);
        run<1, 2>("v_xor_b32", threads, out, cyc); run<4, 2>("v_xor_b32", threads, out, cyc); run<8, 2>("v_xor_b32", threads, out, cyc);
        run<1, 3>("v_alignbit_b32", threads, out, cyc); run<4, 3>("v_alignbit_b32", threads, out, cyc);
    }
    return 0;
}
