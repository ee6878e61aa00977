// Measures the issue rate of the integer VALU instructions the verify kernels are made of
// (gfx950).  Every lane runs ITERS x 8 independent chains; result in lane-ops per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o tools/valu_microbench && ./tools/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 4096
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 1) + threadIdx.x;
    uint32_t b = seed ^ 0x9e3779b9u, c = seed + blockIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 6) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 7) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 11) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 12) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 13) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 14) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xd2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 15) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// 64-bit mad: d(64) = a(32)*b(32) + c(64)
template <int OP>
__global__ void __launch_bounds__(256) k64(uint64_t *out, uint32_t seed) {
    uint64_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 1) + threadIdx.x;
    uint32_t b = seed ^ 0x9e3779b9u, c = seed + blockIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 8) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
            if (OP == 9) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[i]));
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename F>
static void run(const char *name, F launch, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * ITERS * 8;
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("%-18s %8.3f ms  %7.2f T lane-ops/s  = %6.1f lane-ops/clk/CU at %d MHz nominal\n", name, ms, ops / ms / 1e9,
           ops / (ms * 1e-3) / 256 / (clk_khz * 1e3), clk_khz / 1000);
}

int main() {
    void *out; hipMalloc(&out, 8ull << 20 << 3);
    for (int wpc : {4, 8, 16, 32}) {   // waves per CU
        int blocks = 256 * wpc / 4 * 4;   // several rounds of blocks
        printf("--- %d waves/CU resident (blocks=%d of 256 threads)\n", wpc, blocks);
#define R(OP, NAME) run(NAME, [&] { hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, (uint32_t *)out, 12345u); }, blocks)
#define R64(OP, NAME) run(NAME, [&] { hipLaunchKernelGGL(k64<OP>, dim3(blocks), dim3(256), 0, 0, (uint64_t *)out, 12345u); }, blocks)
        R(10, "v_fma_f32"); R(0, "v_xor_b32"); R(1, "v_alignbit_b32"); R(2, "v_bfi_b32"); R(5, "v_add_u32"); R(6, "v_add_co_u32");
        R(7, "v_addc_co_u32"); R(3, "v_mul_lo_u32"); R(4, "v_mul_hi_u32"); R64(8, "v_mad_u64_u32"); R64(9, "v_lshlrev_b64");
        R(11, "v_and_or_b32"); R(12, "v_lshl_or_b32"); R(13, "v_xad_u32"); R(14, "v_bitop3_b32"); R(15, "v_perm_b32");
    }
    return 0;
}
