// What bounds a table-driven AES round on gfx950: issue rates of the address-forming instructions, and the rate of
// ds_read_b32 on a 256-entry table replicated 32 times (entry*32 + lane%32: no bank conflicts) with dependent indices.
//   hipcc --offload-arch=gfx950 -O3 tools/aes_microbench.hip -o tools/aes_microbench && ./tools/aes_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2048
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 1) + threadIdx.x;
    uint32_t b = seed ^ 0x9e3779b9u, c = seed + blockIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_lshl_add_u32 %0, %0, 7, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "v"(c));
            if (OP == 3) asm volatile("v_lshrrev_b32 %0, 24, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 5) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 6) asm volatile("v_and_b32 %0, 0x7f80, %0" : "+v"(a[i]));
            if (OP == 7) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
            if (OP == 8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// LDS: CH independent lookup chains per lane, each x <- T[(x & 0xff) * 32 + lane % 32]
template <int CH, int EXTRA = 0>
__global__ void __launch_bounds__(256) klds(uint32_t *out, uint32_t seed) {
    __shared__ uint32_t T[256 * 32];
    for (uint32_t k = threadIdx.x; k < 256 * 32; k += 256) T[k] = (k >> 5) * 0x01010101u ^ (k * 2654435761u >> 7);
    __syncthreads();
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint32_t *)T + 4 * (threadIdx.x & 31);
    uint32_t x[CH], y[8];
#pragma unroll
    for (int i = 0; i < CH; i++) x[i] = seed * (i + 3) + threadIdx.x * 7;
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] = seed + i;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            uint32_t sh;
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(sh) : "v"(7u), "v"(x[i]));
            x[i] = *(lds_cu32 *)(uintptr_t)(base + sh);
#pragma unroll
            for (int e = 0; e < EXTRA; e++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(y[(i + e) & 7]) : "v"(seed));
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < CH; i++) r ^= x[i];
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename F>
static void run(const char *name, F launch, double ops) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("%-28s %8.3f ms  = %6.1f lane-ops/clk/CU at %d MHz nominal\n", name, ms, ops / (ms * 1e-3) / 256 / (clk_khz * 1e3), clk_khz / 1000);
}

int main() {
    void *out; hipMalloc(&out, 64ull << 20);
    for (int wpc : {4, 8, 16}) {
        int blocks = 256 * wpc / 4 * 4;
        printf("--- %d waves/CU resident (blocks=%d of 256 threads)\n", wpc, blocks);
        const double ops = (double)blocks * 256 * ITERS * 8;
#define R(OP, NAME) run(NAME, [&] { hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, (uint32_t *)out, 12345u); }, ops)
        R(0, "v_bfe_u32"); R(1, "v_lshl_add_u32"); R(2, "v_lshlrev_b32_sdwa BYTE_1"); R(3, "v_lshrrev_b32"); R(4, "v_bitop3_b32 xor3");
        R(5, "v_or_b32"); R(6, "v_and_b32 literal"); R(7, "v_add_u32_sdwa BYTE_2"); R(8, "v_perm_b32"); R(9, "v_mad_u32_u24");
        run("ds_read_b32 table, 8 chains", [&] { hipLaunchKernelGGL(klds<8>, dim3(blocks), dim3(256), 0, 0, (uint32_t *)out, 12345u); }, ops);
        run("ds_read_b32 table, 16 chains", [&] { hipLaunchKernelGGL(klds<16>, dim3(blocks), dim3(256), 0, 0, (uint32_t *)out, 12345u); }, ops * 2);
        run("  + 1 full-rate VALU per read", [&] { hipLaunchKernelGGL((klds<16, 1>), dim3(blocks), dim3(256), 0, 0, (uint32_t *)out, 12345u); }, ops * 2);
        run("  + 2 full-rate VALU per read", [&] { hipLaunchKernelGGL((klds<16, 2>), dim3(blocks), dim3(256), 0, 0, (uint32_t *)out, 12345u); }, ops * 2);
        run("  + 4 full-rate VALU per read", [&] { hipLaunchKernelGGL((klds<16, 4>), dim3(blocks), dim3(256), 0, 0, (uint32_t *)out, 12345u); }, ops * 2);
    }
    return 0;
}
