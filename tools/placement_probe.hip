// Where do the workgroups of a small grid land?  Each block records (XCC, SE, CU) and spins for a
// while so that all blocks of the grid are resident together.  Blocks-per-CU limit set through LDS.
//   hipcc --offload-arch=gfx950 -O3 tools/placement_probe.hip -o tools/placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) probe(uint32_t *out, uint64_t spin) {
    extern __shared__ uint32_t lds[];
    uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
    uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
    uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) { lds[threadIdx.x] = hw; }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
    uint32_t *d; hipMalloc(&d, 1 << 20);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int per_cu : {1, 2, 3, 8}) for (int grid : {244, 489, 978}) {
        size_t lds = per_cu == 8 ? 1024 : (160 * 1024 / per_cu) & ~1023;
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, d, 20000000ull);   // ~0.2 s at 100 MHz counter
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed lds=%zu\n", lds); continue; }
        std::vector<uint32_t> h(2 * grid); hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
        std::map<uint32_t, int> per; 
        for (int b = 0; b < grid; b++) { uint32_t hw = h[2 * b], x = h[2 * b + 1] & 0xf; uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; per[(x << 12) | (se << 8) | (sh << 4) | cu]++; }
        int hist[16] = {0}; for (auto &kv : per) hist[kv.second < 15 ? kv.second : 15]++;
        printf("max %d blocks/CU (lds %zu), grid %d: distinct CUs %zu; CUs holding k blocks:", per_cu, lds, grid, per.size());
        for (int k = 1; k < 16; k++) if (hist[k]) printf(" %dx%d", hist[k], k);
        printf("\n");
    }
    return 0;
}
