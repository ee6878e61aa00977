// The servers' sort of a batch by testing id (server-batch-prio3/src/main.rs:93, 158; server-hh/src/main.rs:175, 266:
// `sort_by(|a, b| a.testing_id.cmp(&b.testing_id))`, a stable sort on a u128) as a least-significant-digit radix sort over the
// sixteen bytes of the id.  Only a permutation moves: every pass reads the digit through the current order (the ids of a
// batch are L2-resident), counts per tile, scans the counts bin-major and scatters in tile order, so equal ids keep their
// input order exactly as the reference's stable sort leaves them.
#include <hip/hip_runtime.h>
#include "sort_dev.hpp"

namespace wh {
namespace {

__device__ __forceinline__ uint32_t digit_of(const uint8_t *ids, uint32_t idx, uint32_t pass) { return ids[16ull * idx + pass]; }

// one wavefront per tile: hist[bin * nblk + blk]
__global__ __launch_bounds__(64) void k_sort_hist(const uint8_t *ids, const uint32_t *in, uint32_t n, uint32_t pass, uint32_t nblk,
                                                  uint32_t *hist) {
    __shared__ uint32_t cnt[256];
    const uint32_t lane = threadIdx.x, blk = blockIdx.x;
    for (uint32_t d = lane; d < 256; d += 64) cnt[d] = 0;
    __syncthreads();
    for (uint32_t it = 0; it < SORT_TILE / 64; it++) {
        const uint32_t i = blk * SORT_TILE + it * 64 + lane;
        if (i < n) atomicAdd(&cnt[digit_of(ids, in ? in[i] : i, pass)], 1u);
    }
    __syncthreads();
    for (uint32_t d = lane; d < 256; d += 64) hist[(uint64_t)d * nblk + blk] = cnt[d];
}

// exclusive scan of the whole table by one workgroup
__global__ __launch_bounds__(1024) void k_sort_scan(uint32_t *hist, uint32_t total) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, chunk = (total + 1023) / 1024;
    const uint32_t lo = t * chunk < total ? t * chunk : total, hi = lo + chunk < total ? lo + chunk : total;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += hist[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {      // Hillis-Steele inclusive scan of the partial sums
        const uint32_t v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - s;
    for (uint32_t i = lo; i < hi; i++) { const uint32_t v = hist[i]; hist[i] = run; run += v; }
}

__global__ __launch_bounds__(64) void k_sort_scatter(const uint8_t *ids, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t pass,
                                                     uint32_t nblk, const uint32_t *hist) {
    __shared__ uint32_t base[256];
    const uint32_t lane = threadIdx.x, blk = blockIdx.x;
    for (uint32_t d = lane; d < 256; d += 64) base[d] = hist[(uint64_t)d * nblk + blk];
    __syncthreads();
    const uint64_t lt = (1ull << lane) - 1;
    for (uint32_t it = 0; it < SORT_TILE / 64; it++) {
        const uint32_t i = blk * SORT_TILE + it * 64 + lane;
        const bool valid = i < n;
        const uint32_t idx = valid ? (in ? in[i] : i) : 0;
        const uint32_t dg = valid ? digit_of(ids, idx, pass) : 0;
        uint64_t peers = __ballot(valid);                  // lanes of this step with the same digit
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            const uint64_t bal = __ballot((dg >> b) & 1);
            peers &= ((dg >> b) & 1) ? bal : ~bal;
        }
        const uint32_t rank = __popcll(peers & lt), count = __popcll(peers);
        uint32_t pos = 0;
        if (valid) pos = base[dg] + rank;
        __syncthreads();
        if (valid) {
            out[pos] = idx;
            if (rank == count - 1) base[dg] += count;      // one lane per digit moves the tile's cursor on
        }
        __syncthreads();
    }
}

// out[k] = rows[order[k]], rows of row_bytes (a multiple of 16)
__global__ __launch_bounds__(256) void k_sort_gather(const uint4 *rows, uint32_t row_u4, const uint32_t *order, uint32_t n, uint4 *out) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (uint64_t)n * row_u4) return;
    const uint64_t k = g / row_u4;
    const uint32_t q = (uint32_t)(g - k * row_u4);
    out[g] = rows[(uint64_t)order[k] * row_u4 + q];
}

// keep[i] = 0 iff ids[i] is in the sorted list `bad` (pairs lo, hi; ascending as u128): binary search per report
__global__ __launch_bounds__(256) void k_keep_mask(const uint8_t *ids, uint32_t n, const uint64_t *bad, uint32_t n_bad, uint8_t *keep) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 v = *reinterpret_cast<const uint4 *>(ids + 16ull * i);
    const uint64_t lo = v.x | ((uint64_t)v.y << 32), hi = v.z | ((uint64_t)v.w << 32);
    uint32_t a = 0, b = n_bad;                 // first entry >= id
    while (a < b) {
        const uint32_t m = (a + b) >> 1;
        const uint64_t ml = bad[2 * m], mh = bad[2 * m + 1];
        if (mh < hi || (mh == hi && ml < lo)) a = m + 1; else b = m;
    }
    keep[i] = !(a < n_bad && bad[2 * a] == lo && bad[2 * a + 1] == hi);
}

}  // namespace

int launch_keep_mask(const uint8_t *d_ids, uint32_t n, const uint64_t *d_bad, uint32_t n_bad, uint8_t *d_keep, hipStream_t st) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_keep_mask, dim3((n + 255) / 256), dim3(256), 0, st, d_ids, n, d_bad, n_bad, d_keep);
    return (int)hipGetLastError();
}

int launch_sort_by_id(const uint8_t *d_ids, uint32_t n, uint32_t *d_order, uint32_t *d_tmp, uint32_t *d_hist, hipStream_t st) {
    if (!n) return 0;
    const uint32_t nblk = (uint32_t)sort_blocks(n);
    // sixteen passes: the permutation ends in d_order when the first pass writes d_tmp
    const uint32_t *in = nullptr;
    uint32_t *out = d_tmp;
    for (uint32_t pass = 0; pass < 16; pass++) {
        hipLaunchKernelGGL(k_sort_hist, dim3(nblk), dim3(64), 0, st, d_ids, in, n, pass, nblk, d_hist);
        hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, st, d_hist, 256 * nblk);
        hipLaunchKernelGGL(k_sort_scatter, dim3(nblk), dim3(64), 0, st, d_ids, in, out, n, pass, nblk, d_hist);
        in = out;
        out = out == d_tmp ? d_order : d_tmp;
    }
    return (int)hipGetLastError();
}

int launch_sort_gather(const uint8_t *d_rows, uint32_t row_bytes, const uint32_t *d_order, uint32_t n, uint8_t *d_out, hipStream_t st) {
    if (!n) return 0;
    const uint32_t row_u4 = row_bytes / 16;
    const uint64_t total = (uint64_t)n * row_u4;
    hipLaunchKernelGGL(k_sort_gather, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint4 *>(d_rows),
                       row_u4, d_order, n, reinterpret_cast<uint4 *>(d_out));
    return (int)hipGetLastError();
}

}  // namespace wh
