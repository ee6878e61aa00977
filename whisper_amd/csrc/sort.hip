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

// one workgroup per bin: exclusive scan of the bin's row of tile counts in place (coalesced, 1 024 tiles at a time) and the
// bin's total; the scatter turns the 256 totals into bin bases itself
__global__ __launch_bounds__(1024) void k_sort_scan(uint32_t *hist, uint32_t nblk, uint32_t *totals) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry_s;
    const uint32_t t = threadIdx.x;
    uint32_t *row = hist + (uint64_t)blockIdx.x * nblk;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblk; base += 1024) {
        const uint32_t i = base + t;
        const uint32_t v = i < nblk ? row[i] : 0;
        part[t] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {      // Hillis-Steele inclusive scan
            const uint32_t u = t >= off ? part[t - off] : 0;
            __syncthreads();
            part[t] += u;
            __syncthreads();
        }
        const uint32_t carry = carry_s;
        if (i < nblk) row[i] = carry + part[t] - v;
        __syncthreads();
        if (t == 1023) carry_s = carry + part[1023];
        __syncthreads();
    }
    if (t == 0) totals[blockIdx.x] = carry_s;
}

__global__ __launch_bounds__(64) void k_sort_scatter(const uint8_t *ids, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t pass,
                                                     uint32_t nblk, const uint32_t *hist, const uint32_t *totals) {
    __shared__ uint32_t base[256];
    const uint32_t lane = threadIdx.x, blk = blockIdx.x;
    {   // bin bases = exclusive scan of the 256 totals: four per lane, then across the wavefront
        const uint32_t t0 = totals[4 * lane], t1 = totals[4 * lane + 1], t2 = totals[4 * lane + 2], t3 = totals[4 * lane + 3];
        uint32_t incl = t0 + t1 + t2 + t3;
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off, 64);
            if (lane >= off) incl += u;
        }
        const uint32_t ex = incl - (t0 + t1 + t2 + t3);
        base[4 * lane] = ex + hist[(uint64_t)(4 * lane) * nblk + blk];
        base[4 * lane + 1] = ex + t0 + hist[(uint64_t)(4 * lane + 1) * nblk + blk];
        base[4 * lane + 2] = ex + t0 + t1 + hist[(uint64_t)(4 * lane + 2) * nblk + blk];
        base[4 * lane + 3] = ex + t0 + t1 + t2 + hist[(uint64_t)(4 * lane + 3) * nblk + blk];
    }
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
        hipLaunchKernelGGL(k_sort_scan, dim3(256), dim3(1024), 0, st, d_hist, nblk, d_hist + 256ull * nblk);
        hipLaunchKernelGGL(k_sort_scatter, dim3(nblk), dim3(64), 0, st, d_ids, in, out, n, pass, nblk, d_hist, d_hist + 256ull * nblk);
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
