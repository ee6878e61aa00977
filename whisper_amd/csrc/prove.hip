// flp::Type::prove (batched-flpcp-rs/src/flp.rs:231-347) on gfx950 for n independent (input, prove_rand, joint_rand)
// tuples: the wire polynomials of the one gadget (ParallelSum<Mul> for SumVec / Histogram, flp/gadgets.rs:231-293;
// PolyEval x^2 - x for Average, :111-219), their product summed over the gadget's sub-circuits, and the proof
// = wire seeds | gadget polynomial coefficients.  Same field elements as the reference's
// poly_fft / discrete_fourier_transform path (fft.rs:31-124, polynomial.rs), obtained with fewer transforms:
//
//   the reference interpolates each wire over the m call points (inverse FFT), evaluates it on 2m points (FFT of the
//   zero-padded coefficients) and multiplies there.  The even half of those 2m evaluations IS the recorded wire values,
//   and the odd half is the size-m transform of the coefficients twisted by w_2m^j: two size-m transforms per wire
//   instead of one of size m and one of size 2m.  The inverse transform runs decimation-in-frequency (natural in,
//   bit-reversed out), the twist table is stored bit-reversed with 1/m folded in, the forward transform runs
//   decimation-in-time (bit-reversed in, natural out): no permutation pass.
//
// One workgroup (4 wavefronts) per tuple; a wavefront takes whole wire pairs (2q, 2q+1), keeps both wires in LDS and its
// share of the 2m product evaluations in registers; the four partial sums meet in LDS and the workgroup runs the
// final size-2m inverse transform together.  Field arithmetic on canonical residues (f128.hpp); VALU-bound.
#include <hip/hip_runtime.h>
#include "prove_dev.hpp"

namespace wh {
namespace {

constexpr int32_t ERR_MODULUS = -3;  // WH_ERR_MODULUS
constexpr uint32_t MAXP = 8;         // m / 64 <= 8  (m <= 512)

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ F128 ldg(const uint8_t *p) { F128 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ uint32_t bitrev(uint32_t v, uint32_t bits) { return __builtin_bitreverse32(v) >> (32 - bits); }

// one radix-2 stage of a size-n transform on `a` (and `b` when TWO), butterflies first, first + step, ...
// tw = w_2m^k table; tscale = 2m / n; inverse transforms read the table backwards
template <bool DIF, bool INV, bool TWO>
__device__ __forceinline__ void stage(F128 *a, F128 *b, uint32_t n, uint32_t half, uint32_t first, uint32_t step,
                                      const F128 *__restrict__ tw, uint32_t two_m) {
    const uint32_t tscale = two_m / n;
    for (uint32_t q = first; q < n / 2; q += step) {
        const uint32_t k = q & (half - 1), i = ((q - k) << 1) + k, j = i + half;
        const uint32_t e = k * (n / (2 * half)) * tscale;
        const F128 w = tw[INV ? (e ? two_m - e : 0) : e];
        F128 u = a[i], v = a[j];
        if (DIF) { a[i] = f_add(u, v); v = f_sub(u, v); a[j] = e ? f_mul(v, w) : v; }
        else { if (e) v = f_mul(v, w); a[i] = f_add(u, v); a[j] = f_sub(u, v); }
        if (TWO) {
            u = b[i]; v = b[j];
            if (DIF) { b[i] = f_add(u, v); v = f_sub(u, v); b[j] = e ? f_mul(v, w) : v; }
            else { if (e) v = f_mul(v, w); b[i] = f_add(u, v); b[j] = f_sub(u, v); }
        }
    }
}

template <bool PAIRS>
__global__ __launch_bounds__(256) void k_flp_prove(ProveDev D, const uint8_t *__restrict__ input, const uint8_t *__restrict__ prove_rand,
                                                   const uint8_t *__restrict__ joint_rand, uint8_t *__restrict__ proof,
                                                   int32_t *__restrict__ status) {
    extern __shared__ F128 sm[];
    const uint32_t m = D.m, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t it = blockIdx.x;
    F128 *rc = sm;                          // (rho^chunk)^(j-1), j = 1 .. calls
    F128 *accbuf = sm + m;                  // the 2m product evaluations, then the gadget polynomial
    F128 *wa = accbuf + 2 * m + (size_t)wv * 2 * m, *wb = wa + m;
    const uint8_t *x = input + it * 16ull * D.input_len, *pr = prove_rand + it * 16ull * D.arity, *jr = joint_rand + it * 16ull * D.jr_len;
    uint8_t *out = proof + it * 16ull * D.proof_len;

    // decode checks: every element must be a canonical residue (field.rs:405-423)
    int bad = 0;
    for (uint32_t k = tid; k < D.input_len + D.arity + D.jr_len; k += 256) {
        const F128 v = ldg(k < D.input_len ? x + 16ull * k : (k < D.input_len + D.arity ? pr + 16ull * (k - D.input_len) : jr + 16ull * (k - D.input_len - D.arity)));
        bad |= !f_in_range(v.lo, v.hi);
    }
    bad = __syncthreads_or(bad);
    if (bad) {
        for (uint32_t k = tid; k < 4 * D.proof_len; k += 256) reinterpret_cast<uint32_t *>(out)[k] = 0;
        if (tid == 0) status[it] = ERR_MODULUS;
        return;
    }
    if (tid == 0) status[it] = 0;
    for (uint32_t w = tid; w < D.arity; w += 256) *reinterpret_cast<F128 *>(out + 16ull * w) = ldg(pr + 16ull * w);   // wire seeds

    const uint32_t P = m < 64 ? 1 : m / 64;
    F128 acc_e[MAXP], acc_o[MAXP];
#pragma unroll
    for (uint32_t p = 0; p < MAXP; p++) { acc_e[p] = f_zero(); acc_o[p] = f_zero(); }
    const F128 minus_one = f_neg(f_one());

    if (PAIRS) {
        const F128 rho = ldg(jr);
        const F128 rho_c = f_pow(rho, D.chunk);
        for (uint32_t t = tid; t < D.calls; t += 256) rc[t] = f_pow(rho_c, t);
        __syncthreads();
        F128 rq = f_pow(rho, wv + 1);       // rho^(q+1) for this wavefront's wire pair q
        const F128 rho4 = f_sqr(f_sqr(rho));
#pragma unroll 1
        for (uint32_t q = wv; q < D.chunk; q += 4) {
            // recorded wire values: seed, then the gadget inputs of each call (types.rs:742-777, num_shares = 1)
#pragma unroll
            for (uint32_t p = 0; p < MAXP; p++) {
                const uint32_t j = lane + 64 * p;
                if (p < P && j < m) {
                    F128 a = f_zero(), b = f_zero();
                    if (j == 0) { a = ldg(pr + 32ull * q); b = ldg(pr + 32ull * q + 16); }
                    else if (j <= D.calls) {
                        const uint32_t i = (j - 1) * D.chunk + q;
                        if (i < D.input_len) {
                            const F128 xi = ldg(x + 16ull * i);
                            a = f_mul(f_mul(rq, rc[j - 1]), xi);
                            b = f_sub(xi, f_one());
                        } else b = minus_one;
                    }
                    wa[j] = a; wb[j] = b;
                    acc_e[p] = f_add(acc_e[p], f_mul(a, b));
                }
            }
            wave_sync();
            for (uint32_t half = m / 2; half >= 1; half >>= 1) { stage<true, true, true>(wa, wb, m, half, lane, 64, D.omega2m, 2 * m); wave_sync(); }
            for (uint32_t t = lane; t < m; t += 64) { const F128 tw = D.twist[t]; wa[t] = f_mul(wa[t], tw); wb[t] = f_mul(wb[t], tw); }
            wave_sync();
            for (uint32_t half = 1; half < m; half <<= 1) { stage<false, false, true>(wa, wb, m, half, lane, 64, D.omega2m, 2 * m); wave_sync(); }
#pragma unroll
            for (uint32_t p = 0; p < MAXP; p++) {
                const uint32_t t = lane + 64 * p;
                if (p < P && t < m) acc_o[p] = f_add(acc_o[p], f_mul(wa[t], wb[t]));
            }
            wave_sync();
            rq = f_mul(rq, rho4);
        }
    } else if (wv == 0) {
        // Average: one wire holding the measurement bits, gadget x^2 - x (types.rs:254-268)
#pragma unroll
        for (uint32_t p = 0; p < MAXP; p++) {
            const uint32_t j = lane + 64 * p;
            if (p < P && j < m) {
                F128 a = f_zero();
                if (j == 0) a = ldg(pr);
                else if (j <= D.calls) a = ldg(x + 16ull * (j - 1));
                wa[j] = a;
                acc_e[p] = f_sub(f_sqr(a), a);
            }
        }
        wave_sync();
        for (uint32_t half = m / 2; half >= 1; half >>= 1) { stage<true, true, false>(wa, wa, m, half, lane, 64, D.omega2m, 2 * m); wave_sync(); }
        for (uint32_t t = lane; t < m; t += 64) wa[t] = f_mul(wa[t], D.twist[t]);
        wave_sync();
        for (uint32_t half = 1; half < m; half <<= 1) { stage<false, false, false>(wa, wa, m, half, lane, 64, D.omega2m, 2 * m); wave_sync(); }
#pragma unroll
        for (uint32_t p = 0; p < MAXP; p++) {
            const uint32_t t = lane + 64 * p;
            if (p < P && t < m) { const F128 a = wa[t]; acc_o[p] = f_sub(f_sqr(a), a); }
        }
    }

    // the four wavefronts' partial sums: evaluation 2t is the even half, 2t + 1 the odd half
    for (uint32_t w = 0; w < 4; w++) {
        if (wv == w) {
#pragma unroll
            for (uint32_t p = 0; p < MAXP; p++) {
                const uint32_t t = lane + 64 * p;
                if (p < P && t < m) {
                    accbuf[2 * t] = w ? f_add(accbuf[2 * t], acc_e[p]) : acc_e[p];
                    accbuf[2 * t + 1] = w ? f_add(accbuf[2 * t + 1], acc_o[p]) : acc_o[p];
                }
            }
        }
        __syncthreads();
    }
    // gadget polynomial = inverse transform of the 2m evaluations; coefficient bitrev(pos) lands at pos
    for (uint32_t half = m; half >= 1; half >>= 1) { stage<true, true, false>(accbuf, accbuf, 2 * m, half, tid, 256, D.omega2m, 2 * m); __syncthreads(); }
    for (uint32_t pos = tid; pos < 2 * m; pos += 256) {
        const uint32_t idx = bitrev(pos, D.log2m + 1);
        if (idx < D.gpoly_len) *reinterpret_cast<F128 *>(out + 16ull * (D.arity + idx)) = f_mul(accbuf[pos], D.inv2m);
    }
}

}  // namespace

size_t flp_prove_lds_bytes(uint32_t m) { return 11ull * m * sizeof(F128); }

int launch_flp_prove(const ProveDev &D, bool pairs, const uint8_t *d_input, const uint8_t *d_prove_rand, const uint8_t *d_joint_rand,
                     uint32_t n, uint8_t *d_proof, int32_t *d_status, hipStream_t stream) {
    if (!n) return 0;
    const size_t lds = flp_prove_lds_bytes(D.m);
    auto fn = pairs ? k_flp_prove<true> : k_flp_prove<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fn, dim3(n), dim3(256), lds, stream, D, d_input, d_prove_rand, d_joint_rand, d_proof, d_status);
    return (int)hipGetLastError();
}

}  // namespace wh
