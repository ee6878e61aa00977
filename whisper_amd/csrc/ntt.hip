// discrete_fourier_transform and its inverse (batched-flpcp-rs/src/fft.rs:31-85, 102-115) for a batch of independent
// Field128 vectors: out[i] = sum_j in[j] w^(ij), w = the primitive size-th root of unity the reference uses (fp.rs:391-421),
// `in` zero-padded from n_in to size; the inverse is the same transform followed by discrete_fourier_transform_inv_finish
// (scale by 1/size, index i <-> size - i).  This is the transform under poly_fft / poly_interpret_eval
// (polynomial.rs:187-231) that flp::Type::prove and ::query call per wire; the product path does not need it as a separate
// step (the prover fuses its transforms, prove.hip; the verifier evaluates in barycentric form, kernels.hip), so this entry
// point is the stand-alone surface for parity checks against the oracle's restatement of fft.rs.
//
// One workgroup per vector.  The vector and the size/2 twiddles w^k live in LDS (24 B per point, 96 KiB at the 4096-point
// maximum); the input is read in bit-reversed order as the reference does (fft.rs:53-60), then log2(size) radix-2
// decimation-in-time stages run in place with one barrier each.  Level l of the reference multiplies by
// r_l^i, r_l = root(l): r_l^i = w^(i size / 2^l), one table for all levels.  Canonical residues in and out (f128.hpp).
#include <hip/hip_runtime.h>
#include "ntt_dev.hpp"

namespace wh {
namespace {

template <bool INV>
__global__ __launch_bounds__(256) void k_ntt_f128(const F128 *__restrict__ in, uint32_t n_in, uint32_t lg, F128 w, F128 size_inv,
                                                  F128 *__restrict__ out) {
    extern __shared__ F128 ntt_sm[];
    const uint32_t size = 1u << lg, tid = threadIdx.x;
    F128 *a = ntt_sm, *tw = ntt_sm + size;
    const F128 *src = in + (uint64_t)blockIdx.x * n_in;
    F128 *dst = out + (uint64_t)blockIdx.x * size;
    for (uint32_t i = tid; i < size; i += 256) {
        const uint32_t j = lg ? __builtin_bitreverse32(i) >> (32 - lg) : 0;
        a[i] = j < n_in ? src[j] : f_zero();
    }
    for (uint32_t k = tid; k < size / 2; k += 256) tw[k] = f_pow(w, k);
    __syncthreads();
    for (uint32_t l = 1; l <= lg; l++) {
        const uint32_t half = 1u << (l - 1), tstep = size >> l;
        for (uint32_t q = tid; q < size / 2; q += 256) {
            const uint32_t i = q & (half - 1), x = ((q - i) << 1) + i;
            const F128 u = a[x];
            const F128 v = i ? f_mul(tw[i * tstep], a[x + half]) : a[x + half];
            a[x] = f_add(u, v);
            a[x + half] = f_sub(u, v);
        }
        __syncthreads();
    }
    for (uint32_t i = tid; i < size; i += 256) dst[i] = INV ? f_mul(a[(size - i) & (size - 1)], size_inv) : a[i];
}

}  // namespace

int launch_ntt_f128(const F128 *d_in, uint32_t n_in, uint32_t lg, bool inverse, const F128 &w, const F128 &size_inv, uint32_t n,
                    F128 *d_out, hipStream_t stream) {
    if (!n) return 0;
    if (lg > NTT_MAX_LOG2 || n_in > (1u << lg)) return (int)hipErrorInvalidValue;
    const size_t lds = ((size_t)1 << lg) * 24;
    auto fn = inverse ? k_ntt_f128<true> : k_ntt_f128<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fn, dim3(n), dim3(256), lds, stream, d_in, n_in, lg, w, size_inv, d_out);
    return (int)hipGetLastError();
}

}  // namespace wh
