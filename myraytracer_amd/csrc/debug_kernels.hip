// Diagnostic kernels behind mrt_debug_arith / mrt_debug_arith_pairs (include/myraytracer_amd_debug.h): the render kernel's
// own forms of division and square root against hipcc's, on the device.  A translation unit of its own so that the render
// path's source (kernels.hip) holds nothing but the render path.
#include <hip/hip_runtime.h>
#include "mrt_internal.h"
#include "mrt_device.h"

namespace mrt {
namespace {

// ---- mrt_debug_arith: the hand-rolled division / square root against hipcc's own, on the device ------------------------
// div_unscaled / sqrt_unscaled replace the compiler's correctly rounded expansions at every root, normal and normalize of
// the render kernel (shader.wgsl:286-299, :354, :381).  This kernel runs both forms side by side over whole operand ranges
// and counts the operands whose results differ in any bit (two NaNs count as equal).
//   mode 0: sqrt_unscaled(x) vs sqrtf(x) for EVERY f32 bit pattern in [r0, r1]
//   mode 1: div_unscaled(n, divisor_of(d)) vs n / d for `count` pairs: |n| a bit pattern drawn uniformly from [r0, r1], |d|
//           from [r2, r3], n of either sign; mode 2: d of either sign too
// out[0] tested, out[1] mismatches, out[2] the smallest mismatching operand (mode 0: x; else bits(n) | bits(d) << 32)
__device__ __forceinline__ bool same_bits_or_both_nan(float a, float b) {
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}
__device__ __forceinline__ uint32_t bits_in_range(uint32_t r, uint32_t lo, uint32_t hi) {
    return lo + (uint32_t)(((unsigned long long)r * ((unsigned long long)(hi - lo) + 1ull)) >> 32);
}
__global__ void __launch_bounds__(256) arith_check_kernel(int mode, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3,
                                                          unsigned long long count, unsigned long long seed,
                                                          unsigned long long* out) {
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0, first = ~0ull, tested = 0;
    if (mode == 0) {
        for (unsigned long long i = (unsigned long long)r0 + tid; i <= (unsigned long long)r1; i += stride) {
            const float x = __uint_as_float((uint32_t)i);
            tested++;
            if (!same_bits_or_both_nan(sqrt_unscaled(x), __builtin_sqrtf(x))) { bad++; first = i < first ? i : first; }
        }
    } else {
        for (unsigned long long i = tid; i < count; i += stride) {
            const unsigned long long z = splitmix64_at(seed, i);
            uint32_t nb = bits_in_range((uint32_t)z, r0, r1), db = bits_in_range((uint32_t)(z >> 32), r2, r3);
            const unsigned long long z2 = splitmix64_at(seed ^ 0x5851F42D4C957F2Dull, i);
            nb |= (uint32_t)(z2 & 1u) << 31;
            if (mode == 2) db |= (uint32_t)(z2 & 2u) << 30;
            const float n = __uint_as_float(nb), d = __uint_as_float(db);
            tested++;
            if (!same_bits_or_both_nan(div_unscaled(n, divisor_of(d)), n / d)) {
                const unsigned long long key = (unsigned long long)nb | ((unsigned long long)db << 32);
                bad++; first = key < first ? key : first;
            }
        }
    }
    atomicAdd(out + 0, tested);
    if (bad) { atomicAdd(out + 1, bad); atomicMin(out + 2, first); }
}
// caller-supplied operands: out[6 i ..] = bits(x / y), bits(div_unscaled(x, y)), bits(sqrtf(x)), bits(sqrt_unscaled(x)),
// normal_unscaled_ok(|x|, y) (x a component of at - centre, y the radius), normalize_unscaled_ok(x, |y|) (x the squared
// length, y a component)
__global__ void __launch_bounds__(256) arith_pairs_kernel(const float* x, const float* y, uint32_t n, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[i], b = y[i];
    out[6u * i + 0u] = __float_as_uint(a / b);
    out[6u * i + 1u] = __float_as_uint(div_unscaled(a, divisor_of(b)));
    out[6u * i + 2u] = __float_as_uint(__builtin_sqrtf(a));
    out[6u * i + 3u] = __float_as_uint(sqrt_unscaled(a));
    out[6u * i + 4u] = normal_unscaled_ok(__builtin_fabsf(a), b) ? 1u : 0u;
    out[6u * i + 5u] = normalize_unscaled_ok(a, __builtin_fabsf(b)) ? 1u : 0u;
}

}  // namespace

int launch_arith_check(int mode, const uint32_t r[4], unsigned long long count, unsigned long long seed, unsigned long long* d_out,
                       void* stream) {
    hipLaunchKernelGGL(arith_check_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, mode, r[0], r[1], r[2], r[3], count, seed, d_out);
    return (int)hipGetLastError();
}
int launch_arith_pairs(const float* d_x, const float* d_y, uint32_t n, uint32_t* d_out, void* stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(arith_pairs_kernel, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, d_x, d_y, n, d_out);
    return (int)hipGetLastError();
}

}  // namespace mrt
