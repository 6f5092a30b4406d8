// Device-side helpers shared by kernels.hip (the render path) and debug_kernels.hip (the diagnostic kernels that test
// them): the correctly rounded division / square root without their operand-scaling steps, the per-wave operand tests
// of their two tested call sites, and the counter-based SplitMix64 of the seed texture.  Internal: not installed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrt {

// ---- IEEE division and square root without the operand scaling ------------------------------------------
// `x / y` and sqrtf() compile to the correctly rounded expansions (v_div_scale x2, v_rcp, 7 fma-class operations,
// v_div_fmas, v_div_fixup; a range test, v_sqrt, two residuals, two selects, a rescale and a class test).  The
// functions below are those expansions WITHOUT the steps that only act on extreme operands, so wherever
// v_div_scale_f32 would pass both operands through unscaled -- numerator and denominator finite and non-zero,
// |n| >= 2^-102, the denominator and its reciprocal normal, -126 < exponent(n) - exponent(d) < 96 -- respectively
// x >= 2^-96 finite, they execute the same operations on the same values and return the same bits as `/` and
// sqrtf().  One refined reciprocal serves every numerator over the same denominator.  Each call site states why its
// operands are in that range, or tests it and takes `/` and sqrtf() otherwise.
struct Divisor { float d, r; };
__device__ __forceinline__ Divisor divisor_of(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    Divisor D; D.d = d; D.r = r;
    return D;
}
__device__ __forceinline__ float div_unscaled(float n, const Divisor D) {
    float q = n * D.r;
    float e = __builtin_fmaf(-D.d, q, n);
    q = __builtin_fmaf(e, D.r, q);
    e = __builtin_fmaf(-D.d, q, n);
    return __builtin_fmaf(e, D.r, q);
}
// (x = +0 -> +0: the neighbour below is a NaN pattern, whose comparison is false, and the residual of the neighbour
// above is +0, not > 0)
__device__ __forceinline__ float sqrt_unscaled(float x) {
    float s = __builtin_amdgcn_sqrtf(x);                 // within 1 ulp: the answer is s or one of its neighbours
    const float down = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_down = __builtin_fmaf(-down, s, x), r_up = __builtin_fmaf(-up, s, x);
    s = (r_down <= 0.0f) ? down : s;
    s = (r_up > 0.0f) ? up : s;
    return s;
}
constexpr float kDivMinNum = 0x1p-90f, kDivMinDen = 0x1p-30f;       // the tested call sites' bounds (upper bounds: 2^30, from
                                                                    // the ABI's |coordinate| <= 1e7, api.cpp)
// The two call sites whose operands are TESTED (per wave: any lane failing sends its wave down the literal `/` and sqrtf()):
// the hit normal (at - centre) / radius -- every component at least 2^-90 in magnitude (in particular not 0), |radius| >= 2^-30 --
// and normalize(dir) = dir / sqrt(dot(dir, dir)) -- the same for the components, the squared length in [2^-60, 2^60).
// mrt_debug_arith_pairs evaluates these very predicates for caller-supplied operands (tests/test_gpu_arith.py).
__device__ __forceinline__ bool normal_unscaled_ok(float rel_min_abs, float radius) {
    return rel_min_abs >= kDivMinNum && __builtin_fabsf(radius) >= kDivMinDen;
}
__device__ __forceinline__ bool normalize_unscaled_ok(float dd, float nd_min_abs) {
    const bool dd_ok = (__float_as_uint(dd) - 0x21800000u) < (0x5D800000u - 0x21800000u);      // bits of 2^-60, 2^60
    return dd_ok && nd_min_abs >= kDivMinNum;
}
// Seed texture (Subject::new, lib.rs:389-415) generated on the device: SplitMix64 used as a
// counter-based generator keyed by the GLOBAL pixel index, two outputs per pixel.
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1u) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace mrt
