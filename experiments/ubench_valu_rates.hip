// Microbenchmark (not product): issue rate of the VALU instruction classes the walk / shade / RNG code
// is made of, relative to v_fma_f32, on gfx950.  16 independent chains per lane, inline asm so that the
// compiler cannot fold or fuse them.  Build: hipcc -O3 --offload-arch=gfx950 -o ubench_valu_rates ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

#define KERNEL(name, ASM)                                                           \
__global__ void __launch_bounds__(256) name(uint32_t* out, int iters) {             \
  uint32_t a[16]; uint32_t x = threadIdx.x * 2654435761u + 12345u, y = 0x3f800123u; \
  _Pragma("unroll") for (int i = 0; i < 16; i++) a[i] = x + i * 977u;               \
  for (int it = 0; it < iters; it++) {                                              \
    _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : "v"(x), "v"(y)); \
  }                                                                                 \
  uint32_t s = 0; _Pragma("unroll") for (int i = 0; i < 16; i++) s ^= a[i];         \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                   \
}
KERNEL(k_fma, "v_fma_f32 %0, %0, %2, %1")
KERNEL(k_fma_k, "v_fma_f32 %0, %0, 2.0, %1")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_fma_s, "v_fma_f32 %0, %0, s20, %1")
KERNEL(k_fma_2v, "v_fma_f32 %0, %0, %0, %1")
KERNEL(k_cndmask_s, "v_cndmask_b32 %0, %0, %1, s[22:23]")
KERNEL(k_cmp_s, "v_cmp_lt_f32 s[22:23], %0, %1")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 2, 2")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0")
KERNEL(k_min, "v_min_f32 %0, %0, %1")
KERNEL(k_or, "v_or_b32 %0, %0, %1")
KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL(k_add_f32, "v_add_f32 %0, %0, %2")
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %2")
KERNEL(k_sub_f32, "v_sub_f32 %0, %1, %0")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 31")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_ffbh, "v_ffbh_u32 %0, %0")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_fmac_dpp, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 5")

// packed fp32 (two lanes' worth of math per wave-instruction, operands in VGPR pairs): does a 3-distinct-operand
// v_pk_fma_f32 cost what a 3-distinct-VGPR v_fma_f32 costs?  (8 independent chains of pairs = 16 results, as above)
#define KERNEL64(name, ASM)                                                         \
__global__ void __launch_bounds__(256) name(uint32_t* out, int iters) {             \
  uint64_t a[8]; uint64_t x = (threadIdx.x * 2654435761ull + 12345ull) | 0x3f8001233f800123ull, y = 0x3f8001233f800456ull; \
  _Pragma("unroll") for (int i = 0; i < 8; i++) a[i] = x + i * 977ull;              \
  for (int it = 0; it < iters; it++) {                                              \
    _Pragma("unroll") for (int r = 0; r < 2; r++)                                   \
    _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(a[i]) : "v"(x), "v"(y)); \
  }                                                                                 \
  uint64_t s = 0; _Pragma("unroll") for (int i = 0; i < 8; i++) s ^= a[i];          \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));           \
}
KERNEL64(k_pk_fma, "v_pk_fma_f32 %0, %0, %2, %1")
KERNEL64(k_pk_fma_2v, "v_pk_fma_f32 %0, %0, %0, %1")
KERNEL64(k_pk_mul, "v_pk_mul_f32 %0, %0, %2")
KERNEL64(k_pk_add, "v_pk_add_f32 %0, %0, %2")

template <typename K> void run(const char* name, K k, uint32_t* d, int waves_per_simd) {
  const int iters = 60000, cus = 256;
  dim3 grid(cus * waves_per_simd), block(256);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, grid, block, 0, 0, d, 100);
  CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k, grid, block, 0, 0, d, iters); CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double insts = (double)cus * 4 * waves_per_simd * iters * 16;     // wave-instructions
  printf("%-12s %d waves/SIMD: %7.3f ms  %6.2f G wave-inst/s  = %5.2f cycles per wave-inst per SIMD at 2.4 GHz\n", name, waves_per_simd,
         ms, insts / ms * 1e-6, 2.4e9 * ms * 1e-3 * 1024 / insts);
}
int main() {
  uint32_t* d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
#define RUN(k) run(#k, k, d, 4)
  RUN(k_fma); RUN(k_fma_k); RUN(k_fmac); RUN(k_fma_s); RUN(k_fma_2v); RUN(k_cndmask_s); RUN(k_cmp_s); RUN(k_bfe); RUN(k_lshr); RUN(k_min); RUN(k_or); RUN(k_sub_u32); RUN(k_add_f32); RUN(k_mul_f32); RUN(k_sub_f32); RUN(k_and); RUN(k_xor); RUN(k_add_u32); RUN(k_lshl); RUN(k_lshl_add);
  RUN(k_alignbit); RUN(k_cndmask); RUN(k_cmp); RUN(k_cvt); RUN(k_mov); RUN(k_bcnt); RUN(k_mbcnt); RUN(k_mul_lo); RUN(k_sqrt); RUN(k_ffbh);
  RUN(k_pk_fma); RUN(k_pk_fma_2v); RUN(k_pk_mul); RUN(k_pk_add);      // 16 wave-instructions per iteration, each TWO operations per lane
  run("k_fma", k_fma, d, 2); run("k_and", k_and, d, 2); run("k_fma", k_fma, d, 8); run("k_and", k_and, d, 8);
  return 0;
}
