// Microbenchmark (not product): how fast can gfx950 run the branch-free
// ray-sphere discriminant loop, by sphere-data source (SGPR scalar loads vs LDS
// broadcast) and by packed vs scalar fp32 math?  Decides the data layout of the
// real kernel.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

typedef float float2v __attribute__((ext_vector_type(2)));

// ---- variant 0: plain fma throughput ----
__global__ void k_fma(float* out, int iters) {
  float a[16];
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = x + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = __builtin_fmaf(a[i], y, x);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// ---- variant 1: packed fma throughput ----
__global__ void k_pkfma(float* out, int iters) {
  float2v a[8];
  float x = threadIdx.x * 1e-3f;
  float2v y = {1.0001f, 1.0002f}, xx = {x, x};
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = (float2v){x + i, x - i};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = __builtin_elementwise_fma(a[i], y, xx);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- sphere loops.  sph[i] = (cx,cy,cz,-r*r) ----
struct Ray { float ox, oy, oz, dx, dy, dz; };
__device__ inline Ray make_ray(int tid) {
  Ray r; float t = tid * 0.001f;
  r.ox = 13.f + t; r.oy = 2.f; r.oz = 3.f - t;
  r.dx = -0.9f + t * 0.01f; r.dy = -0.1f; r.dz = -0.2f + t * 0.003f;
  return r;
}
__device__ inline float disc(const Ray& r, float a, float cx, float cy, float cz, float nr2) {
  float ocx = r.ox - cx, ocy = r.oy - cy, ocz = r.oz - cz;
  float b = __builtin_fmaf(ocz, r.dz, __builtin_fmaf(ocy, r.dy, ocx * r.dx));
  float c = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, nr2)));
  return __builtin_fmaf(b, b, -(a * c));
}

// variant 2: spheres via uniform global loads (compiler should emit s_load)
__global__ void __launch_bounds__(256) k_sgpr(const float4* __restrict__ sph, int n, unsigned* out, int reps) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Ray r = make_ray(tid);
  float a = __builtin_fmaf(r.dz, r.dz, __builtin_fmaf(r.dy, r.dy, r.dx * r.dx));
  unsigned acc = 0;
  for (int rep = 0; rep < reps; rep++) {
    for (int i = 0; i < n; i += 32) {
      unsigned bits = 0;
#pragma unroll
      for (int j = 0; j < 32; j++) {
        float4 s = sph[i + j];
        float d = disc(r, a, s.x, s.y, s.z, s.w);
        bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(d), 31);
      }
      acc += __popc(bits) + bits;
    }
    r.ox += 0.01f;
  }
  out[tid] = acc;
}

// variant 3: spheres staged in LDS, broadcast ds_read_b128
__global__ void __launch_bounds__(256) k_lds(const float4* __restrict__ sph, int n, unsigned* out, int reps) {
  extern __shared__ float4 lds[];
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = sph[i];
  __syncthreads();
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Ray r = make_ray(tid);
  float a = __builtin_fmaf(r.dz, r.dz, __builtin_fmaf(r.dy, r.dy, r.dx * r.dx));
  unsigned acc = 0;
  for (int rep = 0; rep < reps; rep++) {
    for (int i = 0; i < n; i += 32) {
      unsigned bits = 0;
#pragma unroll
      for (int j = 0; j < 32; j++) {
        float4 s = lds[i + j];
        float d = disc(r, a, s.x, s.y, s.z, s.w);
        bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(d), 31);
      }
      acc += __popc(bits) + bits;
    }
    r.ox += 0.01f;
  }
  out[tid] = acc;
}

// variant 4: two rays per lane, SGPR spheres
__global__ void __launch_bounds__(256) k_sgpr2(const float4* __restrict__ sph, int n, unsigned* out, int reps) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Ray r0 = make_ray(2 * tid), r1 = make_ray(2 * tid + 1);
  float a0 = __builtin_fmaf(r0.dz, r0.dz, __builtin_fmaf(r0.dy, r0.dy, r0.dx * r0.dx));
  float a1 = __builtin_fmaf(r1.dz, r1.dz, __builtin_fmaf(r1.dy, r1.dy, r1.dx * r1.dx));
  unsigned acc = 0;
  for (int rep = 0; rep < reps; rep++) {
    for (int i = 0; i < n; i += 32) {
      unsigned b0 = 0, b1 = 0;
#pragma unroll
      for (int j = 0; j < 32; j++) {
        float4 s = sph[i + j];
        float d0 = disc(r0, a0, s.x, s.y, s.z, s.w);
        float d1 = disc(r1, a1, s.x, s.y, s.z, s.w);
        b0 = __builtin_amdgcn_alignbit(b0, __float_as_uint(d0), 31);
        b1 = __builtin_amdgcn_alignbit(b1, __float_as_uint(d1), 31);
      }
      acc += __popc(b0) + b0 + __popc(b1) * 3 + b1;
    }
    r0.ox += 0.01f; r1.ox += 0.01f;
  }
  out[tid] = acc;
}

// variant 5: packed math, two spheres per instruction, pair-SoA layout in SGPRs:
// pair p = (c0x,c1x, c0y,c1y, c0z,c1z, nr2_0, nr2_1)
__global__ void __launch_bounds__(256) k_pk(const float2v* __restrict__ sp, int n, unsigned* out, int reps) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Ray r = make_ray(tid);
  float a = __builtin_fmaf(r.dz, r.dz, __builtin_fmaf(r.dy, r.dy, r.dx * r.dx));
  float2v ox = {r.ox, r.ox}, oy = {r.oy, r.oy}, oz = {r.oz, r.oz};
  const float2v dx = {r.dx, r.dx}, dy = {r.dy, r.dy}, dz = {r.dz, r.dz}, aa = {a, a};
  unsigned acc = 0;
  for (int rep = 0; rep < reps; rep++) {
    for (int i = 0; i < n / 2; i += 16) {
      unsigned bits = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const float2v* q = sp + (size_t)(i + j) * 4;
        float2v cx = q[0], cy = q[1], cz = q[2], nr2 = q[3];
        float2v ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
        float2v b = __builtin_elementwise_fma(ocz, dz, __builtin_elementwise_fma(ocy, dy, ocx * dx));
        float2v c = __builtin_elementwise_fma(ocz, ocz, __builtin_elementwise_fma(ocy, ocy, __builtin_elementwise_fma(ocx, ocx, nr2)));
        float2v d = __builtin_elementwise_fma(b, b, -(aa * c));
        bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(d.x), 31);
        bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(d.y), 31);
      }
      acc += __popc(bits) + bits;
    }
    ox += 0.01f;
  }
  out[tid] = acc;
}

template <class F> float timeit(F f, int n = 5) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  f(); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int i = 0; i < n; i++) {
    CHECK(hipEventRecord(e0)); f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  const int blocks = 256 * 8, threads = 256;   // 8 WG/CU -> 8 waves/SIMD
  float* fout; CHECK(hipMalloc(&fout, blocks * threads * 4));
  unsigned* uout; CHECK(hipMalloc(&uout, blocks * threads * 4 * 2));
  const int iters = 4096;
  float ms = timeit([&] { k_fma<<<blocks, threads>>>(fout, iters); });
  printf("fma      : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * 16 * iters * blocks * threads / ms * 1e-9);
  ms = timeit([&] { k_pkfma<<<blocks, threads>>>(fout, iters); });
  printf("pk_fma   : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * 16 * iters * blocks * threads / ms * 1e-9);

  for (int n : {512, 10240}) {
    std::vector<float> h(n * 4);
    srand(1);
    for (int i = 0; i < n; i++) {
      h[4 * i] = (rand() % 2200) * 0.01f - 11; h[4 * i + 1] = 0.2f; h[4 * i + 2] = (rand() % 2200) * 0.01f - 11;
      h[4 * i + 3] = -0.04f;
    }
    std::vector<float> hp(n * 4);
    for (int pI = 0; pI < n / 2; pI++) for (int k = 0; k < 4; k++) { hp[pI * 8 + 2 * k] = h[(2 * pI) * 4 + k]; hp[pI * 8 + 2 * k + 1] = h[(2 * pI + 1) * 4 + k]; }
    float4* ds; CHECK(hipMalloc(&ds, n * 16)); CHECK(hipMemcpy(ds, h.data(), n * 16, hipMemcpyHostToDevice));
    float2v* dp; CHECK(hipMalloc(&dp, n * 16)); CHECK(hipMemcpy(dp, hp.data(), n * 16, hipMemcpyHostToDevice));
    int reps = n == 512 ? 200 : 10;
    for (int wg_per_cu : {2, 4, 8}) {
      int nb = 256 * wg_per_cu;
      double tests = (double)nb * threads * n * reps;
      ms = timeit([&] { k_sgpr<<<nb, threads>>>(ds, n, uout, reps); });
      printf("n=%5d wg/cu=%d sgpr   : %8.3f ms  %.2f Gtest/s\n", n, wg_per_cu, ms, tests / ms * 1e-6);
      if (n * 16 <= 160 * 1024 / wg_per_cu) {
        ms = timeit([&] { k_lds<<<nb, threads, n * 16>>>(ds, n, uout, reps); });
        printf("n=%5d wg/cu=%d lds    : %8.3f ms  %.2f Gtest/s\n", n, wg_per_cu, ms, tests / ms * 1e-6);
      }
      ms = timeit([&] { k_sgpr2<<<nb, threads>>>(ds, n, uout, reps); });
      printf("n=%5d wg/cu=%d sgpr2  : %8.3f ms  %.2f Gtest/s\n", n, wg_per_cu, ms, 2 * tests / ms * 1e-6);
      ms = timeit([&] { k_pk<<<nb, threads>>>(dp, n, uout, reps); });
      printf("n=%5d wg/cu=%d pk     : %8.3f ms  %.2f Gtest/s\n", n, wg_per_cu, ms, tests / ms * 1e-6);
    }
  }
  return 0;
}
