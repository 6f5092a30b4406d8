// Tile-queue order: sort this shard's 8x8 tiles by the cost (bounce-loop trips of their pixels)
// measured in the previous frame -- or in a small pilot pass before the first frame -- heaviest
// first.  render_kernel's persistent waves pull tiles from the queue in that order, so the
// lightest tiles are the last ones anybody starts.  No reference counterpart: the reference
// issues one full-screen draw (raytracer/src/lib.rs:262-267) and leaves scheduling to the GPU.
// The order only affects WHICH wave renders a tile and when, never its pixels (every pixel owns
// its RNG stream and output texel).
//
// A bucket sort is enough (ties may land in any order): key = 5-bit exponent | 5-bit
// mantissa of the cost, 1024 buckets, three tiny launches.

#include <hip/hip_runtime.h>
#include "mrt_internal.h"

namespace mrt {
namespace {

constexpr uint32_t kBuckets = 1024;

// monotonic in cost; bucket 1023 = heaviest
__device__ __forceinline__ uint32_t cost_bucket(uint32_t cost) {
    if (cost < 32u) return cost;                       // exact for tiny costs
    const uint32_t e = 31u - (uint32_t)__builtin_clz(cost);   // 5..31
    const uint32_t m = (cost >> (e - 5u)) & 31u;              // 5 bits below the leading one
    const uint32_t key = (e - 4u) * 32u + m;                  // 32.. 895
    return key < kBuckets ? key : kBuckets - 1u;
}

// Equal costs are common (every pure-sky tile costs exactly spp trips per pixel), so global atomics per
// tile would pile up on one address: both passes first combine a block's 1024 tiles in LDS.
constexpr uint32_t kTilesPerBlock = 1024;

__global__ void __launch_bounds__(256) tile_hist_kernel(const uint32_t* __restrict__ cost, uint32_t* hist, uint32_t n) {
    __shared__ uint32_t local[kBuckets];
    for (uint32_t k = threadIdx.x; k < kBuckets; k += 256) local[k] = 0;
    __syncthreads();
    for (uint32_t j = 0; j < kTilesPerBlock / 256; j++) {
        const uint32_t i = blockIdx.x * kTilesPerBlock + j * 256 + threadIdx.x;
        if (i < n) atomicAdd(&local[cost_bucket(cost[i])], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kBuckets; k += 256)
        if (local[k]) atomicAdd(&hist[k], local[k]);
}

// one block of 1024 threads: hist[b] <- number of tiles in heavier buckets (descending exclusive scan)
__global__ void __launch_bounds__(1024) tile_scan_kernel(uint32_t* hist) {
    __shared__ uint32_t s[kBuckets];
    const uint32_t t = threadIdx.x;
    s[t] = hist[kBuckets - 1u - t];                    // reversed: index 0 = heaviest bucket
    __syncthreads();
    for (uint32_t off = 1; off < kBuckets; off <<= 1) {
        const uint32_t v = (t >= off) ? s[t - off] : 0u;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    hist[kBuckets - 1u - t] = (t == 0) ? 0u : s[t - 1u];
}

__global__ void __launch_bounds__(256) tile_scatter_kernel(const uint32_t* __restrict__ cost, uint32_t* offsets,
                                                           uint32_t* __restrict__ order, uint32_t n) {
    __shared__ uint32_t local[kBuckets];      // count, then this block's base offset, per bucket
    for (uint32_t k = threadIdx.x; k < kBuckets; k += 256) local[k] = 0;
    __syncthreads();
    uint32_t bucket[kTilesPerBlock / 256], rank[kTilesPerBlock / 256];
    for (uint32_t j = 0; j < kTilesPerBlock / 256; j++) {
        const uint32_t i = blockIdx.x * kTilesPerBlock + j * 256 + threadIdx.x;
        bucket[j] = 0; rank[j] = 0;
        if (i < n) { bucket[j] = cost_bucket(cost[i]); rank[j] = atomicAdd(&local[bucket[j]], 1u); }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kBuckets; k += 256)
        if (local[k]) local[k] = atomicAdd(&offsets[k], local[k]);     // reserve this block's range
    __syncthreads();
    for (uint32_t j = 0; j < kTilesPerBlock / 256; j++) {
        const uint32_t i = blockIdx.x * kTilesPerBlock + j * 256 + threadIdx.x;
        if (i < n) order[local[bucket[j]] + rank[j]] = i;
    }
}

}  // namespace

int launch_sort_tiles(const uint32_t* cost, uint32_t* order, uint32_t* scratch, uint32_t n_tiles, void* stream) {
    if (n_tiles == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, kBuckets * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    const uint32_t blocks = (n_tiles + kTilesPerBlock - 1u) / kTilesPerBlock;
    hipLaunchKernelGGL(tile_hist_kernel, dim3(blocks), dim3(256), 0, st, cost, scratch, n_tiles);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, scratch);
    hipLaunchKernelGGL(tile_scatter_kernel, dim3(blocks), dim3(256), 0, st, cost, scratch, order, n_tiles);
    return (int)hipGetLastError();
}

}  // namespace mrt
