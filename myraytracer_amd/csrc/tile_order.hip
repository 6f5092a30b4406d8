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
//
// The sort of frame n+2's queue runs while frame n+1's persistent render grid owns nearly all of every CU's LDS
// (157.5 of 160 KB at C3) and 5 of its 8 wave slots per SIMD.  A kernel that needs LDS, or a 1,024-thread
// workgroup, cannot become resident next to it and stays queued until that grid drains -- and the next render
// with it (round 1: render n+1 started only ~14 ms before render n ended).  So these kernels use NO LDS, few
// registers and 256-thread (one wave per SIMD) or single-wave workgroups: they slip in beside the render
// waves.  Equal costs are common (every pure-sky tile costs exactly spp trips per pixel), so a wave first combines
// its lanes per bucket with ballots and issues one global atomic per distinct bucket.

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

// Every lane with `valid` adds 1 to counters[bucket] and learns its own position (the counter's value before
// the wave's addition + its rank among the wave's lanes of the same bucket): one atomic per distinct bucket.
__device__ __forceinline__ uint32_t wave_bucket_add(uint32_t* counters, uint32_t bucket, bool valid) {
    uint32_t pos = 0;
    bool pending = valid;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(pending);
    while (todo != 0ull) {                                                     // wave-uniform loop
        const int leader = __builtin_ctzll(todo);
        const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)bucket, leader);
        const bool mine = pending && bucket == lb;
        const unsigned long long group = __builtin_amdgcn_ballot_w64(mine);
        uint32_t base = 0;
        if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(&counters[lb], (uint32_t)__popcll(group));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (mine) {
            pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(group >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)group, 0u));
            pending = false;
        }
        todo &= ~group;
    }
    return pos;
}

__global__ void __launch_bounds__(256) tile_hist_kernel(const uint32_t* __restrict__ cost, uint32_t* hist, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool ok = i < n;
    (void)wave_bucket_add(hist, ok ? cost_bucket(cost[i]) : 0u, ok);
}

// one wave: hist[b] <- number of tiles in heavier buckets (descending exclusive scan); lane l owns the 16
// buckets 1023 - 16 l ... 1008 - 16 l, heaviest first
__global__ void __launch_bounds__(64) tile_scan_kernel(uint32_t* hist) {
    const uint32_t lane = threadIdx.x;
    uint32_t v[16], sum = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) { v[j] = hist[kBuckets - 1u - (16u * lane + (uint32_t)j)]; sum += v[j]; }
    uint32_t incl = sum;                               // inclusive scan of the lanes' sums
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += up;
    }
    uint32_t run = incl - sum;
#pragma unroll
    for (int j = 0; j < 16; j++) { hist[kBuckets - 1u - (16u * lane + (uint32_t)j)] = run; run += v[j]; }
}

__global__ void __launch_bounds__(256) tile_scatter_kernel(const uint32_t* __restrict__ cost, uint32_t* offsets,
                                                           uint32_t* __restrict__ order, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool ok = i < n;
    const uint32_t pos = wave_bucket_add(offsets, ok ? cost_bucket(cost[i]) : 0u, ok);
    if (ok) order[pos] = i;
}

// probe_stream_concurrency (api.cpp): one wave that stays resident for `ticks` of the 100 MHz wall clock and then ends -- or
// after `max_polls` polls, whichever comes first: the exit never depends on the clock alone.  Writes {start, end} ticks.
__global__ void __launch_bounds__(64) hold_kernel(unsigned long long ticks, uint32_t max_polls, unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long now = t0;
    for (uint32_t i = 0; i < max_polls && now - t0 < ticks; i++) {
        __builtin_amdgcn_s_sleep(32);
        now = __builtin_amdgcn_s_memrealtime();
    }
    if (threadIdx.x == 0 && out) { out[0] = t0; out[1] = now; }
}

}  // namespace

int launch_hold(unsigned long long ticks, uint32_t max_polls, unsigned long long* out, void* stream) {
    hipLaunchKernelGGL(hold_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks, max_polls, out);
    return (int)hipGetLastError();
}

int launch_sort_tiles(const uint32_t* cost, uint32_t* order, uint32_t* scratch, uint32_t n_tiles, void* stream) {
    if (n_tiles == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, kBuckets * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    const uint32_t blocks = (n_tiles + 255u) / 256u;
    hipLaunchKernelGGL(tile_hist_kernel, dim3(blocks), dim3(256), 0, st, cost, scratch, n_tiles);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(64), 0, st, scratch);
    hipLaunchKernelGGL(tile_scatter_kernel, dim3(blocks), dim3(256), 0, st, cost, scratch, order, n_tiles);
    return (int)hipGetLastError();
}

}  // namespace mrt
