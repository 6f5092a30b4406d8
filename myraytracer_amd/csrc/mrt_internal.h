// Internal types shared by the HIP kernels (kernels.hip) and the host side of the C ABI
// (api.cpp).  Not installed; the public contract is include/myraytracer_amd.h.
#pragma once
#include <stdint.h>
#include "../../include/myraytracer_amd.h"
#include "../../include/myraytracer_amd_debug.h"

namespace mrt {

constexpr uint32_t kBandRows = 8;        // shard granule: 8 image rows (one row of 8x8 wave tiles)
constexpr uint32_t kTileW = 8;           // one 64-lane wave (= one workgroup) covers an 8x8 pixel tile
constexpr uint32_t kChunk = 16;          // spheres per chunk of the discriminant sweep (one u16 sign mask)
constexpr uint32_t kGroup = 8;           // records per scalar-load group; the sweep list is padded to this
constexpr uint32_t kClusterK = 4;        // spheres per sweep record (cluster)
constexpr uint32_t kMaxSpheres = 1u << 20;
constexpr uint32_t kMaxLevels = 4;       // levels of bounding spheres above the member spheres
// Conservativeness of a bounding-sphere test: the ray direction is stretched by kBoundStretch
// in the test and the stored radius is kBoundInflate x the enclosing radius.
constexpr float kBoundStretch = 1.0001f;
constexpr double kBoundInflate = 1.015;
constexpr uint32_t kMaxFrameBatch = 32;   // frames one render launch may cover (stream mode, mrt_render)
constexpr uint32_t kQueueCap = 320;      // a work queue of the walk: < 64 left over + 4 x 64 pushed by one round
constexpr uint32_t kStackReserve = 16;   // large scenes' work stack: entries beyond its capacity a one-item round may use (3 per level)
constexpr uint32_t kMaxDirect = 4;       // very large spheres tested by every ray directly, outside the hierarchy

// (cx, cy, cz, -(r*r)): the only per-sphere data the discriminant loop reads.  Derived on
// the host from the reference's SoA arrays (centres: vec4_f32_data, radii: f32_data;
// lib.rs:722-799) and padded to a multiple of kGroup with never-hit entries (w = +inf).
struct alignas(16) SphereRec { float cx, cy, cz, neg_r2; };

// Axis-aligned box of the member spheres under a node of the hierarchy (large scenes: the walk's second, much tighter bound
// -- a kd-built group of spheres on a plane fills its box, not its bounding sphere).
// BoxFull: what the host derives per node (api.cpp, build_boxes) and the diagnostics report: centre, half extents (measured
// from the f32 centre, rounded up) and the two coefficients of the test's slack K = kc X + kpad (X = |p|^2 or |p|_1 of the ray
// origin relative to the centre, per scene: KParams::box_quad); kc is ONE value per scene (KParams::box_kc).
// BoxRec: what the kernel reads, 24 bytes: the centre and the half extents WITH kpad folded in (e + kpad, rounded up) -- on the
// axis d x e_i the slack kpad (|d_j| + |d_k|) that gives covers what the "+ kpad" of the test covered (api.cpp, pack_boxes) --
// so an inner item's four children are 96 bytes instead of 128: a large scene's rounds wait for the vector-memory path's
// 64 bytes per clock and CU.  A never-hit box has extents -3e38.
struct BoxFull { float cx, cy, cz, ex, ey, ez, kc, kpad; };
struct alignas(8) BoxRec { float cx, cy, cz, ex, ey, ez; };

// Everything one raytrace pass needs, passed by value as kernel arguments (-> SGPRs).
// Mirrors the three bind groups of State::redraw (lib.rs:262-265): Locals + seeds,
// World + data arrays, previous framebuffer.
struct KParams {
    mrt_locals locals;          // shader.wgsl:8-17
    mrt_world world;            // shader.wgsl:178-182 (+ dielectric range)
    mrt_camera_raw cam;
    uint32_t n_spheres;         // world.spheres.length
    uint32_t n_padded;          // cluster records, multiple of kGroup
    uint32_t mask_chunks;       // chunks of kChunk records per sweep block: min(16, ceil(n_padded / kChunk))
    uint32_t shard_rank, shard_world;
    uint32_t cus;               // compute units of the device (host-side launch sizing only)
    const SphereRec* spheres;   // n_spheres records in the reference's order (exact tests)
    // Bounding-sphere hierarchy (api.cpp build_hierarchy): level 0 = the member spheres in cluster order
    // (4 per cluster, short clusters padded with never-hit records), level 1 = the clusters' bounds
    // (cx,cy,cz,-R^2), level k+1 = bounds of 4 consecutive level-k nodes; node j of level k has the
    // children 4j..4j+3 of level k-1.  The sweep runs over the TOP level (`levels`): `clusters`, n_padded
    // records (multiple of kGroup, padded with never-hit entries).  `nodes` holds levels 0..levels-1,
    // level k at level_base[k]; member_index[] is each member's index in the reference's sphere order.
    const SphereRec* clusters;
    // the same top-level records as the A operand of the matrix-core sweep (kernels.hip, mfma_sweep_tile):
    // per tile of 32 records 64 lanes x 8 bf16, record order within a tile permuted to the result layout;
    // use_mfma selects that variant of the sweep (api.cpp decides per scene and camera)
    const uint16_t* top_mfma;
    uint32_t use_mfma;
    float mfma_origin[3];       // the records of top_mfma are relative to this point (centre of their bounding box)
    // The ray-side factors of the matrix-core sweep (api.cpp, fill_scene_params): with K a power of two such that
    // |K oc.ds| <= 1/2 for every ray the sweep admits, {kBoundStretch K, 2 K^2, -(1 - 2^-13) K^2, the largest admitted
    // |o - mfma_origin|^2}, and -K^2 as a pair of bf16 (kernels.hip, mfma_ray_operands)
    float mfma_scale[4];
    uint32_t mfma_neg_k2_pair;
    const SphereRec* nodes;
    const uint32_t* member_index;
    uint32_t levels, n_nodes, n_members, gen_cap;
    uint32_t level_base[kMaxLevels];
    // Large scenes (more than 1,024 member slots): the axis-aligned boxes of the hierarchy's nodes, numbered TOP-DOWN over the
    // complete 4-ary tree below the n_padded swept records: depth t occupies [o_t, o_t + n_padded 4^t), o_t = n_padded
    // (4^t - 1) / 3, so the children of node g -- whatever its depth -- are 4 g + n_padded .. + 3 (never-hit boxes where the
    // tree has no node).  box_cluster_first = o_(levels - 1): the nodes from there on are the clusters (node g = cluster
    // g - box_cluster_first, members 4 m .. 4 m + 3 of level 0); box_cluster_parent_first = o_(levels - 2): the nodes from
    // there on have clusters as children.  box_quad: the form of the slack (selects the kernel instantiation).  gen_cap is
    // the capacity of the wave's work stack.  Null / 0 for small scenes.
    const BoxRec* boxes;
    uint32_t box_cluster_first, box_cluster_parent_first, box_quad;
    float box_kc;               // the slack's coefficient of X, one per scene (api.cpp, build_boxes)
    // the first box_lds_count boxes of that numbering (the swept top, and the level below it where it fits) are copied into
    // the workgroup's LDS: what the owners' filter and the first inner rounds read (kernels.hip)
    uint32_t box_lds_count;
    // Spheres far larger than the rest (a ground sphere) are candidates for nearly every ray: up to kMaxDirect
    // of them stay out of the hierarchy and every ray evaluates their discriminant itself, from SGPRs.
    // They are the members direct_first .. direct_first + n_direct - 1 of level 0.
    uint32_t n_direct, direct_first;
    SphereRec direct[kMaxDirect];
    uint32_t direct_index[kMaxDirect];      // their indices in the reference's sphere order
    const float* vec4_data;     // r_vec4_f32_data (shader.wgsl:189-190), 4 floats per texel
    const float* f32_data;      // r_f32_data
    const int32_t* i32_data;    // r_i32_data
    // per sphere, in the reference's order: (cx, cy, cz, radius) (albedo r, g, b, fuzz | ior) -- copies of the
    // entries of the three arrays above that shading the sphere reads (api.cpp)
    const float* shade;
    const uint32_t* seeds;      // r_rands: local_rows x W x [u32;4]
    const float* prev;          // r_framebuffer: local_rows x W x rgba
    float* out;                 // render target
    unsigned long long* counters;  // 4 x u64 (mrt_counters) or null
    uint32_t count_draws;       // launch the instantiation that also counts the RNG draws (mrt_set_draw_counting)
    // the frame's tile queue: persistent waves pull tiles tile_order[atomicAdd(tile_queue, 1)]
    const uint32_t* tile_order; // n_tiles tile ids, heaviest first (tile_order.hip), or null = index order
    uint32_t* tile_queue;       // one u32, zeroed before every launch
    uint32_t tiles_x, n_tiles;  // tiles per band row; tiles in this shard
    uint32_t pilot_spp;         // samples per pixel of the cost-estimating pilot launch
    uint32_t* tile_cost;        // n_tiles: sum of its pixels' loop trips in this frame, or null
    void* pix_acc;              // per local pixel (and per block of samples): colour sum + cost (16 B), render -> finalize
    // counter-RNG mode: a pixel's samples are independent, so the frame is n_blocks layers (block b = samples
    // [64 b, 64 b + 64)), each summed into pix_acc[b * pix_stride + texel]; finalize adds the layers in order.  1 otherwise.
    // Stream mode: layer b = frame b of a batch of consecutive frames rendered by one launch (mrt_render), with
    // rng_shuffle layer_shuffle[b]; layer_shuffle[0] is always the (first) frame's shuffle.
    // queue_layers = layers the tile QUEUE holds (n_tiles x queue_layers items): n_blocks, except for a batch of short frames
    // of the stream mode, where the queue holds every tile once and the lane that takes a pixel renders it for all
    // lane_frames frames of the batch, frame b into layer b (lane_frames = n_blocks then, 1 otherwise).
    uint32_t n_blocks, pix_stride, queue_layers, lane_frames;
    uint32_t layer_shuffle[kMaxFrameBatch][4];
    unsigned long long* wave_log;  // diagnostic (-DMRT_STAMPS builds): 4 x u64 per wave, or null
    // mrt_debug_world_hit (the DBG instantiation of render_kernel): rays in (origin xyz, direction xyz), out: winner
    // {sphere index | -1, bits of t} per ray and the bitmap of spheres that reached the root tests
    const float* dbg_rays;
    int32_t* dbg_hit;
    uint32_t* dbg_cand;
    uint32_t dbg_words;         // bitmap words per ray
};

// api.cpp: message behind mrt_last_error(NULL), for failures of entry points that have no context
void set_global_error(const char* msg);

// host-callable launchers (kernels.hip)
// queue reset + n_waves persistent render waves (a pilot launch also runs its cost-only finalize)
int launch_render(const KParams& p, bool pilot, uint32_t n_waves, void* stream, uint32_t* which = nullptr);
// the per-tile finalize pass of a rendered frame: colour sums -> framebuffer, tile costs
int launch_finalize(const KParams& p, void* stream);
int launch_debug_world_hit(const KParams& p, uint32_t n_waves, void* stream);
int render_waves_per_cu(int* out);
// host only: {LDS bytes of one render workgroup, workgroups per CU} for p's scene layout; the work-stack capacity (entries)
// of a large scene's wave with `mask_chunks` chunks of candidate masks
void render_lds_layout(const KParams& p, uint32_t out[2]);
uint32_t render_resident_waves(const KParams& p);
uint32_t large_scene_stack_cap(uint32_t mask_chunks, uint32_t box_lds_count);
// how many leading boxes of the top-down numbering a large scene's workgroup keeps in LDS: the top level (n_top padded records)
// and the 4 n_top slots of the level below, or the top alone, or none -- the most that leaves the waves their work stacks
uint32_t large_scene_box_lds_count(uint32_t n_top_padded, uint32_t levels, uint32_t mask_chunks, uint32_t cap);
constexpr uint32_t kBoxLdsCap = 1024;    // never more boxes (32 B each) than this in a workgroup's LDS
// tile_order.hip: order[] = tile ids sorted by cost[] descending (bucket sort; ties in any order).
// scratch: 1024 u32.
int launch_sort_tiles(const uint32_t* cost, uint32_t* order, uint32_t* scratch, uint32_t n_tiles, void* stream);
// one single-wave kernel that stays resident for `ticks` of the 100 MHz clock (at most max_polls polls); out (optional, device-
// visible) gets its {start, end} ticks
int launch_hold(unsigned long long ticks, uint32_t max_polls, unsigned long long* out, void* stream);
int launch_fill_seeds(uint32_t* seeds, uint64_t seed, uint32_t width, uint32_t height,
                      uint32_t shard_rank, uint32_t shard_world, uint32_t local_bands, void* stream);

// mrt_debug_arith / mrt_debug_arith_pairs: div_unscaled / sqrt_unscaled against `/` and sqrtf() on the device
int launch_arith_check(int mode, const uint32_t r[4], unsigned long long count, unsigned long long seed, unsigned long long* d_out,
                       void* stream);
int launch_arith_pairs(const float* d_x, const float* d_y, uint32_t n, uint32_t* d_out, void* stream);

}  // namespace mrt
