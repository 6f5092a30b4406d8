// Host side of the C ABI (include/myraytracer_amd.h): the in-scope parts of the
// reference's `State` (raytracer/src/lib.rs:206-308) -- Subject (Locals + seed texture),
// Object (scene packing + upload), DoubleFramebuffers (ping-pong accumulators) and the
// tail of State::redraw (accumulation weights, reshuffle) -- over HIP allocations and one
// kernel launch per frame instead of wgpu bind groups and a full-screen draw.
//
// There is deliberately no CPU fallback: without a gfx950 device mrt_create fails with
// MRT_ERR_NO_DEVICE.

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "mrt_internal.h"
#include "mrt_ctx.h"

static_assert(sizeof(mrt_args) == 20, "mrt_args layout");
static_assert(sizeof(mrt_locals) == 48, "Locals is 48 bytes (lib.rs:368-377)");
static_assert(sizeof(mrt_world) == 80, "raw::World 64 B + DielectricRange 16 B");
static_assert(offsetof(mrt_world, dielectrics) == MRT_WORLD_BYTES_REFERENCE, "raw::World is the first 64 bytes of mrt_world");
static_assert(sizeof(mrt_sphere) == 36, "mrt_sphere layout");
static_assert(sizeof(mrt_camera) == 52, "mrt_camera layout");
static_assert(sizeof(mrt_camera_raw) == 80, "mrt_camera_raw layout");
static_assert(sizeof(mrt::SphereRec) == 16, "SphereRec layout");

namespace {

thread_local std::string g_err;   // for failures before a ctx exists

}  // namespace

namespace mrt {
int fail(mrt_ctx* ctx, int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_err = buf;
    return status;
}
}  // namespace mrt

namespace {

using mrt::fail;
using mrt::local_texels;
using mrt::total_bands;

void free_frame_buffers(mrt_ctx* c) {
    if (c->d_seeds) (void)hipFree(c->d_seeds);
    if (c->d_fb[0]) (void)hipFree(c->d_fb[0]);
    if (c->d_fb[1]) (void)hipFree(c->d_fb[1]);
    for (auto& S : c->slot) {
        if (S.d_tile_cost) (void)hipFree(S.d_tile_cost);
        if (S.d_tile_order) (void)hipFree(S.d_tile_order);
        if (S.d_sort_scratch) (void)hipFree(S.d_sort_scratch);
        if (S.d_pix_acc) (void)hipFree(S.d_pix_acc);
        S.d_tile_cost = S.d_tile_order = S.d_sort_scratch = nullptr;
        S.d_pix_acc = nullptr;
        S.pix_acc_layers = 0; S.cost_first_layer = 0; S.cost_layers = 1;
        S.cost_valid = false;
    }
    c->d_seeds = nullptr; c->d_fb[0] = c->d_fb[1] = nullptr;
}

void free_world(mrt_ctx* c) {
    if (c->d_spheres) (void)hipFree(c->d_spheres);
    if (c->d_clusters) (void)hipFree(c->d_clusters);
    if (c->d_nodes) (void)hipFree(c->d_nodes);
    if (c->d_boxes) (void)hipFree(c->d_boxes);
    if (c->d_boxes_open) (void)hipFree(c->d_boxes_open);
    if (c->d_shade) (void)hipFree(c->d_shade);
    if (c->d_top_mfma) (void)hipFree(c->d_top_mfma);
    if (c->d_member_index) (void)hipFree(c->d_member_index);
    if (c->d_vec4) (void)hipFree(c->d_vec4);
    if (c->d_f32) (void)hipFree(c->d_f32);
    if (c->d_i32) (void)hipFree(c->d_i32);
    c->d_spheres = nullptr; c->d_clusters = nullptr; c->d_nodes = nullptr; c->d_boxes = nullptr; c->d_boxes_open = nullptr; c->d_shade = nullptr; c->d_top_mfma = nullptr; c->d_member_index = nullptr; c->d_vec4 = nullptr; c->d_f32 = nullptr; c->d_i32 = nullptr;
    c->have_world = false;
}

// Subject::new + DoubleFramebuffers::new for the current shard (lib.rs:389-415, 514-538)
int alloc_frame_buffers(mrt_ctx* c) {
    free_frame_buffers(c);
    const uint32_t nb = total_bands(c->args.height);
    c->local_bands = (nb + c->shard_world - 1) / c->shard_world;   // same on every rank (gather-friendly)
    const size_t n = local_texels(c);
    HIP_TRY(c, hipMalloc(&c->d_seeds, n * 4 * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc(&c->d_fb[0], n * 4 * sizeof(float)));
    HIP_TRY(c, hipMalloc(&c->d_fb[1], n * 4 * sizeof(float)));
    HIP_TRY(c, hipMemsetAsync(c->d_fb[0], 0, n * 4 * sizeof(float), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_fb[1], 0, n * 4 * sizeof(float), c->stream));
    c->tiles_x = (c->args.width + mrt::kTileW - 1) / mrt::kTileW;
    c->n_tiles = c->tiles_x * c->local_bands;
    for (auto& S : c->slot) {
        HIP_TRY(c, hipMalloc(&S.d_tile_cost, (size_t)(c->n_tiles ? c->n_tiles : 1) * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&S.d_tile_order, (size_t)(c->n_tiles ? c->n_tiles : 1) * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&S.d_sort_scratch, (1024 + 16) * sizeof(uint32_t)));
        HIP_TRY(c, hipMemsetAsync(S.d_sort_scratch, 0, (1024 + 16) * sizeof(uint32_t), c->stream));   // [1024] = the tile queue's counter
        S.pix_acc_layers = 0; S.cost_first_layer = 0; S.cost_layers = 1;
        S.cost_valid = false;
        if (&S - c->slot >= 2) continue;        // further slots (pixel-starved shards only) get their colour sums on first use
        HIP_TRY(c, hipMalloc(&S.d_pix_acc, (n ? n : 1) * 16));
        HIP_TRY(c, hipMemsetAsync(S.d_pix_acc, 0, (n ? n : 1) * 16, c->stream));
        S.pix_acc_layers = 1;
    }
    c->frame_slots = 2;
    c->width.div = 0;
    c->inputs_dirty = true;
    // as many persistent single-wave workgroups as the chip holds
    {
        hipDeviceProp_t prop;
        HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
        int wpc = 0;
        if (mrt::render_waves_per_cu(&wpc) != 0 || wpc <= 0) wpc = 16;
        if (c->waves_per_cu_override > 0) wpc = c->waves_per_cu_override;
        c->n_waves = (uint32_t)prop.multiProcessorCount * (uint32_t)wpc;
        c->cus = (uint32_t)prop.multiProcessorCount;
    }
    int e = mrt::launch_fill_seeds(c->d_seeds, c->seed, c->args.width, c->args.height, c->shard_rank,
                                   c->shard_world, c->local_bands, c->stream);
    if (e) return fail(c, MRT_ERR_HIP, "fill_seeds launch failed: %s", hipGetErrorString((hipError_t)e));
    c->target = 0;
    return MRT_OK;
}

void reset_locals(mrt_ctx* c) {
    // lib.rs:419-426: initial Locals
    std::memset(&c->locals, 0, sizeof c->locals);
    c->locals.shape[0] = c->args.width;
    c->locals.shape[1] = c->args.height;
    c->locals.samples_per_frame = c->args.samples_per_frame;
    c->locals.ray_depth = c->args.ray_depth;
    c->locals.framebuffer_weight = 0.0f;
    c->frames_done = 0;
}

uint64_t splitmix64_at(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1u) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

bool finite_in_range(float v, float lim) { return std::isfinite(v) && std::fabs(v) <= lim; }

// Clusters.  The kernel's sweep does not need the spheres themselves, only a conservative "could this
// ray touch it" test, so spatially close spheres are tested in CLUSTERS of up to kClusterK through one
// bounding sphere and the per-sphere discriminants are evaluated only for the members of the few clusters
// that pass.  A ray's expected number of candidates is proportional to the sum of the bounds' cross
// sections, so the grouping minimises sum(R^2): spheres are split kd-tree fashion (widest axis of the
// centres, at a multiple of kClusterK near the median) down to groups of <= 8, and such a group is cut
// into 4 + rest by trying every choice.  Spheres far larger than the median (a ground sphere) stay alone;
// factor == 0 (diagnostic) gives every sphere a cluster of its own.  Consecutive clusters are kd siblings, which is what
// the upper levels (build_hierarchy) group.  R is 1.5 % above the enclosing radius measured from the
// f32-rounded centre: part of the conservativeness argument in DESIGN.md §4.  Clusters are padded to
// kClusterK members and the list to a multiple of kGroup with never-hit records (-r^2 = +inf gives a
// discriminant of -inf).
void build_clusters(const float* centers4, const float* radii, uint32_t n, float factor,
                    std::vector<mrt::SphereRec>& clusters, std::vector<mrt::SphereRec>& members,
                    std::vector<uint32_t>& member_index, std::vector<uint32_t>& direct) {
    clusters.clear(); members.clear(); member_index.clear(); direct.clear();
    const mrt::SphereRec never{0.0f, 0.0f, 0.0f, INFINITY};
    std::vector<double> rs(n);
    for (uint32_t i = 0; i < n; i++) rs[i] = std::fabs((double)radii[i]);
    double big = 1e300;
    if (n > 1) {
        std::vector<double> sorted = rs;
        std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
        big = 8.0 * sorted[n / 2];
    }
    // enclosing sphere of a set: centre of the members' common bounding box, R = max(|c_m - centre| + r_m)
    auto enclose = [&](const uint32_t* idx, uint32_t cnt, double ctr[3]) -> double {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (uint32_t m = 0; m < cnt; m++)
            for (int k = 0; k < 3; k++) {
                lo[k] = std::min(lo[k], (double)centers4[4 * idx[m] + k] - rs[idx[m]]);
                hi[k] = std::max(hi[k], (double)centers4[4 * idx[m] + k] + rs[idx[m]]);
            }
        // the record stores the centre as f32: measure R from the ROUNDED centre so that it stays an enclosure
        for (int k = 0; k < 3; k++) ctr[k] = (double)(float)(0.5 * (lo[k] + hi[k]));
        double R = 0;
        for (uint32_t m = 0; m < cnt; m++) {
            double d2 = 0;
            for (int k = 0; k < 3; k++) { const double d = (double)centers4[4 * idx[m] + k] - ctr[k]; d2 += d * d; }
            R = std::max(R, std::sqrt(d2) + rs[idx[m]]);
        }
        return R;
    };
    std::vector<std::vector<uint32_t>> groups;
    std::vector<uint32_t> pool;                      // spheres that may share a cluster
    std::vector<uint32_t> alone;
    for (uint32_t i = 0; i < n; i++) (factor > 0.0f && rs[i] <= big ? pool : alone).push_back(i);
    // iterative kd split of pool[lo, hi)
    std::vector<std::pair<uint32_t, uint32_t>> stack;
    if (!pool.empty()) stack.push_back({0u, (uint32_t)pool.size()});
    std::vector<std::pair<uint32_t, uint32_t>> leaves;      // in kd order
    while (!stack.empty()) {
        const auto [lo, hi] = stack.back();
        stack.pop_back();
        const uint32_t m = hi - lo;
        if (m <= 2 * mrt::kClusterK) { leaves.push_back({lo, hi}); continue; }
        double bl[3] = {1e300, 1e300, 1e300}, bh[3] = {-1e300, -1e300, -1e300};
        for (uint32_t q = lo; q < hi; q++)
            for (int k = 0; k < 3; k++) {
                bl[k] = std::min(bl[k], (double)centers4[4 * pool[q] + k]);
                bh[k] = std::max(bh[k], (double)centers4[4 * pool[q] + k]);
            }
        int ax = 0;
        for (int k = 1; k < 3; k++) if (bh[k] - bl[k] > bh[ax] - bl[ax]) ax = k;
        std::stable_sort(pool.begin() + lo, pool.begin() + hi, [&](uint32_t x, uint32_t y) {
            const float cx = centers4[4 * x + ax], cy = centers4[4 * y + ax];
            return cx < cy || (cx == cy && x < y);
        });
        uint32_t h = (m / 2 + mrt::kClusterK - 1) / mrt::kClusterK * mrt::kClusterK;
        if (h >= m) h = m - mrt::kClusterK;
        stack.push_back({lo + h, hi});              // popped second: keeps the leaves in left-to-right order
        stack.push_back({lo, lo + h});
    }
    for (const auto& [lo, hi] : leaves) {
        const uint32_t m = hi - lo;
        if (m <= mrt::kClusterK) { groups.emplace_back(pool.begin() + lo, pool.begin() + hi); continue; }
        // 5..8 spheres: the first one plus the 3 others that minimise R_A^2 + R_B^2
        uint32_t bestmask = 0;
        double best = 1e300;
        for (uint32_t mask = 0; mask < (1u << m); mask++) {
            if (!(mask & 1u) || __builtin_popcount(mask) != (int)mrt::kClusterK) continue;
            uint32_t A[8], B[8], na = 0, nb = 0;
            for (uint32_t q = 0; q < m; q++) ((mask >> q) & 1u ? A[na++] : B[nb++]) = pool[lo + q];
            double ctr[3];
            const double ra = enclose(A, na, ctr), rb = enclose(B, nb, ctr);
            if (ra * ra + rb * rb < best) { best = ra * ra + rb * rb; bestmask = mask; }
        }
        std::vector<uint32_t> A, B;
        for (uint32_t q = 0; q < m; q++) ((bestmask >> q) & 1u ? A : B).push_back(pool[lo + q]);
        groups.push_back(A);
        groups.push_back(B);
    }
    // Refinement: swap one member between two clusters whose bounds overlap (or move one into a cluster
    // with a free slot) whenever that lowers R_a^2 + R_b^2, until nothing improves (C3: sum R^2 185 -> 175).
    // Every pair for up to 4,096 clusters, otherwise the 32 following clusters in kd order.
    {
        const size_t ng = groups.size();
        std::vector<double> gr(ng);
        std::vector<std::array<double, 3>> gc(ng);
        auto refresh = [&](size_t g) { double c3[3]; gr[g] = enclose(groups[g].data(), (uint32_t)groups[g].size(), c3); gc[g] = {c3[0], c3[1], c3[2]}; };
        for (size_t g = 0; g < ng; g++) refresh(g);
        const size_t window = ng <= 4096 ? ng : 32;
        for (int pass = 0; pass < (ng <= 4096 ? 4 : 2); pass++) {
            size_t improved = 0;
            for (size_t a = 0; a < ng; a++) {
                for (size_t b = a + 1; b < ng && b <= a + window; b++) {
                    double d2 = 0;
                    for (int k = 0; k < 3; k++) d2 += (gc[a][k] - gc[b][k]) * (gc[a][k] - gc[b][k]);
                    if (d2 > (gr[a] + gr[b]) * (gr[a] + gr[b])) continue;
                    const double base = gr[a] * gr[a] + gr[b] * gr[b];
                    double best = base - 1e-12 * base;
                    std::vector<uint32_t> bestA, bestB;
                    std::vector<uint32_t> A, B;
                    double c3[3];
                    auto consider = [&]() {
                        const double ra = enclose(A.data(), (uint32_t)A.size(), c3), rb = enclose(B.data(), (uint32_t)B.size(), c3);
                        if (ra * ra + rb * rb < best) { best = ra * ra + rb * rb; bestA = A; bestB = B; }
                    };
                    for (size_t i = 0; i < groups[a].size(); i++)
                        for (size_t j = 0; j < groups[b].size(); j++) {
                            A = groups[a]; B = groups[b];
                            std::swap(A[i], B[j]);
                            consider();
                        }
                    if (groups[b].size() < mrt::kClusterK && groups[a].size() > 1)
                        for (size_t i = 0; i < groups[a].size(); i++) {
                            A = groups[a]; B = groups[b];
                            B.push_back(A[i]); A.erase(A.begin() + (long)i);
                            consider();
                        }
                    if (groups[a].size() < mrt::kClusterK && groups[b].size() > 1)
                        for (size_t j = 0; j < groups[b].size(); j++) {
                            A = groups[a]; B = groups[b];
                            A.push_back(B[j]); B.erase(B.begin() + (long)j);
                            consider();
                        }
                    if (!bestA.empty()) {
                        groups[a] = bestA; groups[b] = bestB;
                        refresh(a); refresh(b);
                        improved++;
                    }
                }
            }
            if (!improved) break;
        }
    }
    // the largest of the big spheres are tested by every ray directly (KParams::direct); the others get a
    // cluster of their own
    std::stable_sort(alone.begin(), alone.end(), [&](uint32_t x, uint32_t y) { return rs[x] > rs[y]; });
    for (uint32_t q = 0; q < alone.size(); q++) {
        if (factor > 0.0f && q < mrt::kMaxDirect && rs[alone[q]] > big) direct.push_back(alone[q]);
        else groups.push_back({alone[q]});
    }
    for (auto& g : groups) {
        std::sort(g.begin(), g.end());
        double ctr[3];
        const double R = enclose(g.data(), (uint32_t)g.size(), ctr);
        const float Rf = (float)(R * mrt::kBoundInflate) + 1e-30f;     // rounding R to f32 moves it by 6e-8 R, the 1.5 % is for the proof
        clusters.push_back(mrt::SphereRec{(float)ctr[0], (float)ctr[1], (float)ctr[2], -(Rf * Rf)});
        for (uint32_t m = 0; m < mrt::kClusterK; m++) {
            if (m < g.size()) {
                const float r = radii[g[m]];
                members.push_back(mrt::SphereRec{centers4[4 * g[m]], centers4[4 * g[m] + 1], centers4[4 * g[m] + 2], -(r * r)});
                member_index.push_back(g[m]);
            } else {
                members.push_back(never);
                member_index.push_back(0u);
            }
        }
    }
    while (clusters.empty() || clusters.size() % mrt::kGroup != 0) {
        clusters.push_back(never);                                    // S = -inf: never a candidate
        for (uint32_t m = 0; m < mrt::kClusterK; m++) { members.push_back(never); member_index.push_back(0u); }
    }
}

// Upper levels of the hierarchy: level k+1 bounds 4 consecutive level-k nodes (consecutive in kd order,
// so neighbours in space); its bounding sphere is measured from the MEMBER spheres under
// it, R = kBoundInflate x the enclosing radius from the f32-rounded centre, so the conservativeness argument of
// the clusters holds for every level.  Levels are added while the top has more than
// top_target records (the sweep costs every ray one test per top record; a walk round costs about 1.5
// wave-instructions per item).  Every level is padded to a multiple of 4 (the top: kGroup) with
// never-hit records; the children of a never-hit node are never read.
struct Hierarchy {
    std::vector<mrt::SphereRec> top, nodes;
    std::vector<uint32_t> member_index;
    std::vector<mrt::BoxFull> boxes;          // levels 1 .. levels (the top last), level k at box_base[k]
    uint32_t box_base[mrt::kMaxLevels + 1] = {0, 0, 0, 0, 0};
    bool box_quad = false;
    float box_kc = 0.0f;                      // the slack's coefficient of X: one per scene
    uint32_t levels = 1, n_members = 0;       // n_members: level 0 including the direct spheres
    uint32_t level_base[mrt::kMaxLevels] = {0, 0, 0, 0};
    uint32_t n_direct = 0, direct_first = 0;
    mrt::SphereRec direct[mrt::kMaxDirect] = {};
    uint32_t direct_index[mrt::kMaxDirect] = {};
};

// The boxes of every node (levels 1 .. top), for the walk of large scenes (kernels.hip, box_may_touch).  Node j of level k
// covers the members [j 4^k, (j+1) 4^k) of the hierarchy part of level 0.  The test is "the LINE of the ray passes the box
// grown by K on every side", three separating axes d x e_i; it must hold whenever the reference's discriminant of a member
// under the node is computed >= 0, i.e. whenever the line passes within h of the member's centre,
// h^2 <= r^2 + E, E = 14 eps |oc|^2 / a: h - r <= E / (2 r) (the quadratic form) and <= sqrt(E) (the linear form).  With
// |oc| <= |p| + |e| (p: origin - box centre, e: half extents) and |d_j| + |d_k| <= 1.4143:
//     quadratic   K = kc |p|^2 + kpad,  kc = 1.3e-6 / r_min,  kpad = kc |e|^2 + 4.4e-14 / kc
//     linear      K = kc |p|_1 + kpad,  kc = 1.5e-3,          kpad = kc |e|_1
// (each with >= 9 % to spare over 1.4143 x the bound; the 4.4e-14 / kc makes the quadratic form cover the test's own
// rounding, 4 eps |p|_1, by the inequality of the means).  The quadratic form is far smaller at moderate distances, the
// linear one at large distances from tiny spheres; the scene takes the one that is smaller at its own reach.
// kc is ONE value per scene (round 5; r_min = the scene's smallest radius, which only makes K larger for the other boxes): the
// kernel takes it from its arguments, and kpad -- the only other per-box part of K -- is folded into the extents the kernel
// reads (pack_boxes), so a box is 24 bytes on the device.
void build_boxes(const float* centers4, const float* radii, const std::vector<mrt::SphereRec>& members, Hierarchy& H) {
    const mrt::BoxFull never_box{0.0f, 0.0f, 0.0f, -3.0e38f, -3.0e38f, -3.0e38f, 0.0f, 0.0f};
    H.boxes.clear();
    // the scene's reach and median radius decide the form of the slack
    double lo_all[3] = {1e300, 1e300, 1e300}, hi_all[3] = {-1e300, -1e300, -1e300};
    std::vector<double> rr;
    for (size_t m = 0; m < members.size(); m++) {
        if (!std::isfinite(members[m].neg_r2)) continue;
        const uint32_t i = H.member_index[m];
        rr.push_back(std::fabs((double)radii[i]));
        for (int k = 0; k < 3; k++) {
            lo_all[k] = std::min(lo_all[k], (double)centers4[4 * i + k]);
            hi_all[k] = std::max(hi_all[k], (double)centers4[4 * i + k]);
        }
    }
    double reach = 0.0, r_small = 1e300;
    if (!rr.empty()) {
        for (int k = 0; k < 3; k++) reach += (hi_all[k] - lo_all[k]) * (hi_all[k] - lo_all[k]);
        reach = std::sqrt(reach);
        r_small = *std::min_element(rr.begin(), rr.end());
    }
    // (by the scene's SMALLEST radius, since kc is one value per scene: a scene with a few tiny spheres takes the linear form)
    H.box_quad = 1.3e-6 / std::max(r_small, 1e-300) * reach < 1.5e-3 * 2.0;      // quadratic slack at the reach < 2 x the linear one
    const double kc_scene = H.box_quad ? 1.3e-6 / std::max(r_small, 1e-30) : 1.5e-3;
    auto up = [](double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, INFINITY); return f; };
    for (uint32_t k = 1; k <= H.levels; k++) {
        H.box_base[k] = (uint32_t)H.boxes.size();
        const size_t n_k = k == H.levels ? H.top.size() : (size_t)((k + 1 < H.levels ? H.level_base[k + 1] : (uint32_t)H.nodes.size()) - H.level_base[k]);
        const size_t span = (size_t)1 << (2 * k);
        for (size_t j = 0; j < n_k; j++) {
            const size_t m0 = j * span, m1 = std::min(members.size(), (j + 1) * span);
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            bool any = false;
            for (size_t m = m0; m < m1; m++) {
                if (!std::isfinite(members[m].neg_r2)) continue;
                const uint32_t i = H.member_index[m];
                const double r = std::fabs((double)radii[i]);
                for (int q = 0; q < 3; q++) {
                    lo[q] = std::min(lo[q], (double)centers4[4 * i + q] - r);
                    hi[q] = std::max(hi[q], (double)centers4[4 * i + q] + r);
                }
                any = true;
            }
            if (!any) { H.boxes.push_back(never_box); continue; }
            mrt::BoxFull b;
            float c[3], e[3];
            double e1 = 0.0, e2 = 0.0;
            for (int q = 0; q < 3; q++) {
                c[q] = (float)(0.5 * (lo[q] + hi[q]));
                const double ext = std::max(hi[q] - (double)c[q], (double)c[q] - lo[q]) * (1.0 + 1e-6) + 1e-37;   // (the 1e-6: the three roundings of the test's right-hand side)
                e[q] = up(ext);
                e1 += (double)e[q];
                e2 += (double)e[q] * (double)e[q];
            }
            b.cx = c[0]; b.cy = c[1]; b.cz = c[2]; b.ex = e[0]; b.ey = e[1]; b.ez = e[2];
            b.kc = H.box_quad ? up(kc_scene) : 1.5e-3f;
            b.kpad = H.box_quad ? up((double)b.kc * e2 + 4.4e-14 / kc_scene) : up(1.5e-3 * e1);
            H.boxes.push_back(b);
        }
    }
    H.box_kc = H.box_quad ? up(kc_scene) : 1.5e-3f;
}

// What the kernel reads of a box (mrt_internal.h, BoxRec): the centre and the extents with kpad folded in, e' = e + kpad rounded
// up.  The test on the axis d x e_i then has the slack kc X + kpad (|d_j| + |d_k|) instead of kc X + kpad; what is needed there is
// rho |d x e_i| + (the test's rounding) (|d_j| + |d_k|), rho the distance beyond the box the line of a candidate can pass, and
// |d x e_i| <= s = |d_j| + |d_k| <= 1.4143: both sides are linear in s on [0, 1] and on [1, 1.4143], at s = 0 the left side is
// kc X >= 0, and at s = 1 and s = 1.4143 the inequality is the one build_boxes provides (kc X + kpad >= 1.4143 rho + the
// rounding: tests/test_hierarchy_host.py checks it box by box).
void pack_boxes(const std::vector<mrt::BoxFull>& full, std::vector<mrt::BoxRec>& out) {
    auto up = [](double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, INFINITY); return f; };
    out.resize(full.size());
    for (size_t i = 0; i < full.size(); i++) {
        const mrt::BoxFull& b = full[i];
        const bool real = b.ex >= 0.0f && b.ex < 1.0e37f;           // (never-hit: -3e38; opened wide: 3e37)
        out[i] = mrt::BoxRec{b.cx, b.cy, b.cz, real ? up((double)b.ex + (double)b.kpad) : b.ex, real ? up((double)b.ey + (double)b.kpad) : b.ey,
                             real ? up((double)b.ez + (double)b.kpad) : b.ez};
    }
}

// The boxes in the order the kernel walks them (KParams::boxes): depth t of the hierarchy (0 = the swept top = level
// `levels`, levels - 1 = the clusters = level 1) at o_t = n_top (4^t - 1) / 3, n_top = the padded top: the children of node g
// are 4 g + n_top .. + 3 whatever its depth, so a work item needs no level.  Slots without a node hold never-hit boxes.
// `open`: every real box opened wide (extents 3e37: the test never rejects) -- the A/B form of mrt_debug_set_boxes(0).
void boxes_top_down(const Hierarchy& H, bool open, std::vector<mrt::BoxFull>& out, uint32_t* cluster_first, uint32_t* cluster_parent_first) {
    const mrt::BoxFull never_box{0.0f, 0.0f, 0.0f, -3.0e38f, -3.0e38f, -3.0e38f, 0.0f, 0.0f};
    const size_t n_top = H.top.size();
    size_t o[mrt::kMaxLevels + 1];
    o[0] = 0;
    for (uint32_t t = 0; t < H.levels; t++) o[t + 1] = o[t] + (n_top << (2 * t));
    out.assign(o[H.levels], never_box);
    for (uint32_t t = 0; t < H.levels; t++) {
        const uint32_t k = H.levels - t;                 // the level at this depth
        const size_t first = H.box_base[k], last = k < H.levels ? H.box_base[k + 1] : H.boxes.size();
        for (size_t j = 0; j < last - first && j < (n_top << (2 * t)); j++) {
            mrt::BoxFull b = H.boxes[first + j];
            if (open && b.ex >= 0.0f) b.ex = b.ey = b.ez = 3.0e37f;
            out[o[t] + j] = b;
        }
    }
    *cluster_first = (uint32_t)o[H.levels - 1];
    *cluster_parent_first = H.levels >= 2 ? (uint32_t)o[H.levels - 2] : 0u;
}

constexpr uint32_t kBoxMinMembers = 4096;     // member slots from which the walk tests boxes by default (fill_scene_params)
void build_hierarchy(const float* centers4, const float* radii, uint32_t n, float factor, uint32_t max_levels,
                     uint32_t top_target, Hierarchy& H) {
    const mrt::SphereRec never{0.0f, 0.0f, 0.0f, INFINITY};
    std::vector<mrt::SphereRec> members, cur;
    std::vector<uint32_t> direct;
    build_clusters(centers4, radii, n, factor, cur, members, H.member_index, direct);
    H.nodes = members;
    // the direct spheres follow the clusters' members in level 0 (no cluster, no bound above them)
    H.n_direct = (uint32_t)direct.size();
    H.direct_first = (uint32_t)members.size();
    for (uint32_t j = 0; j < mrt::kClusterK; j++) {
        mrt::SphereRec rec = never;
        uint32_t idx = 0;
        if (j < direct.size()) {
            idx = direct[j];
            const float r = radii[idx];
            rec = mrt::SphereRec{centers4[4 * idx], centers4[4 * idx + 1], centers4[4 * idx + 2], -(r * r)};
        }
        H.direct[j] = rec;
        H.direct_index[j] = idx;
        if (!direct.empty()) { H.nodes.push_back(rec); H.member_index.push_back(idx); }
    }
    static_assert(mrt::kMaxDirect == mrt::kClusterK, "level 0 stays a multiple of kClusterK");
    H.n_members = (uint32_t)H.nodes.size();
    H.levels = 1;
    H.level_base[0] = 0;
    // scenes whose members fit 10-bit ids (the kernel's SMALL variant) keep one level: with <= 256 clusters
    // the sweep is cheap and the bounds of 16 spheres are loose (C3: a ray touches 10 of 38 such bounds)
    if (H.n_members <= 1024u) max_levels = 1;      // (same test as scene_is_small() in kernels.hip)
    // top_target 0 = automatic: levels are added while the top has more than 256 records -- 128 where the walk tests boxes
    // below the top, which make a smaller top cheaper (round 3: 4,901 spheres 34.1 -> 33.3 ms per 64-spp frame, 10,001 spheres
    // 48.2 -> 47.3; without boxes 1,297 / 2,501 spheres lose 20 % with a top of <= 64)
    if (top_target == 0) top_target = H.n_members > kBoxMinMembers ? 128u : 256u;
    while (H.levels < max_levels && cur.size() > top_target) {
        const size_t span = (size_t)1 << (2 * (H.levels + 1));        // members under one node of the new level
        const size_t n_par = (cur.size() + 3) / 4;
        std::vector<mrt::SphereRec> par;
        par.reserve(n_par + mrt::kGroup);
        for (size_t j = 0; j < n_par; j++) {
            const size_t m0 = j * span, m1 = std::min(members.size(), (j + 1) * span);
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            bool any = false;
            for (size_t m = m0; m < m1; m++) {
                if (!std::isfinite(members[m].neg_r2)) continue;
                const uint32_t i = H.member_index[m];
                const double r = std::fabs((double)radii[i]);
                for (int k = 0; k < 3; k++) {
                    lo[k] = std::min(lo[k], (double)centers4[4 * i + k] - r);
                    hi[k] = std::max(hi[k], (double)centers4[4 * i + k] + r);
                }
                any = true;
            }
            if (!any) { par.push_back(never); continue; }
            double ctr[3], R = 0;
            for (int k = 0; k < 3; k++) ctr[k] = (double)(float)(0.5 * (lo[k] + hi[k]));
            for (size_t m = m0; m < m1; m++) {
                if (!std::isfinite(members[m].neg_r2)) continue;
                const uint32_t i = H.member_index[m];
                double d2 = 0;
                for (int k = 0; k < 3; k++) { const double d = (double)centers4[4 * i + k] - ctr[k]; d2 += d * d; }
                R = std::max(R, std::sqrt(d2) + std::fabs((double)radii[i]));
            }
            const float Rf = (float)(R * mrt::kBoundInflate) + 1e-30f;
            par.push_back(mrt::SphereRec{(float)ctr[0], (float)ctr[1], (float)ctr[2], -(Rf * Rf)});
        }
        while (cur.size() % 4 != 0) cur.push_back(never);
        H.level_base[H.levels] = (uint32_t)H.nodes.size();
        H.nodes.insert(H.nodes.end(), cur.begin(), cur.end());
        cur.swap(par);
        H.levels++;
    }
    while (cur.empty() || cur.size() % 32 != 0) cur.push_back(never);      // 32 = one tile of the matrix-core sweep
    H.top.swap(cur);
    build_boxes(centers4, radii, members, H);
}

// matrix-core sweep or SGPR-fed VALU sweep for the next launch: forced by mrt_debug_set_sweep,
// else the scene's verdict (mrt_set_world_raw) and the same test on the camera's distance from the origin
bool use_matrix_core_sweep(const mrt_ctx* c);

// The top level once more, as the A operand of the matrix-core sweep (kernels.hip, mfma_sweep_tile): per
// tile of 32 records 64 lanes x 8 bf16, lane l = row (l & 31), k = 8 (l >> 5) + j:
//     k 0..2 C_hi, 3..5 C_hi, 6..8 C_lo, 9..11 (1,1,1), 12..14 Ck (hi, mid, lo), 15: 0
// where row m of tile t is record 32 t + 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3) -- the order in which the
// MFMA result registers come out, so that the two 16-bit sign words per tile are the masks of chunks 2t and
// 2t + 1.  Ck = C.C - R^2 - 2^-13 (C.C + R^2): the record's share of the slack that covers what the bf16
// split drops.  A never-hit record gets Ck = 3e38 (finite: an infinity would turn the other
// GEMM's 0 x Ck into NaN).  Also returns what set_world needs to decide whether the slack is negligible:
// the largest C.C and the median R^2.
constexpr double kMfmaSlack = 0x1p-13;
uint16_t bf16_rne(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);      // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
float bf16_value(uint16_t h) { const uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
void build_top_mfma(const std::vector<mrt::SphereRec>& top, std::vector<uint16_t>& out, float origin[3], double* max_c2,
                    double* med_r2, size_t* n_real) {
    const size_t tiles = top.size() / 32;
    out.assign(tiles * 512, 0);
    std::vector<double> r2s;
    *max_c2 = 0.0;
    // the GEMMs run in coordinates relative to the centre of the records' bounding box (the slack grows with the
    // squared distances from THAT point, wherever the scene sits); the kernel subtracts it from the ray origin
    double lo[3] = {1e300, 1e300, 1e300}, hi3[3] = {-1e300, -1e300, -1e300};
    for (const auto& r : top) {
        if (!std::isfinite(r.neg_r2)) continue;
        const double c[3] = {r.cx, r.cy, r.cz};
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], c[k]); hi3[k] = std::max(hi3[k], c[k]); }
    }
    for (int k = 0; k < 3; k++) origin[k] = lo[k] <= hi3[k] ? (float)(0.5 * (lo[k] + hi3[k])) : 0.0f;
    const uint16_t one = bf16_rne(1.0f);
    for (size_t t = 0; t < tiles; t++)
        for (uint32_t m = 0; m < 32; m++) {
            const mrt::SphereRec& r = top[32 * t + 16 * ((m >> 2) & 1u) + 4 * (m >> 3) + (m & 3u)];
            float ck = 3.0e38f;
            // centre relative to the origin: exact in double, then rounded to f32 -- the rounding moves the bound by
            // at most 2 eps |c|, which its radius absorbs
            const float c[3] = {(float)((double)r.cx - origin[0]), (float)((double)r.cy - origin[1]), (float)((double)r.cz - origin[2])};
            if (std::isfinite(r.neg_r2)) {
                const double c2 = (double)c[0] * c[0] + (double)c[1] * c[1] + (double)c[2] * c[2];
                const double R = std::sqrt(-(double)r.neg_r2) + 2.0 * 0x1p-24 * std::sqrt(c2), R2 = R * R;
                const double v = c2 - R2 - kMfmaSlack * (c2 + R2);
                ck = (float)v;
                if ((double)ck > v) ck = std::nextafterf(ck, -INFINITY);
                *max_c2 = std::max(*max_c2, c2);
                r2s.push_back(R2);
            }
            uint16_t hi[3], lo16[3];
            for (int k = 0; k < 3; k++) { hi[k] = bf16_rne(c[k]); lo16[k] = bf16_rne(c[k] - bf16_value(hi[k])); }
            const uint16_t k0 = bf16_rne(ck);
            const float ck1 = ck - bf16_value(k0);
            const uint16_t k1 = bf16_rne(ck1), k2 = bf16_rne(ck1 - bf16_value(k1));
            const uint16_t kvals[16] = {hi[0], hi[1], hi[2], hi[0], hi[1], hi[2], lo16[0], lo16[1], lo16[2], one, one, one, k0, k1, k2, 0};
            uint16_t* o = out.data() + t * 512;
            for (int k = 0; k < 16; k++) o[((k >> 3) * 32 + m) * 8 + (k & 7)] = kvals[k];
        }
    *n_real = r2s.size();
    *med_r2 = 0.0;
    if (!r2s.empty()) { std::nth_element(r2s.begin(), r2s.begin() + r2s.size() / 2, r2s.end()); *med_r2 = r2s[r2s.size() / 2]; }
}

bool use_matrix_core_sweep(const mrt_ctx* c) {
    if (c->sweep_mode == 2) return true;
    if (c->sweep_mode == 1 || !c->mfma_scene_ok) return false;
    double o2 = 0.0;                              // squared distance of the camera from the GEMMs' origin
    for (int k = 0; k < 3; k++) {
        const double d = (c->cam_raw.mode ? (double)c->cam_raw.origin[k] : 0.0) - (double)c->mfma_origin[k];
        o2 += d * d;
    }
    return kMfmaSlack * o2 <= 0.1 * c->mfma_r2_ref;
}

// the side stream of a frame slot and its two events
hipError_t create_slot_streams(mrt_ctx::FrameSlot& S) {
    hipError_t e = hipSuccess;
    if (!S.stream) e = hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking);
    if (e == hipSuccess && !S.render_done) e = hipEventCreateWithFlags(&S.render_done, hipEventDisableTiming);
    if (e == hipSuccess && !S.finalize_done) e = hipEventCreateWithFlags(&S.finalize_done, hipEventDisableTiming);
    if (e == hipSuccess && !S.stats_ready) e = hipEventCreateWithFlags(&S.stats_ready, hipEventDisableTiming);
    return e;
}

}  // namespace

namespace mrt {

// Bounded host waits.  Every wait for the GPU in this library goes through these: poll (spin briefly, then sleep in growing
// steps up to 200 us -- a frame is 0.2 ms at its shortest) until the event / stream is complete or the context's deadline has
// passed; then fail with MRT_ERR_STALLED and a message that names the wait, so that a stall is a loud status and never a silent
// hang (round 4's parity campaign lost a 420-s run to one: DESIGN_HISTORY.md, round 5).
template <typename Query>
static int bounded_wait(mrt_ctx* c, Query query, const char* what) {
    if (c->stalled) return fail(c, MRT_ERR_STALLED, "%s: the context has stalled before (destroy it)", what);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = query();
        if (q == hipSuccess) return MRT_OK;
        if (q != hipErrorNotReady) return fail(c, MRT_ERR_HIP, "%s: %s", what, hipGetErrorString(q));
        (void)hipGetLastError();                                        // (hipErrorNotReady is not an error)
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (c->wait_timeout_s > 0.0 && waited > c->wait_timeout_s) {
            c->stalled = true;
            return fail(c, MRT_ERR_STALLED, "stalled in %s: not complete after %.1f s (mrt_set_wait_timeout)", what, waited);
        }
        if (waited < 50e-6) continue;                                   // spin
        std::this_thread::sleep_for(std::chrono::microseconds(waited < 2e-3 ? 20 : waited < 50e-3 ? 100 : 200));
    }
}
int wait_event(mrt_ctx* c, hipEvent_t ev, const char* what) { return bounded_wait(c, [&]() { return hipEventQuery(ev); }, what); }
int wait_stream(mrt_ctx* c, hipStream_t s, const char* what) { return bounded_wait(c, [&]() { return hipStreamQuery(s); }, what); }
// everything this context has in flight (the side streams, then the caller's stream)
int wait_all(mrt_ctx* c, const char* what) {
    char buf[160];
    for (uint32_t i = 0; i < mrt_ctx::kMaxFrameSlots; i++) {
        mrt_ctx::FrameSlot& S = c->slot[i];
        if (!S.stream) continue;
        std::snprintf(buf, sizeof buf, "%s (side stream of slot %u, last frame %llu, %u frames in flight)", what, i,
                      (unsigned long long)S.render_seq, c->frame_slots);
        MRT_TRY(wait_stream(c, S.stream, buf));
        S.render_pending = false;
    }
    if (c->stream) {
        std::snprintf(buf, sizeof buf, "%s (the context's stream, frame %llu)", what, (unsigned long long)c->frame_seq);
        MRT_TRY(wait_stream(c, c->stream, buf));
    }
    return MRT_OK;
}

}  // namespace mrt

namespace {

}  // namespace

namespace mrt { void set_global_error(const char* msg) { g_err = msg ? msg : ""; } }

extern "C" {

int mrt_abi_version(void) { return MRT_ABI_VERSION; }

const char* mrt_status_string(int s) {
    switch (s) {
        case MRT_OK: return "ok";
        case MRT_ERR_INVALID_ARG: return "invalid argument";
        case MRT_ERR_NO_DEVICE: return "no usable HIP device";
        case MRT_ERR_HIP: return "HIP runtime error";
        case MRT_ERR_NO_SCENE: return "no scene set";
        case MRT_ERR_BAD_SCENE: return "invalid scene data";
        case MRT_ERR_TOO_SMALL: return "buffer too small";
        case MRT_ERR_STATE: return "call not allowed in this state";
        case MRT_ERR_IO: return "i/o error";
        case MRT_ERR_STALLED: return "a wait for the GPU passed its deadline";
        default: return "unknown status";
    }
}

const char* mrt_last_error(mrt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

void mrt_args_default(mrt_args* out) {
    if (!out) return;
    out->width = 0; out->height = 0;
    out->samples_per_frame = 1; out->ray_depth = 50; out->max_framebuffer_weight = 1.0f;
}

void mrt_args_resolve_size(mrt_args* a) {
    if (!a) return;
    if (a->width == 0 && a->height == 0) { a->width = MRT_DEFAULT_WIDTH; a->height = MRT_DEFAULT_HEIGHT; }
    else if (a->width == 0) a->width = a->height;
    else if (a->height == 0) a->height = a->width;
}

float mrt_frame_weight(uint32_t frames_done, float max_w) {
    if (frames_done == 0) return 0.0f;                                 // lib.rs:424
    const float w = (float)frames_done / (float)(frames_done + 1u);    // lib.rs:304
    return max_w < w ? max_w : w;                                      // f32::min, lib.rs:301-304
}

void mrt_pixel_seed(uint64_t seed, uint64_t pixel_index, uint32_t out[4]) {
    const uint64_t a = splitmix64_at(seed, 2u * pixel_index), b = splitmix64_at(seed, 2u * pixel_index + 1u);
    out[0] = (uint32_t)a; out[1] = (uint32_t)(a >> 32); out[2] = (uint32_t)b; out[3] = (uint32_t)(b >> 32);
    if ((out[0] | out[1] | out[2] | out[3]) == 0u) {
        out[0] = 0x9E3779B9u; out[1] = 0x7F4A7C15u; out[2] = 0xBF58476Du; out[3] = 0x1CE4E5B9u;
    }
}

void mrt_frame_shuffle(uint64_t seed, uint32_t frame, uint32_t out[4]) {
    if (frame == 0) { out[0] = out[1] = out[2] = out[3] = 0; return; }  // lib.rs:422
    const uint64_t s2 = seed ^ 0xD1B54A32D192ED03ull;
    const uint64_t a = splitmix64_at(s2, 2ull * frame), b = splitmix64_at(s2, 2ull * frame + 1u);
    out[0] = (uint32_t)a; out[1] = (uint32_t)(a >> 32); out[2] = (uint32_t)b; out[3] = (uint32_t)(b >> 32);
}

int mrt_camera_derive(const mrt_camera* cam, mrt_camera_raw* out) {
    if (!cam || !out) return MRT_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    out->mode = cam->mode;
    if (cam->mode == 0) return MRT_OK;
    if (cam->mode != 1) return MRT_ERR_INVALID_ARG;
    double lf[3], la[3], up[3], w[3], u[3], v[3];
    for (int i = 0; i < 3; i++) { lf[i] = cam->lookfrom[i]; la[i] = cam->lookat[i]; up[i] = cam->vup[i]; }
    for (int i = 0; i < 3; i++) w[i] = lf[i] - la[i];
    const double wl = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int i = 0; i < 3; i++) w[i] /= wl;
    u[0] = up[1] * w[2] - up[2] * w[1];
    u[1] = up[2] * w[0] - up[0] * w[2];
    u[2] = up[0] * w[1] - up[1] * w[0];
    const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int i = 0; i < 3; i++) u[i] /= ul;
    v[0] = w[1] * u[2] - w[2] * u[1];
    v[1] = w[2] * u[0] - w[0] * u[2];
    v[2] = w[0] * u[1] - w[1] * u[0];
    const double deg = 3.14159265358979323846 / 180.0;
    const double focus = cam->focus_dist;
    const double s = std::tan(0.5 * (double)cam->vfov_deg * deg) * focus;
    const double r = std::tan(0.5 * (double)cam->defocus_angle_deg * deg) * focus;
    out->defocus = cam->defocus_angle_deg > 0.0f;
    for (int i = 0; i < 3; i++) {
        out->origin[i] = cam->lookfrom[i];
        out->su[i] = (float)(s * u[i]);
        out->sv[i] = (float)(s * v[i]);
        out->fw[i] = (float)(focus * w[i]);
        out->ru[i] = (float)(r * u[i]);
        out->rv[i] = (float)(r * v[i]);
    }
    for (int i = 0; i < 3; i++)
        if (!std::isfinite(out->su[i]) || !std::isfinite(out->sv[i]) || !std::isfinite(out->fw[i]) ||
            !std::isfinite(out->ru[i]) || !std::isfinite(out->rv[i]) || !std::isfinite(out->origin[i]))
            return MRT_ERR_INVALID_ARG;
    return MRT_OK;
}

// lib.rs:722-799
int mrt_pack_world(const mrt_sphere* sp, size_t n, mrt_world* world,
                   float* vec4, size_t cap_vec4, size_t* n_vec4,
                   float* f32, size_t cap_f32, size_t* n_f32,
                   int32_t* i32, size_t cap_i32, size_t* n_i32) {
    if ((!sp && n) || !world || !vec4 || !f32 || !i32 || !n_vec4 || !n_f32 || !n_i32) return MRT_ERR_INVALID_ARG;
    if (n > (size_t)INT32_MAX / 4) return MRT_ERR_INVALID_ARG;
    size_t nl = 0, nm = 0, nd = 0;
    for (size_t i = 0; i < n; i++) {
        switch (sp[i].material_ty) {
            case MRT_LAMBERTIAN: nl++; break;
            case MRT_METAL: nm++; break;
            case MRT_DIELECTRIC: nd++; break;
            default: return MRT_ERR_BAD_SCENE;
        }
    }
    if (cap_vec4 < n + nl + nm || cap_f32 < n + nm + nd || cap_i32 < 2 * n) return MRT_ERR_TOO_SMALL;
    std::memset(world, 0, sizeof *world);
    size_t v = 0, f = 0, k = 0;
    auto push4 = [&](const float* p) { vec4[4 * v] = p[0]; vec4[4 * v + 1] = p[1]; vec4[4 * v + 2] = p[2]; vec4[4 * v + 3] = 1.0f; v++; };
    world->spheres.center_base_idx = (int32_t)v;
    for (size_t i = 0; i < n; i++) push4(sp[i].center);
    world->spheres.radius_base_idx = (int32_t)f;
    for (size_t i = 0; i < n; i++) f32[f++] = sp[i].radius;
    world->spheres.material_ty_base_idx = (int32_t)k;
    for (size_t i = 0; i < n; i++) i32[k++] = sp[i].material_ty;
    world->spheres.material_idx_base_idx = (int32_t)k;
    {
        int32_t cl = 0, cm = 0, cd = 0;
        for (size_t i = 0; i < n; i++)
            i32[k++] = sp[i].material_ty == MRT_LAMBERTIAN ? cl++ : sp[i].material_ty == MRT_METAL ? cm++ : cd++;
    }
    world->spheres.length = (int32_t)n;
    world->lambertians.albedo_base_idx = (int32_t)v;
    for (size_t i = 0; i < n; i++) if (sp[i].material_ty == MRT_LAMBERTIAN) push4(sp[i].albedo);
    world->lambertians.length = (int32_t)nl;
    world->metals.albedo_base_idx = (int32_t)v;
    for (size_t i = 0; i < n; i++) if (sp[i].material_ty == MRT_METAL) push4(sp[i].albedo);
    world->metals.fuzz_base_idx = (int32_t)f;
    for (size_t i = 0; i < n; i++) if (sp[i].material_ty == MRT_METAL) f32[f++] = sp[i].param;
    world->metals.length = (int32_t)nm;
    world->dielectrics.ior_base_idx = (int32_t)f;
    for (size_t i = 0; i < n; i++) if (sp[i].material_ty == MRT_DIELECTRIC) f32[f++] = sp[i].param;
    world->dielectrics.length = (int32_t)nd;
    *n_vec4 = v; *n_f32 = f; *n_i32 = k;
    return MRT_OK;
}

int mrt_create(const mrt_args* args, uint64_t seed, int device, mrt_ctx** out) {
    if (!args || !out) return fail(nullptr, MRT_ERR_INVALID_ARG, "mrt_create: null argument");
    *out = nullptr;
    mrt_args a = *args;
    mrt_args_resolve_size(&a);
    if (a.width > (1u << 20) || a.height > (1u << 20))
        return fail(nullptr, MRT_ERR_INVALID_ARG, "mrt_create: image %ux%u too large", a.width, a.height);
    // (The frames in flight of a pixel-starved shard -- up to 8, each on a side stream of its own -- only run side by side on
    // hardware queues of their own, and HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES of them, 4 by default.  That
    // variable is the HOST's to set, before its first HIP call: INTEGRATION.md 2a; the Python package and bench.py do.  This
    // library does not touch the environment: it measures how many of its streams really run at a time when it first wants
    // more than two frames in flight -- probe_stream_concurrency -- and holds the schedule to that.)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, MRT_ERR_NO_DEVICE, "mrt_create: no HIP device (this backend has no CPU fallback)");
    if (device < 0 || device >= ndev)
        return fail(nullptr, MRT_ERR_NO_DEVICE, "mrt_create: device %d out of range (%d present)", device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return fail(nullptr, MRT_ERR_NO_DEVICE, "mrt_create: cannot query device %d", device);
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MRT_ERR_NO_DEVICE, "mrt_create: device %d is %s; kernels are built for gfx950 only",
                    device, prop.gcnArchName);
    mrt_ctx* c = new (std::nothrow) mrt_ctx();
    if (!c) return fail(nullptr, MRT_ERR_INVALID_ARG, "mrt_create: out of host memory");
    c->device = device; c->args = a; c->seed = seed;
    if (const char* t = std::getenv("MRT_WAIT_TIMEOUT_S")) {
        char* end = nullptr;
        const double v = std::strtod(t, &end);
        if (end != t && v >= 0.0 && std::isfinite(v)) c->wait_timeout_s = v;
    }
    c->cam_raw.mode = 0;
    reset_locals(c);
    auto bail = [&](int st) { g_err = c->err; mrt_destroy(c); return st; };
    if (hipSetDevice(device) != hipSuccess) { c->err = "hipSetDevice failed"; return bail(MRT_ERR_HIP); }
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { c->err = "hipStreamCreate failed"; return bail(MRT_ERR_HIP); }
    c->stream = c->own_stream;
    // (slots 0 and 1 now; the further ones -- pixel-starved shards only -- when redraw_frames first needs them)
    for (uint32_t i = 0; i < 2; i++)
        if (create_slot_streams(c->slot[i]) != hipSuccess) { c->err = "side stream creation failed"; return bail(MRT_ERR_HIP); }
    if (hipEventCreateWithFlags(&c->ev_inputs, hipEventDisableTiming) != hipSuccess) { c->err = "hipEventCreate failed"; return bail(MRT_ERR_HIP); }
    for (uint32_t i = 0; i < mrt_ctx::kEventRing; i++)
        if (hipEventCreate(&c->ev_start[i]) != hipSuccess || hipEventCreate(&c->ev_stop[i]) != hipSuccess) { c->err = "hipEventCreate failed"; return bail(MRT_ERR_HIP); }
    if (hipMalloc(&c->d_counters, 16 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemsetAsync(c->d_counters, 0, 16 * sizeof(unsigned long long), c->stream) != hipSuccess) { c->err = "counter allocation failed"; return bail(MRT_ERR_HIP); }
    if (hipHostMalloc((void**)&c->h_stats, 5 * mrt_ctx::kMaxFrameSlots * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { c->err = "pinned allocation failed"; return bail(MRT_ERR_HIP); }
    int st = alloc_frame_buffers(c);
    if (st != MRT_OK) return bail(st);
    *out = c;
    return MRT_OK;
}

void mrt_destroy(mrt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    // A context that has stalled (MRT_ERR_STALLED) may hold work that never finishes: hipFree, hipStreamDestroy and
    // hipEventDestroy all wait for it.  Its device resources are then left to the process' exit rather than waited for.
    if (mrt::wait_all(c, "mrt_destroy") != MRT_OK || c->stalled) {
        delete c;
        return;
    }
    free_frame_buffers(c);
    for (auto& S : c->slot) {
        if (S.render_done) (void)hipEventDestroy(S.render_done);
        if (S.finalize_done) (void)hipEventDestroy(S.finalize_done);
        if (S.stats_ready) (void)hipEventDestroy(S.stats_ready);
        if (S.stream) (void)hipStreamDestroy(S.stream);
    }
    if (c->ev_inputs) (void)hipEventDestroy(c->ev_inputs);
    free_world(c);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    if (c->d_wave_log) (void)hipFree(c->d_wave_log);
    if (c->d_gather) (void)hipFree(c->d_gather);
    if (c->d_gather_stage) (void)hipFree(c->d_gather_stage);
    if (c->ev_gather) (void)hipEventDestroy(c->ev_gather);
    if (c->ev_gather_root) (void)hipEventDestroy(c->ev_gather_root);
    for (uint32_t i = 0; i < mrt_ctx::kEventRing; i++) {
        if (c->ev_start[i]) (void)hipEventDestroy(c->ev_start[i]);
        if (c->ev_stop[i]) (void)hipEventDestroy(c->ev_stop[i]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int mrt_set_shard(mrt_ctx* c, uint32_t rank, uint32_t world) {
    if (!c) return MRT_ERR_INVALID_ARG;
    if (world == 0 || rank >= world) return fail(c, MRT_ERR_INVALID_ARG, "mrt_set_shard: rank %u of %u", rank, world);
    if (c->frames_done != 0) return fail(c, MRT_ERR_STATE, "mrt_set_shard: frames already rendered; call mrt_reset first");
    HIP_TRY(c, hipSetDevice(c->device));
    MRT_TRY(mrt::wait_all(c, __func__));
    c->shard_rank = rank; c->shard_world = world;
    return alloc_frame_buffers(c);
}

int mrt_set_stream(mrt_ctx* c, void* s) {
    if (!c) return MRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    MRT_TRY(mrt::wait_all(c, __func__));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    c->inputs_dirty = true;
    return MRT_OK;
}

int mrt_set_world_raw(mrt_ctx* c, const void* world, size_t world_bytes, const float* vec4, size_t n_vec4,
                      const float* f32, size_t n_f32, const int32_t* i32, size_t n_i32) {
    if (!c || !world) return MRT_ERR_INVALID_ARG;
    const auto t_begin = std::chrono::steady_clock::now();
    // 64 bytes = the reference's raw::World (lib.rs:676-684) as it is; 80 = with the DielectricRange extension.
    // Only world_bytes bytes of the caller's struct are read; a 64-byte World has no dielectrics.
    if (world_bytes != MRT_WORLD_BYTES_REFERENCE && world_bytes != sizeof(mrt_world))
        return fail(c, MRT_ERR_INVALID_ARG, "mrt_set_world_raw: world_bytes %zu is neither %d (raw::World) nor %zu (mrt_world)",
                    world_bytes, MRT_WORLD_BYTES_REFERENCE, sizeof(mrt_world));
    mrt_world w_copy;
    std::memset(&w_copy, 0, sizeof w_copy);
    std::memcpy(&w_copy, world, world_bytes);
    const mrt_world* const w = &w_copy;
    if ((n_vec4 && !vec4) || (n_f32 && !f32) || (n_i32 && !i32)) return fail(c, MRT_ERR_INVALID_ARG, "mrt_set_world_raw: null array");
    const int64_t n = w->spheres.length;
    if (n < 0 || n > (int64_t)mrt::kMaxSpheres) return fail(c, MRT_ERR_BAD_SCENE, "spheres.length %lld out of range [0, %u]", (long long)n, mrt::kMaxSpheres);
    auto in_range = [](int64_t base, int64_t len, size_t cap) { return base >= 0 && len >= 0 && (uint64_t)(base + len) <= cap; };
    if (!in_range(w->spheres.center_base_idx, n, n_vec4) || !in_range(w->spheres.radius_base_idx, n, n_f32) ||
        !in_range(w->spheres.material_ty_base_idx, n, n_i32) || !in_range(w->spheres.material_idx_base_idx, n, n_i32) ||
        !in_range(w->lambertians.albedo_base_idx, w->lambertians.length, n_vec4) ||
        !in_range(w->metals.albedo_base_idx, w->metals.length, n_vec4) ||
        !in_range(w->metals.fuzz_base_idx, w->metals.length, n_f32) ||
        !in_range(w->dielectrics.ior_base_idx, w->dielectrics.length, n_f32))
        return fail(c, MRT_ERR_BAD_SCENE, "a World range points outside its data array");
    // geometry must be finite and moderate so that no discriminant can overflow to inf/NaN
    const float kLim = 1.0e7f;
    for (int64_t i = 0; i < n; i++) {
        const float* ctr = vec4 + 4 * (w->spheres.center_base_idx + i);
        const float r = f32[w->spheres.radius_base_idx + i];
        if (!finite_in_range(ctr[0], kLim) || !finite_in_range(ctr[1], kLim) || !finite_in_range(ctr[2], kLim) ||
            !finite_in_range(r, kLim))
            return fail(c, MRT_ERR_BAD_SCENE, "sphere %lld: centre/radius not finite or |v| > 1e7", (long long)i);
        const int32_t ty = i32[w->spheres.material_ty_base_idx + i];
        const int32_t mi = i32[w->spheres.material_idx_base_idx + i];
        const int32_t len = ty == MRT_LAMBERTIAN ? w->lambertians.length : ty == MRT_METAL ? w->metals.length
                          : ty == MRT_DIELECTRIC ? w->dielectrics.length : INT32_MAX;   // unknown ty: absorbs, idx unused
        if (mi < 0 || (ty >= MRT_LAMBERTIAN && ty <= MRT_DIELECTRIC && mi >= len))
            return fail(c, MRT_ERR_BAD_SCENE, "sphere %lld: material index %d out of range for type %d", (long long)i, mi, ty);
    }
    HIP_TRY(c, hipSetDevice(c->device));
    MRT_TRY(mrt::wait_all(c, __func__));
    free_world(c);

    // exact-test records, in the reference's sphere order
    std::vector<mrt::SphereRec> recs((size_t)n ? (size_t)n : 1);
    for (int64_t i = 0; i < n; i++) {
        const float* ctr = vec4 + 4 * (w->spheres.center_base_idx + i);
        const float r = f32[w->spheres.radius_base_idx + i];
        recs[(size_t)i] = mrt::SphereRec{ctr[0], ctr[1], ctr[2], -(r * r)};
    }
    // bounding-sphere hierarchy over spatially close spheres; the sweep tests its top level
    Hierarchy hier;
    build_hierarchy(vec4 + 4 * w->spheres.center_base_idx, f32 + w->spheres.radius_base_idx, (uint32_t)n,
                    c->cluster_factor, c->max_levels, c->top_target, hier);
    const uint32_t n_padded = (uint32_t)hier.top.size();
    auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
        if (e != hipSuccess || !bytes) return e;
        return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    HIP_TRY(c, upload((void**)&c->d_spheres, recs.data(), recs.size() * sizeof(mrt::SphereRec)));
    HIP_TRY(c, upload((void**)&c->d_clusters, hier.top.data(), hier.top.size() * sizeof(mrt::SphereRec)));
    HIP_TRY(c, upload((void**)&c->d_nodes, hier.nodes.data(), hier.nodes.size() * sizeof(mrt::SphereRec)));
    if (hier.n_members > 1024u) {           // large scenes (the kernel's !SMALL layouts) walk the boxes
        std::vector<mrt::BoxFull> full;
        std::vector<mrt::BoxRec> dev;
        boxes_top_down(hier, false, full, &c->box_cluster_first, &c->box_cluster_parent_first);
        pack_boxes(full, dev);
        HIP_TRY(c, upload((void**)&c->d_boxes, dev.data(), dev.size() * sizeof(mrt::BoxRec)));
        boxes_top_down(hier, true, full, &c->box_cluster_first, &c->box_cluster_parent_first);
        pack_boxes(full, dev);
        HIP_TRY(c, upload((void**)&c->d_boxes_open, dev.data(), dev.size() * sizeof(mrt::BoxRec)));
    }
    c->box_quad = hier.box_quad;
    c->box_kc = hier.box_kc;
    {
        std::vector<uint16_t> top_mfma;
        double max_c2 = 0.0, med_r2 = 0.0;
        size_t n_real = 0;
        build_top_mfma(hier.top, top_mfma, c->mfma_origin, &max_c2, &med_r2, &n_real);
        HIP_TRY(c, upload((void**)&c->d_top_mfma, top_mfma.data(), top_mfma.size() * sizeof(uint16_t)));
        // The matrix-core sweep inflates R^2 by 2^-13 (o.o + C.C + R^2), o and C relative to mfma_origin; rays
        // start in or around the scene.
        // Selected where that stays below about a tenth of the typical R^2 (mrt_redraw checks the camera's
        // own distance the same way) and there are enough records to fill most of a 32-record tile.
        c->mfma_r2_ref = med_r2;
        c->mfma_reach = 0.0;
        for (int64_t i = 0; i < n; i++) {
            const float* ctr = vec4 + 4 * (w->spheres.center_base_idx + i);
            double d2 = 0.0;
            for (int k = 0; k < 3; k++) { const double d = (double)ctr[k] - (double)c->mfma_origin[k]; d2 += d * d; }
            c->mfma_reach = std::max(c->mfma_reach, std::sqrt(d2) + std::fabs((double)f32[w->spheres.radius_base_idx + i]));
        }
        c->mfma_scene_ok = n_real >= 24 && med_r2 > 0.0 && kMfmaSlack * 2.0 * max_c2 <= 0.1 * med_r2;
    }
    HIP_TRY(c, upload((void**)&c->d_member_index, hier.member_index.data(), hier.member_index.size() * sizeof(uint32_t)));
    // what shading a hit on sphere i reads, gathered per sphere (bit copies of the SoA entries)
    std::vector<float> shade(8 * ((size_t)n ? (size_t)n : 1), 0.0f);
    for (int64_t i = 0; i < n; i++) {
        const float* ctr = vec4 + 4 * (w->spheres.center_base_idx + i);
        float* sh = shade.data() + 8 * (size_t)i;
        sh[0] = ctr[0]; sh[1] = ctr[1]; sh[2] = ctr[2];
        sh[3] = f32[w->spheres.radius_base_idx + i];
        const int32_t ty = i32[w->spheres.material_ty_base_idx + i];
        const int32_t mi = i32[w->spheres.material_idx_base_idx + i];
        sh[4] = sh[5] = sh[6] = 1.0f; sh[7] = 0.0f;
        if (ty == MRT_LAMBERTIAN) {
            std::memcpy(sh + 4, vec4 + 4 * (w->lambertians.albedo_base_idx + mi), 3 * sizeof(float));
        } else if (ty == MRT_METAL) {
            std::memcpy(sh + 4, vec4 + 4 * (w->metals.albedo_base_idx + mi), 3 * sizeof(float));
            sh[7] = f32[w->metals.fuzz_base_idx + mi];
        } else if (ty == MRT_DIELECTRIC) {
            // A Dielectric attenuates by (1,1,1) (a constant in the kernel), so its colour slots carry what its
            // scatter derives from the sphere alone, evaluated here with the same f32 operations in the same order
            // (correctly rounded '/', no contraction): ri = 1/ior for a front-face hit, and the Schlick r0 =
            // ((1-ri)/(1+ri))^2 for either face.  Bit-identical to evaluating them per hit.
            const float ior = f32[w->dielectrics.ior_base_idx + mi];
            auto schlick_r0 = [](float ri) { float r0 = (1.0f - ri) / (1.0f + ri); return r0 * r0; };
            const float inv_ior = 1.0f / ior;
            sh[4] = inv_ior; sh[5] = schlick_r0(inv_ior); sh[6] = schlick_r0(ior);
            sh[7] = ior;
        }
    }
    HIP_TRY(c, upload((void**)&c->d_shade, shade.data(), shade.size() * sizeof(float)));
    HIP_TRY(c, upload((void**)&c->d_vec4, vec4, n_vec4 * 4 * sizeof(float)));
    HIP_TRY(c, upload((void**)&c->d_f32, f32, n_f32 * sizeof(float)));
    HIP_TRY(c, upload((void**)&c->d_i32, i32, n_i32 * sizeof(int32_t)));
    for (auto& S : c->slot) S.cost_valid = false;
    c->width.div = 0;                   // (the launch-width controller starts over with the new workload)
    c->inputs_dirty = true;
    c->world = *w;
    c->n_spheres = (uint32_t)n;
    c->n_padded = n_padded;
    c->levels = hier.levels; c->n_nodes = (uint32_t)hier.nodes.size(); c->n_members = hier.n_members;
    for (uint32_t k = 0; k < mrt::kMaxLevels; k++) c->level_base[k] = hier.level_base[k];
    c->n_direct = hier.n_direct; c->direct_first = hier.direct_first;
    for (uint32_t k = 0; k < mrt::kMaxDirect; k++) { c->direct[k] = hier.direct[k]; c->direct_index[k] = hier.direct_index[k]; }
    c->have_world = true;
    c->set_world_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return MRT_OK;
}

int mrt_debug_last_set_world_ms(mrt_ctx* c, float* ms) {
    if (!c || !ms) return MRT_ERR_INVALID_ARG;
    if (!c->have_world) return fail(c, MRT_ERR_NO_SCENE, "mrt_debug_last_set_world_ms: no scene");
    *ms = c->set_world_ms;
    return MRT_OK;
}

int mrt_set_world(mrt_ctx* c, const mrt_sphere* spheres, size_t n) {
    if (!c || (!spheres && n)) return MRT_ERR_INVALID_ARG;
    std::vector<float> vec4(8 * n + 4), f32(2 * n + 1);
    std::vector<int32_t> i32(2 * n + 1);
    mrt_world w;
    size_t nv = 0, nf = 0, ni = 0;
    int st = mrt_pack_world(spheres, n, &w, vec4.data(), 2 * n + 1, &nv, f32.data(), 2 * n + 1, &nf, i32.data(), 2 * n + 1, &ni);
    if (st != MRT_OK) return fail(c, st, "mrt_set_world: packing failed (%s)", mrt_status_string(st));
    return mrt_set_world_raw(c, &w, sizeof w, vec4.data(), nv, f32.data(), nf, i32.data(), ni);
}

int mrt_set_camera(mrt_ctx* c, const mrt_camera* cam) {
    if (!c || !cam) return MRT_ERR_INVALID_ARG;
    mrt_camera_raw raw;
    int st = mrt_camera_derive(cam, &raw);
    if (st != MRT_OK) return fail(c, st, "mrt_set_camera: degenerate or invalid camera");
    // like the geometry (mrt_set_world_raw): moderate, so that no discriminant of a camera ray can overflow
    for (int k = 0; k < 3; k++)
        if (raw.mode != 0 && !(std::fabs(raw.origin[k]) <= 1.0e7f && std::fabs(raw.ru[k]) <= 1.0e7f && std::fabs(raw.rv[k]) <= 1.0e7f))
            return fail(c, MRT_ERR_INVALID_ARG, "mrt_set_camera: |lookfrom| or the lens radius exceeds 1e7");
    c->cam_raw = raw;
    for (auto& S : c->slot) S.cost_valid = false;
    c->width.div = 0;                   // (the launch-width controller starts over with the new workload)
    return MRT_OK;
}

int mrt_shard_info(mrt_ctx* c, uint32_t* rank, uint32_t* world, uint32_t* local_rows, uint32_t* width) {
    if (!c) return MRT_ERR_INVALID_ARG;
    if (rank) *rank = c->shard_rank;
    if (world) *world = c->shard_world;
    if (local_rows) *local_rows = c->local_bands * mrt::kBandRows;
    if (width) *width = c->args.width;
    return MRT_OK;
}

// copy between a full bottom-up image on the host and this shard's packed rows on the device
static int copy_rows(mrt_ctx* c, void* device_base, void* host_full, size_t texel_bytes, bool to_device) {
    const uint32_t W = c->args.width, H = c->args.height;
    const size_t band_bytes = (size_t)mrt::kBandRows * W * texel_bytes;
    for (uint32_t b = 0; b < c->local_bands; b++) {
        const uint32_t gb = b * c->shard_world + c->shard_rank;
        const uint32_t y0 = gb * mrt::kBandRows;
        if (y0 >= H) break;
        const uint32_t rows = (H - y0 < mrt::kBandRows) ? H - y0 : mrt::kBandRows;
        char* dptr = (char*)device_base + b * band_bytes;
        char* hptr = (char*)host_full + (size_t)y0 * W * texel_bytes;
        const size_t bytes = (size_t)rows * W * texel_bytes;
        if (to_device) HIP_TRY(c, hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, c->stream));
        else HIP_TRY(c, hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, c->stream));
    }
    MRT_TRY(mrt::wait_stream(c, c->stream, __func__));
    return MRT_OK;
}

int mrt_set_seeds(mrt_ctx* c, const uint32_t* seeds, size_t n_u32) {
    if (!c || !seeds) return MRT_ERR_INVALID_ARG;
    if (n_u32 != (size_t)c->args.width * c->args.height * 4) return fail(c, MRT_ERR_INVALID_ARG, "mrt_set_seeds: expected W*H*4 u32");
    HIP_TRY(c, hipSetDevice(c->device));
    MRT_TRY(mrt::wait_all(c, __func__));
    c->inputs_dirty = true;
    return copy_rows(c, c->d_seeds, const_cast<uint32_t*>(seeds), 16, true);
}

int mrt_read_seeds(mrt_ctx* c, uint32_t* out, size_t cap) {
    if (!c || !out) return MRT_ERR_INVALID_ARG;
    const size_t n = local_texels(c) * 4;
    if (cap < n) return fail(c, MRT_ERR_TOO_SMALL, "mrt_read_seeds: need %zu u32", n);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(out, c->d_seeds, n * 4, hipMemcpyDeviceToHost, c->stream));
    MRT_TRY(mrt::wait_stream(c, c->stream, __func__));
    return MRT_OK;
}

}  // extern "C"

// KParams::mfma_scale / mfma_neg_k2_pair for rays and records within `all` of the sweep's origin (mrt_debug_mfma_scale)
static void mfma_scales(double all, float scale[4], uint32_t* neg_k2_pair) {
    if (!(all > 1e-30)) all = 1.0;
    int e = 0;
    (void)std::frexp(5.01 * all, &e);                       // 5.01 all < 2^e
    const double K = std::ldexp(1.0, -(e + 1)), K2 = K * K;
    scale[0] = (float)((double)mrt::kBoundStretch * K);
    scale[1] = (float)(2.0 * K2);
    scale[2] = (float)(-(1.0 - kMfmaSlack) * K2);
    scale[3] = (float)(16.0 * all * all);
    const uint32_t nk2 = (uint32_t)bf16_rne((float)-K2);    // a power of two: exact
    *neg_k2_pair = nk2 | (nk2 << 16);
}

// the scene / hierarchy / sweep-variant part of the kernel arguments (everything that does not depend on the frame)
static void fill_scene_params(const mrt_ctx* c, mrt::KParams& p) {
    p.world = c->world;
    p.cam = c->cam_raw;
    p.n_spheres = c->n_spheres;
    p.n_padded = c->n_padded;
    { const uint32_t ch = (c->n_padded + mrt::kChunk - 1) / mrt::kChunk; p.mask_chunks = ch < 16u ? ch : 16u; }
    {
        p.use_mfma = use_matrix_core_sweep(c) ? 1u : 0u;
        for (int k = 0; k < 3; k++) p.mfma_origin[k] = c->mfma_origin[k];
        // The sweep squares K oc.ds through an instruction that saturates at 1 (kernels.hip, mfma_sweep_tile), K a power of
        // two: rays start on the camera's lens or on a sphere, i.e. within `all` of mfma_origin; the sweep admits origins up
        // to 4 x that (others take the literal loop), records lie within `all`, |ds| < 1.001: |K oc.ds| < 5.01 all K <= 1/2.
        double cam_d2 = 0.0, lens = 0.0;
        for (int k = 0; k < 3; k++) {
            const double d = (c->cam_raw.mode ? (double)c->cam_raw.origin[k] : 0.0) - (double)c->mfma_origin[k];
            cam_d2 += d * d;
        }
        if (c->cam_raw.mode) {
            double u2 = 0.0, v2 = 0.0;
            for (int k = 0; k < 3; k++) { u2 += (double)c->cam_raw.ru[k] * c->cam_raw.ru[k]; v2 += (double)c->cam_raw.rv[k] * c->cam_raw.rv[k]; }
            lens = std::sqrt(u2) + std::sqrt(v2);
        }
        mfma_scales(std::max(c->mfma_reach, std::sqrt(cam_d2) + lens), p.mfma_scale, &p.mfma_neg_k2_pair);
    }
    p.levels = c->levels; p.n_nodes = c->n_nodes; p.n_members = c->n_members;
    // small scenes: the top queue holds a ray's candidates among ALL top records; large scenes: the wave's one work stack
    p.box_lds_count = c->n_members <= 1024u ? 0u : mrt::large_scene_box_lds_count(c->n_padded, c->levels, p.mask_chunks, mrt::kBoxLdsCap);
    p.gen_cap = c->n_members <= 1024u ? 576u : mrt::large_scene_stack_cap(p.mask_chunks, p.box_lds_count);
    for (uint32_t k = 0; k < mrt::kMaxLevels; k++) p.level_base[k] = c->level_base[k];
    // large scenes only (kernels.hip: !SMALL): every node's box, in the kernel's top-down numbering
    p.boxes = c->boxes_mode == 0 ? c->d_boxes_open : c->d_boxes;
    p.box_cluster_first = c->box_cluster_first; p.box_cluster_parent_first = c->box_cluster_parent_first;
    p.box_quad = c->box_quad ? 1u : 0u;
    p.box_kc = c->box_kc;
    p.n_direct = c->n_direct; p.direct_first = c->direct_first;
    for (uint32_t k = 0; k < mrt::kMaxDirect; k++) { p.direct[k] = c->direct[k]; p.direct_index[k] = c->direct_index[k]; }
    p.cus = c->cus;
    p.spheres = c->d_spheres; p.clusters = c->d_clusters; p.nodes = c->d_nodes; p.top_mfma = c->d_top_mfma; p.member_index = c->d_member_index; p.vec4_data = c->d_vec4; p.shade = c->d_shade; p.f32_data = c->d_f32; p.i32_data = c->d_i32;
}

// How many of `k` side streams of this process really run at a time: one clock-bounded single-wave kernel per stream (0.5 ms
// each; launch_hold) stamps its start and end on the device's wall clock; the answer is the largest number of them resident at
// one instant.  HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the host set the variable
// before its first HIP call) and kernels of streams that share one serialise: round 4 measured 8 frames in flight running 2.7
// at a time on the default, all 8 on 16 queues (C5's 1/8 share 1,050 -> 2,370 Msamples/s).  Called once per context, with
// nothing in flight, when the schedule first asks for more than two frames.
static int probe_stream_concurrency(mrt_ctx* c, uint32_t k, float* out) {
    if (k < 2u) k = 2u;
    if (k > mrt_ctx::kMaxFrameSlots) k = mrt_ctx::kMaxFrameSlots;
    for (uint32_t i = 0; i < k; i++) HIP_TRY(c, create_slot_streams(c->slot[i]));
    MRT_TRY(mrt::wait_all(c, "probe_stream_concurrency"));
    unsigned long long* const stamps = c->h_stats + 3 * mrt_ctx::kMaxFrameSlots;      // pinned, device-visible: 2 per stream
    uint32_t best = 0;
    for (int pass = 0; pass < 2; pass++) {           // (the first pass also pays for the code object and the queues' creation)
        std::memset(stamps, 0, 2 * mrt_ctx::kMaxFrameSlots * sizeof(unsigned long long));
        for (uint32_t i = 0; i < k; i++) {
            const int e = mrt::launch_hold(50000ull, 1u << 14, stamps + 2 * i, c->slot[i].stream);
            if (e) return fail(c, MRT_ERR_HIP, "probe_stream_concurrency: launch failed: %s", hipGetErrorString((hipError_t)e));
        }
        for (uint32_t i = 0; i < k; i++) MRT_TRY(mrt::wait_stream(c, c->slot[i].stream, "probe_stream_concurrency"));
        best = 0;
        for (uint32_t i = 0; i < k; i++) {           // at the start of kernel i: how many are resident?
            uint32_t n = 0;
            for (uint32_t j = 0; j < k; j++) n += (stamps[2 * j] <= stamps[2 * i] && stamps[2 * i] < stamps[2 * j + 1]) ? 1u : 0u;
            best = std::max(best, n);
        }
    }
    *out = (float)best;
    return MRT_OK;
}

// The most frames in flight this process can really run side by side: kMaxFrameSlots where the probe says so, else the largest
// power of two it supports (>= 2), with ONE line of warning behind mrt_last_error(NULL).
static int probe_max_slots(mrt_ctx* c) {
    if (c->slots_probed) return MRT_OK;
    float conc = 0.0f;
    MRT_TRY(probe_stream_concurrency(c, mrt_ctx::kMaxFrameSlots, &conc));
    c->slots_probed = true;
    // (all sixteen or eight: with 15 of 16 -- what GPU_MAX_HW_QUEUES=16 gives, the context's own stream holds a queue too -- two
    // of the sixteen frames take turns on one queue, and C5's 1/8 share renders 2,790 Msamples/s instead of 3,750, less than
    // with eight frames in flight)
    uint32_t cap = mrt_ctx::kMaxFrameSlots;
    while (cap > 2u && conc < (cap == 8u ? 7.0f : (float)cap)) cap /= 2u;
    c->max_slots = cap;
    if (cap < mrt_ctx::kMaxFrameSlots) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "myraytracer_amd: only %.0f of %u side streams run at a time in this process: at most %u frames in flight "
                      "(set GPU_MAX_HW_QUEUES=20 before the process' first HIP call: INTEGRATION.md 2a)", conc, mrt_ctx::kMaxFrameSlots, cap);
        g_err = buf;
        static const bool trace = std::getenv("MRT_TRACE_WIDTH") != nullptr;
        if (trace) std::fprintf(stderr, "%s\n", buf);
    }
    return MRT_OK;
}

static mrt::WidthWorkload width_workload(const mrt_ctx* c, bool counter) {
    mrt::WidthWorkload w;
    w.n_tiles = c->n_tiles; w.n_waves = c->n_waves; w.max_slots = c->max_slots;
    w.spp = c->locals.samples_per_frame; w.n_members = c->n_members; w.counter = counter ? 1u : 0u;
    return w;
}

// a new setting: the frames of the old one drain first; samples and timings count again from the THIRD generation of frames at
// the new one (the first starts on an empty chip -- a change waits for everything in flight -- and the second still inherits
// its convoys: judged on those, C3 read 4 % faster at a quarter width, where it renders 2 % less)
static void width_restart_measurement(mrt_ctx* c) {
    c->width_timing = false;
    c->width_valid_from = c->frame_seq + 2u * mrt::width_frames_in_flight(c->width.div, c->width.mult, c->max_slots);
    c->stat_base.valid = c->stat_last.valid = false;
}

// The launch-width controller's bookkeeping for the frame about to be launched (adaptive launches only: one frame per launch,
// no diagnostic override): the setting for a new workload, the back-pressure, the lane-utilisation samples that have landed,
// the measurement windows and -- through width_policy.h -- the trials.  *want = frames in flight, *frames_running = earlier
// frames whose render kernels are still queued or running.
static int schedule_frame(mrt_ctx* c, bool counter, uint32_t* want, uint32_t* frames_running) {
    static const bool trace = std::getenv("MRT_TRACE_WIDTH") != nullptr;      // diagnostics: every decision, on stderr
    mrt::WidthWorkload w = width_workload(c, counter);
    if (c->width.div == 0) {
        if (c->hint_div != 0) {                  // the caller's setting (mrt_set_schedule_hint)
            c->width = mrt::WidthState();
            c->width.div = c->hint_div; c->width.mult = c->hint_mult; c->width.settled = 1u;
        } else {
            mrt::width_policy_start(c->width, w);
            // a setting this context has already settled at for the same workload returns without trials
            for (const auto& m : c->width_memo)
                if (m.n_tiles == w.n_tiles && m.spp == w.spp && m.large == (w.n_members > 1024u ? 1u : 0u) && m.counter == w.counter &&
                    m.n_spheres == c->n_spheres) { c->width.div = m.div; c->width.mult = m.mult; c->width.settled = 1u; }
        }
        width_restart_measurement(c);
    }
    // The host may not run further ahead than the frames in flight: before a slot is used again, its previous frame's
    // render kernel has completed (a swap-chain's back-pressure; the GPU still holds a full set of frames, queued or
    // running).  It bounds the queued work and is what lets the samples below arrive while they can still matter -- a
    // caller that issues its redraws in one burst would otherwise see none of them before its last call.
    {
        const uint32_t own = (uint32_t)(c->frame_seq % c->frame_slots);
        mrt_ctx::FrameSlot& Own = c->slot[own];
        if (Own.stats_pending) {
            char what[160];
            std::snprintf(what, sizeof what, "mrt_redraw: back-pressure of slot %u (render kernel of frame %llu, next frame %llu, %u frames in flight)",
                          own, (unsigned long long)Own.stats_seq, (unsigned long long)c->frame_seq, c->frame_slots);
            MRT_TRY(mrt::wait_event(c, Own.stats_ready, what));
        }
    }
    // lane-utilisation samples that have landed (an event that is not ready yet is looked at next time), and the frames still
    // queued or running
    uint32_t running = 0;
    for (uint32_t i = 0; i < c->frame_slots; i++) {
        mrt_ctx::FrameSlot& T = c->slot[i];
        if (T.render_pending) {
            if (hipEventQuery(T.render_done) == hipSuccess) T.render_pending = false; else running++;
        }
        if (!T.stats_pending || hipEventQuery(T.stats_ready) != hipSuccess) continue;
        T.stats_pending = false;
        if (T.stats_seq < c->width_valid_from) continue;
        mrt_ctx::LaneStat st{T.stats_seq, c->h_stats[3 * i], c->h_stats[3 * i + 2], true};
        if (!c->stat_base.valid || st.seq < c->stat_base.seq) c->stat_base = st;
        if (!c->stat_last.valid || st.seq > c->stat_last.seq) c->stat_last = st;
    }
    (void)hipGetLastError();        // (hipEventQuery's hipErrorNotReady is not an error)
    // How many frames the CALLER keeps in flight: the most seen still queued or running over the last (frames in flight) calls
    // -- not this call's count alone, which dips whenever a convoy of frames has just ended (launched a little wider, the next
    // frame then holds more of the chip and the dips feed on themselves: C5's 1/8 share 3,066 -> 2,840 Msamples/s), and which is
    // 0, 1, 2, ... while a burst of calls fills an empty pipeline.  A new setting starts from "the caller keeps them all in
    // flight".
    {
        const uint32_t window = mrt::width_frames_in_flight(c->width.div, c->width.mult, c->max_slots);
        if (c->running_seen_n != window) {          // (a new setting, or the first call)
            c->running_seen_n = window;
            for (uint32_t i = 0; i < window; i++) c->running_seen[i] = window - 1u;
        }
        c->running_seen[c->frame_seq % window] = running;
        // ... except that NOTHING running at three calls in a row is a caller that waits for every frame (a pipeline that is
        // kept full never shows that): known at once, not after a window of up to sixteen slow frames
        c->nothing_running_calls = running == 0u ? c->nothing_running_calls + 1u : 0u;
        if (c->nothing_running_calls >= 3u)
            for (uint32_t i = 0; i < window; i++) c->running_seen[i] = 0u;
        uint32_t most = 0;
        for (uint32_t i = 0; i < window; i++) most = std::max(most, c->running_seen[i]);
        *frames_running = most;
    }
    // A measurement window: from the first frame launched at the current setting with the pipeline full, over
    // 2 x (frames in flight) + 2 frames -- their lane utilisation (the samples above) and, the calls being paced by
    // the completions (the back-pressure above), their rate on the host's clock -- and over at least 20 ms: frames of a
    // fifth of a millisecond (C1, 1 spp) filled a window in 2-3 ms of host time, whose jitter decided 1 trial in 9 the wrong way.
    const uint32_t in_flight = std::max(2u, c->width.div) * c->width.mult;
    const auto now = std::chrono::steady_clock::now();
    if (!c->width.settled && !c->width_timing && c->frame_seq >= c->width_valid_from) {
        c->width_timing = true;
        c->width_t0_seq = c->frame_seq;
        c->width_t0 = now;
    }
    if (c->width_timing && c->frame_seq >= c->width_t0_seq + 2u * in_flight + 2u &&
        std::chrono::duration<double>(now - c->width_t0).count() >= 0.020 && c->stat_base.valid && c->stat_last.valid &&
        c->stat_last.seq > c->stat_base.seq && c->stat_last.slots > c->stat_base.slots && c->stat_last.hits >= c->stat_base.hits) {
        mrt::WidthWindow m;
        m.util = (double)(c->stat_last.hits - c->stat_base.hits) / (double)(c->stat_last.slots - c->stat_base.slots);
        m.rate = (double)(c->frame_seq - c->width_t0_seq) / std::max(1e-9, std::chrono::duration<double>(now - c->width_t0).count());
        // The frame rate, better: frames END in convoys (the launches that share the chip start together), so a count of the calls
        // the completions let through over a window of a few convoys is off by up to a convoy -- 4 frames in 18, far beyond the
        // 3 % a trial is judged by (the first cut of this round kept a quarter width for C3 that renders 8 % less).  Every frame
        // slot is refilled the moment its frame ends (the back-pressure), so slots / (a slot's start-to-start time) is the rate
        // (Little's law), and start-to-start times are whole frames on the DEVICE's clock: the start events of frame f and of
        // frame f + slots, the next on the same slot, over the window's frames.
        {
            const uint32_t slots = c->frame_slots;
            double sum_ms = 0.0;
            uint32_t n = 0;
            for (uint64_t f = c->width_t0_seq; f + slots < c->frame_seq; f++) {
                if (c->frame_seq - f > mrt_ctx::kEventRing) continue;               // (overwritten since)
                hipEvent_t a = c->ev_start[f % mrt_ctx::kEventRing], b = c->ev_start[(f + slots) % mrt_ctx::kEventRing];
                if (hipEventQuery(b) != hipSuccess) break;                          // (not started yet, nor are the later ones)
                float ms = 0.0f;
                if (hipEventElapsedTime(&ms, a, b) == hipSuccess && ms > 0.0f) { sum_ms += ms; n++; }
            }
            (void)hipGetLastError();
            if (n >= std::max(2u, slots / 2u)) m.rate = (double)slots * 1e3 * (double)n / sum_ms;     // (else: the host's count above)
        }
        if (trace) std::fprintf(stderr, "mrt width: frame %llu: div %u x %u, window %llu frames, utilisation %.4f, %.2f frames/s%s\n",
                                (unsigned long long)c->frame_seq, c->width.div, c->width.mult, (unsigned long long)(c->frame_seq - c->width_t0_seq),
                                m.util, m.rate, c->width.prev_div != 0 ? " (trial)" : "");
        mrt::width_policy_step(c->width, w, m);
        if (c->width.settled) {
            if (trace) std::fprintf(stderr, "mrt width: settled at div %u x %u\n", c->width.div, c->width.mult);
            const mrt_ctx::WidthMemo memo{w.n_tiles, w.spp, w.n_members > 1024u ? 1u : 0u, w.counter, c->n_spheres, c->width.div, c->width.mult};
            bool known = false;
            for (auto& m : c->width_memo)
                if (m.n_tiles == memo.n_tiles && m.spp == memo.spp && m.large == memo.large && m.counter == memo.counter && m.n_spheres == memo.n_spheres) {
                    m = memo;
                    known = true;
                }
            if (!known) c->width_memo.push_back(memo);
        }
        width_restart_measurement(c);
    }
    // more than two frames in flight only where they really run side by side (measured once, when a setting first asks for them)
    if (mrt::width_frames_in_flight(c->width.div, c->width.mult, mrt_ctx::kMaxFrameSlots) > 2u && !c->slots_probed) {
        MRT_TRY(probe_max_slots(c));
        // (a pinned setting keeps its width and is held to the frames that run side by side all the same: sixteen frames on
        // fewer queues take turns -- C5's 1/8 share 2,790 Msamples/s, less than eight in flight give)
        if (c->max_slots < mrt_ctx::kMaxFrameSlots && c->hint_div == 0) {       // start over within what the process can do
            w.max_slots = c->max_slots;
            mrt::width_policy_start(c->width, w);
            width_restart_measurement(c);
        }
        c->running_seen_n = 0;
    }
    *want = mrt::width_frames_in_flight(c->width.div, c->width.mult, c->max_slots);
    return MRT_OK;
}

// `want` frame slots in use from the next frame on: a change waits for the frames under way
static int set_frame_slots(mrt_ctx* c, uint32_t want) {
    if (want == c->frame_slots) return MRT_OK;
    MRT_TRY(mrt::wait_all(c, "mrt_redraw: change of the frames in flight"));
    c->frame_slots = want;
    // the further slots' streams and colour sums now, in one go: allocated on first use each would wait for the frames in flight
    for (uint32_t i = 0; i < want; i++) {
        mrt_ctx::FrameSlot& T = c->slot[i];
        HIP_TRY(c, create_slot_streams(T));
        T.stats_pending = false;
        T.render_pending = false;
        if (T.pix_acc_layers != 0) continue;
        const size_t nt = local_texels(c) ? local_texels(c) : 1;
        HIP_TRY(c, hipMalloc(&T.d_pix_acc, nt * 16));
        HIP_TRY(c, hipMemsetAsync(T.d_pix_acc, 0, nt * 16, c->stream));
        T.pix_acc_layers = 1;
    }
    return MRT_OK;
}

extern "C" {

// State::redraw, lib.rs:241-307 (raytrace pass + swap + weight/shuffle update; the present
// pass needs a window surface and is out of scope)
// `batch` >= 1 consecutive frames with ONE render launch (batch > 1: stream mode only, see mrt_render): the raytrace pass
// of State::redraw for each of them, then per frame -- in order -- the blend, the swap and the weight / shuffle update.
static int redraw_frames(mrt_ctx* c, uint32_t batch, bool frames_in_lane = false) {
    if (!c->have_world) return fail(c, MRT_ERR_NO_SCENE, "mrt_redraw: no scene (call mrt_set_world first)");
    HIP_TRY(c, hipSetDevice(c->device));
    const bool counter = c->locals.rng_mode == MRT_RNG_COUNTER;
    if (batch < 1 || batch > mrt::kMaxFrameBatch || (counter && batch != 1)) return fail(c, MRT_ERR_INVALID_ARG, "redraw_frames: batch %u", batch);
    mrt::KParams p;
    std::memset(&p, 0, sizeof p);
    p.locals = c->locals;
    fill_scene_params(c, p);
    p.shard_rank = c->shard_rank; p.shard_world = c->shard_world;
    p.seeds = c->d_seeds;
    p.counters = c->d_counters;
    p.count_draws = c->count_draws ? 1u : 0u;
    p.wave_log = nullptr;            // (stamps builds: this frame's part of the log ring, below)
    p.tiles_x = c->tiles_x; p.n_tiles = c->n_tiles;
    p.pilot_spp = c->pilot_spp;
    // Launch width and frames in flight (the controller: schedule_frame below); a change waits for the frames under way.
    const bool adaptive = batch == 1 && c->waves_per_cu_override == 0 && c->frame_slots_override == 0 && c->n_tiles != 0 &&
                          c->locals.samples_per_frame != 0u;
    uint32_t frames_running = 0;
    {
        uint32_t want = c->frame_slots_override > 0 ? (uint32_t)c->frame_slots_override : 2u;
        if (adaptive) MRT_TRY(schedule_frame(c, counter, &want, &frames_running));
        MRT_TRY(set_frame_slots(c, want));
    }
    c->last_slot = (uint32_t)(c->frame_seq % c->frame_slots);
    mrt_ctx::FrameSlot& S = c->slot[c->last_slot];
    p.tile_queue = S.d_sort_scratch + 1024;
    p.tile_order = nullptr;
    p.tile_cost = S.d_tile_cost;
    // Layers of colour sums.  Counter-RNG mode: one per block of MRT_COUNTER_BLOCK samples of the frame.  Stream
    // mode: one per frame of the batch, each with the rng_shuffle the frame would have had on its own (lib.rs:305's stand-in).
    // The slot's buffer grows on demand (a frame of this slot that is still in flight is waited for first).
    const size_t n = local_texels(c) ? local_texels(c) : 1;
    {
        const uint32_t spp = c->locals.samples_per_frame;
        uint32_t layers = batch;
        if (counter && spp > MRT_COUNTER_BLOCK) layers = (spp + MRT_COUNTER_BLOCK - 1) / MRT_COUNTER_BLOCK;
        if ((uint64_t)layers * n >= (1ull << 32) || (uint64_t)layers * c->n_tiles >= (1ull << 26))
            return fail(c, MRT_ERR_INVALID_ARG, "mrt_redraw: %u layers of colour sums over %zu pixels exceed the tile queue's range", layers, n);
        if (S.pix_acc_layers < layers) {
            MRT_TRY(mrt::wait_stream(c, S.stream, "mrt_redraw: regrowing a slot's colour sums (its side stream)"));
            MRT_TRY(mrt::wait_stream(c, c->stream, __func__));
            if (S.d_pix_acc) (void)hipFree(S.d_pix_acc);
            S.d_pix_acc = nullptr; S.pix_acc_layers = 0;
            HIP_TRY(c, hipMalloc(&S.d_pix_acc, (size_t)layers * n * 16));
            S.pix_acc_layers = layers;
        }
        // what mrt_debug_read_pixel_costs reads back: a counter-mode frame's cost is the sum over its blocks, a batch's last
        // frame is its last layer
        S.cost_first_layer = counter ? 0u : batch - 1u;
        S.cost_layers = counter ? layers : 1u;
        p.n_blocks = layers;
        p.pix_stride = (uint32_t)n;
        // a batch of SHORT frames: the queue holds every tile once, a lane renders its pixel for all frames of the batch
        const bool in_lane = frames_in_lane && !counter && batch > 1 && spp != 0;
        p.queue_layers = in_lane ? 1u : layers;
        p.lane_frames = in_lane ? batch : 1u;
        for (uint32_t b = 0; b < batch; b++) {
            if (b == 0) std::memcpy(p.layer_shuffle[0], c->locals.rng_shuffle, 16);
            else mrt_frame_shuffle(c->seed, c->frames_done > UINT32_MAX - b ? UINT32_MAX : c->frames_done + b, p.layer_shuffle[b]);   // saturating, as :300
        }
    }
    p.pix_acc = S.d_pix_acc;
    // side stream: wait for the scene / seeds uploads and for this slot's previous frame (n-2) to
    // have been finalized (its colour sums and tile costs are about to be overwritten / used)
    if (c->inputs_dirty) {
        HIP_TRY(c, hipEventRecord(c->ev_inputs, c->stream));
        c->inputs_dirty = false;
    }
    HIP_TRY(c, hipStreamWaitEvent(S.stream, c->ev_inputs, 0));
    HIP_TRY(c, hipStreamWaitEvent(S.stream, S.finalize_done, 0));
    // (an earlier frame of this slot failed half way: its queue counter was never reset -- before the pilot launch, which
    // pulls from the same queue)
    if (S.queue_dirty) HIP_TRY(c, hipMemsetAsync(p.tile_queue, 0, sizeof(uint32_t), S.stream));
    // The tile queue is ordered by the per-tile cost this slot measured two frames ago, heaviest
    // first; before the slot's first frame of a scene a small pilot launch (no output) provides
    // the estimate when the frame is long enough to pay for it.  Without an estimate: index order.
    // With no more tiles than persistent waves every tile starts at once and the order cannot matter: no pilot, no sort.
    // ... nor when a pixel's chain is a handful of bounces (fewer than 4 samples per pixel and launch): three launches saved.
    const uint32_t chain_spp = c->locals.samples_per_frame * p.lane_frames;
    // (launches outside the controller's reach -- batches, overrides -- whose chains are a handful of bounces: 8 waves per CU,
    // round 3's rule for such frames)
    uint32_t launch_waves = c->n_waves;
    if (chain_spp < 4u && c->waves_per_cu_override == 0) launch_waves = std::min(launch_waves, c->cus * 8u);
    if (adaptive) {
        // (above: launch width) -- a share of the waves the chip HOLDS for this scene's kernel: a large scene's 16 per CU, not
        // the 20 of n_waves.  Shares of n_waves had made the frames in flight ask for a quarter more waves than fit: the 1/8
        // share of C5 ran at 0.82 lane utilisation instead of 0.92, C5 itself 4 % slower
        // ... and never a smaller share than the frames that really share the chip leave (width_policy.h, width_launch_div): a
        // caller that waits for every frame gets all of it
        const uint32_t whole = std::min(c->n_waves, mrt::render_resident_waves(p));
        c->last_launch_div = mrt::width_launch_div(c->width.div, frames_running);
        c->last_frames_running = frames_running;
        launch_waves = std::max(whole / c->last_launch_div, 1u);
    }
    if (c->lpt_enabled && c->n_tiles > launch_waves && chain_spp >= 4u) {
        if (!S.cost_valid && c->locals.samples_per_frame >= 8u * c->pilot_spp) {
            int pe = mrt::launch_render(p, true, launch_waves, S.stream, &c->last_launch[1]);
            if (pe) return fail(c, MRT_ERR_HIP, "pilot launch failed: %s", hipGetErrorString((hipError_t)pe));
            S.cost_valid = true;
        }
        if (S.cost_valid) {
            int se = mrt::launch_sort_tiles(S.d_tile_cost, S.d_tile_order, S.d_sort_scratch, c->n_tiles, S.stream);
            if (se) return fail(c, MRT_ERR_HIP, "tile sort launch failed: %s", hipGetErrorString((hipError_t)se));
            p.tile_order = S.d_tile_order;
        }
    }
    const uint32_t ev = (uint32_t)(c->frame_seq % mrt_ctx::kEventRing);       // (the ring is indexed by the frame: schedule_frame reads it back)
    if (c->d_wave_log) {            // diagnostic (mrt_debug_wave_log): the frame's own part of the ring, cleared (a narrow launch leaves most of it unwritten)
        p.wave_log = c->d_wave_log + (size_t)(c->frame_seq % mrt_ctx::kWaveLogFrames) * c->wave_log_waves * 4;
        HIP_TRY(c, hipMemsetAsync(p.wave_log, 0, c->wave_log_waves * 4 * sizeof(unsigned long long), S.stream));
    }
    S.queue_dirty = true;                        // until this frame's last finalize pass has been queued
    HIP_TRY(c, hipEventRecord(c->ev_start[ev], S.stream));
    int e = mrt::launch_render(p, false, launch_waves, S.stream, &c->last_launch[0]);
    if (e) return fail(c, MRT_ERR_HIP, "render launch failed: %s", hipGetErrorString((hipError_t)e));
    HIP_TRY(c, hipEventRecord(c->ev_stop[ev], S.stream));
    HIP_TRY(c, hipEventRecord(S.render_done, S.stream));
    S.render_pending = true;
    S.render_seq = c->frame_seq;
    if (adaptive) {         // the launch-width controller's sample: cumulative world_hit calls and lane slots after this kernel
        // (counters 1 .. 3 in ONE copy: world_hit calls and lane slots of the same instant)
        HIP_TRY(c, hipMemcpyAsync(c->h_stats + 3 * c->last_slot, c->d_counters + 1, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, S.stream));
        HIP_TRY(c, hipEventRecord(S.stats_ready, S.stream));
        S.stats_seq = c->frame_seq;
        S.stats_pending = true;
    }
    // caller's stream: blend into the accumulated framebuffer (shader.wgsl:383-385) once the render is done -- frame by frame
    HIP_TRY(c, hipStreamWaitEvent(c->stream, S.render_done, 0));
    for (uint32_t b = 0; b < batch; b++) {
        p.out = c->d_fb[c->target];              // framebuffers.target  (lib.rs:250)
        p.prev = c->d_fb[c->target ^ 1];         // framebuffers.secondary (lib.rs:265)
        p.locals.framebuffer_weight = c->locals.framebuffer_weight;
        if (!counter) { p.pix_acc = (char*)S.d_pix_acc + (size_t)b * n * 16; p.n_blocks = 1; }
        int fe = mrt::launch_finalize(p, c->stream);
        if (fe) return fail(c, MRT_ERR_HIP, "finalize launch failed: %s", hipGetErrorString((hipError_t)fe));
        c->target ^= 1;                                                       // framebuffers.swap(), lib.rs:299
        if (c->frames_done != UINT32_MAX) c->frames_done++;                   // saturating_add, lib.rs:300
        c->locals.framebuffer_weight = mrt_frame_weight(c->frames_done, c->args.max_framebuffer_weight);  // :301-304
        mrt_frame_shuffle(c->seed, c->frames_done, c->locals.rng_shuffle);    // :305 (deterministic stand-in)
    }
    HIP_TRY(c, hipEventRecord(S.finalize_done, c->stream));
    S.queue_dirty = false;
    S.cost_valid = true;
    c->frame_seq++;
    c->shuffle_overridden = false;
    return MRT_OK;
}

int mrt_redraw(mrt_ctx* c) {
    if (!c) return MRT_ERR_INVALID_ARG;
    return redraw_frames(c, 1);
}

// `frames` x State::redraw.  Frames are independent until their blend (each has its own rng_shuffle and its own colour
// sums), so when the shard has too few pixels to fill the GPU -- a pixel is one sequential chain of samples -- one launch
// renders up to kMaxFrameBatch consecutive frames: a lane that finishes a pixel of frame f takes one of frame f+1.  Every
// frame's image is the one mrt_redraw would have produced.
static constexpr uint64_t kBatchBytes = 1ull << 30;     // colour sums of one launch's frames (x 2 slots): 32 frames of 1080p, 8 of 4K
int mrt_render(mrt_ctx* c, uint32_t frames) {
    if (!c) return MRT_ERR_INVALID_ARG;
    while (frames != 0) {
        uint32_t batch = 1;
        if (c->batch_frames && frames >= 2 && c->locals.rng_mode == MRT_RNG_PIXEL_STREAM && !c->shuffle_overridden && c->n_tiles != 0) {
            // too few pixels to fill the GPU (fewer than two per lane): about six pixel chains per lane (they differ 10 x in length)
            // -- for SHORT chains only: from 64 samples per pixel on, frames launched one by one run eight at a time on an eighth
            // of the waves each (redraw_frames), which packs the lanes better than the layers of a batch do (C5's 1/8 share:
            // 2,380 Msamples/s at 0.68 lane utilisation in batches of 7, 2,660 at 0.92 frame by frame)
            uint32_t want = (c->n_tiles < 2u * c->n_waves && c->locals.samples_per_frame < 64u) ? (6u * c->n_waves + c->n_tiles - 1u) / c->n_tiles : 1u;
            // too short a frame (1 spp interactive accumulation: 0.2 ms of work behind six launches): about 128 M samples per launch
            const uint64_t per_frame = (uint64_t)c->n_tiles * 64u * std::max(c->locals.samples_per_frame, 1u);
            want = std::max<uint64_t>(want, ((128ull << 20) + per_frame - 1) / per_frame);
            batch = std::min(std::min(frames, (uint32_t)mrt::kMaxFrameBatch), std::max(want, 1u));
            // every frame of a batch parks its colour sums in a layer of its own (16 B per pixel): at most kBatchBytes per slot
            const uint64_t layer_bytes = (uint64_t)std::max<size_t>(local_texels(c), 1) * 16u;
            batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(batch, kBatchBytes / layer_bytes));
        }
        // Two reasons to batch, two forms (kernels.hip): a shard with too few pixels needs more pixel chains at once -- the
        // frames of the batch become layers of the tile queue; a frame that is merely short has pixels enough -- the lane that
        // takes a pixel renders it for every frame of the batch (one queue atomic / seed fetch per pixel and batch).
        // (In-lane only below 4 samples per frame: it makes a pixel's chain `batch` times longer, which costs the launch's tail
        // more than the acquisitions cost from 8 samples up -- measured, DESIGN.md: 1 spp 5,470 -> 10,360 Msamples/s, 8 spp
        // 11,260 -> 11,050, C2's 64 spp 11,640 -> 10,650.)
        const bool starved = c->n_tiles < 2u * c->n_waves;
        const bool in_lane = !starved && c->locals.samples_per_frame < 4u;      // (a batch of 1 is a plain redraw, whatever its form)
        int st = redraw_frames(c, batch, c->batch_form == 0 ? in_lane : c->batch_form == 1);
        if (st != MRT_OK) return st;
        frames -= batch;
    }
    return MRT_OK;
}

int mrt_debug_set_frame_batching(mrt_ctx* c, int enabled) {
    if (!c) return MRT_ERR_INVALID_ARG;
    c->batch_frames = enabled != 0;
    c->batch_form = enabled == 2 ? 1 : enabled == 3 ? 2 : 0;
    return MRT_OK;
}

// diagnostic: per-pixel cost (bounce-loop trips) of the last frame, this shard's packed rows
int mrt_debug_read_pixel_costs(mrt_ctx* c, uint32_t* out, size_t cap) {
    if (!c || !out) return MRT_ERR_INVALID_ARG;
    const size_t n = local_texels(c);
    if (cap < n) return fail(c, MRT_ERR_TOO_SMALL, "mrt_debug_read_pixel_costs: need %zu", n);
    HIP_TRY(c, hipSetDevice(c->device));
    MRT_TRY(mrt::wait_all(c, __func__));
    const mrt_ctx::FrameSlot& S = c->slot[c->last_slot];
    std::vector<uint32_t> tmp(n * 4), layer(n * 4);
    HIP_TRY(c, hipMemcpyAsync(tmp.data(), (const char*)S.d_pix_acc + (size_t)S.cost_first_layer * n * 16, n * 16, hipMemcpyDeviceToHost, c->stream));
    MRT_TRY(mrt::wait_stream(c, c->stream, __func__));
    for (uint32_t b = 1; b < S.cost_layers; b++) {       // counter mode: a pixel's cost is the sum over its blocks
        HIP_TRY(c, hipMemcpyAsync(layer.data(), (const char*)S.d_pix_acc + (size_t)(S.cost_first_layer + b) * n * 16, n * 16, hipMemcpyDeviceToHost, c->stream));
        MRT_TRY(mrt::wait_stream(c, c->stream, __func__));
        for (size_t i = 0; i < n; i++) tmp[4 * i + 3] += layer[4 * i + 3];
    }
    for (size_t i = 0; i < n; i++) out[i] = tmp[4 * i + 3];
    return MRT_OK;
}

// diagnostic / tuning: cluster growth factor used by the NEXT mrt_set_world* call
int mrt_debug_set_cluster_factor(mrt_ctx* c, float factor) {
    if (!c || !(factor >= 0.0f)) return MRT_ERR_INVALID_ARG;
    c->cluster_factor = factor;
    return MRT_OK;
}

int mrt_debug_set_sweep(mrt_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 2) return MRT_ERR_INVALID_ARG;
    c->sweep_mode = mode;
    return MRT_OK;
}

int mrt_debug_build_hierarchy(const mrt_sphere* spheres, size_t n, uint32_t max_levels, uint32_t top_target,
                              float* top_out, size_t top_cap, float* nodes_out, size_t nodes_cap,
                              uint32_t* member_index_out, size_t member_cap, uint16_t* mfma_out, size_t mfma_cap,
                              float mfma_origin_out[3], uint32_t info[10]) {
    if ((!spheres && n) || !info || max_levels < 1 || max_levels > mrt::kMaxLevels || n > mrt::kMaxSpheres)
        return MRT_ERR_INVALID_ARG;
    std::vector<float> centers(4 * (n ? n : 1)), radii(n ? n : 1);
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) centers[4 * i + k] = spheres[i].center[k];
        centers[4 * i + 3] = 1.0f;
        radii[i] = spheres[i].radius;
    }
    Hierarchy h;
    build_hierarchy(centers.data(), radii.data(), (uint32_t)n, 8.0f, max_levels, top_target, h);
    std::vector<uint16_t> mf;
    float origin[3];
    double max_c2, med_r2;
    size_t n_real;
    build_top_mfma(h.top, mf, origin, &max_c2, &med_r2, &n_real);
    info[0] = h.levels; info[1] = (uint32_t)h.top.size(); info[2] = (uint32_t)h.nodes.size(); info[3] = h.n_members;
    info[4] = h.n_direct; info[5] = h.direct_first;
    for (uint32_t k = 0; k < mrt::kMaxLevels; k++) info[6 + k] = h.level_base[k];
    if ((top_out && top_cap < h.top.size()) || (nodes_out && nodes_cap < h.nodes.size()) ||
        (member_index_out && member_cap < h.member_index.size()) || (mfma_out && mfma_cap < mf.size()))
        return MRT_ERR_TOO_SMALL;
    if (top_out) std::memcpy(top_out, h.top.data(), h.top.size() * sizeof(mrt::SphereRec));
    if (nodes_out) std::memcpy(nodes_out, h.nodes.data(), h.nodes.size() * sizeof(mrt::SphereRec));
    if (member_index_out) std::memcpy(member_index_out, h.member_index.data(), h.member_index.size() * sizeof(uint32_t));
    if (mfma_out) std::memcpy(mfma_out, mf.data(), mf.size() * sizeof(uint16_t));
    if (mfma_origin_out) for (int k = 0; k < 3; k++) mfma_origin_out[k] = origin[k];
    return MRT_OK;
}

int mrt_debug_build_boxes_top_down(const mrt_sphere* spheres, size_t n, uint32_t max_levels, uint32_t top_target, int open,
                                   float* boxes_out, size_t boxes_cap, uint32_t info[5]) {
    if ((!spheres && n) || !info || max_levels < 1 || max_levels > mrt::kMaxLevels || n > mrt::kMaxSpheres)
        return MRT_ERR_INVALID_ARG;
    std::vector<float> centers(4 * (n ? n : 1)), radii(n ? n : 1);
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) centers[4 * i + k] = spheres[i].center[k];
        centers[4 * i + 3] = 1.0f;
        radii[i] = spheres[i].radius;
    }
    Hierarchy h;
    build_hierarchy(centers.data(), radii.data(), (uint32_t)n, 8.0f, max_levels, top_target, h);
    std::vector<mrt::BoxFull> full;
    std::vector<mrt::BoxRec> packed;
    uint32_t cf = 0, cpf = 0;
    boxes_top_down(h, open != 0, full, &cf, &cpf);
    pack_boxes(full, packed);
    // what the kernel reads, in the 8-float form of mrt_debug_build_boxes: centre, the extents WITH kpad folded in, the scene's kc, 0
    std::vector<mrt::BoxFull> dev(full.size());
    for (size_t i = 0; i < full.size(); i++)
        dev[i] = mrt::BoxFull{packed[i].cx, packed[i].cy, packed[i].cz, packed[i].ex, packed[i].ey, packed[i].ez, full[i].ex >= 0.0f ? h.box_kc : 0.0f, 0.0f};
    info[0] = h.levels; info[1] = (uint32_t)dev.size(); info[2] = (uint32_t)h.top.size(); info[3] = cf; info[4] = cpf;
    if (boxes_out && boxes_cap < dev.size()) return MRT_ERR_TOO_SMALL;
    if (boxes_out) std::memcpy(boxes_out, dev.data(), dev.size() * sizeof(mrt::BoxFull));
    return MRT_OK;
}

int mrt_debug_build_boxes(const mrt_sphere* spheres, size_t n, uint32_t max_levels, uint32_t top_target, float* boxes_out,
                          size_t boxes_cap, uint32_t info[8]) {
    if ((!spheres && n) || !info || max_levels < 1 || max_levels > mrt::kMaxLevels || n > mrt::kMaxSpheres)
        return MRT_ERR_INVALID_ARG;
    std::vector<float> centers(4 * (n ? n : 1)), radii(n ? n : 1);
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) centers[4 * i + k] = spheres[i].center[k];
        centers[4 * i + 3] = 1.0f;
        radii[i] = spheres[i].radius;
    }
    Hierarchy h;
    build_hierarchy(centers.data(), radii.data(), (uint32_t)n, 8.0f, max_levels, top_target, h);
    info[0] = h.levels; info[1] = (uint32_t)h.boxes.size(); info[2] = h.box_quad ? 1u : 0u;
    for (uint32_t k = 0; k <= mrt::kMaxLevels; k++) info[3 + k] = h.box_base[k];
    if (boxes_out && boxes_cap < h.boxes.size()) return MRT_ERR_TOO_SMALL;
    if (boxes_out) std::memcpy(boxes_out, h.boxes.data(), h.boxes.size() * sizeof(mrt::BoxFull));
    return MRT_OK;
}

int mrt_debug_world_hit(mrt_ctx* c, const float* rays, size_t n, int32_t* hit_out, uint32_t* cand_out, size_t cand_words) {
    if (!c || !rays || !hit_out || n == 0) return MRT_ERR_INVALID_ARG;
    if (!c->have_world) return fail(c, MRT_ERR_NO_SCENE, "mrt_debug_world_hit: no scene");
    const size_t need_words = ((size_t)c->n_spheres + 31) / 32;
    if (cand_out && cand_words < need_words) return fail(c, MRT_ERR_TOO_SMALL, "mrt_debug_world_hit: need %zu bitmap words per ray", need_words);
    if (n > (1u << 26)) return fail(c, MRT_ERR_INVALID_ARG, "mrt_debug_world_hit: too many rays");
    for (size_t i = 0; i < 6 * n; i++)
        if (!(std::fabs(rays[i]) <= 2.0e7f)) return fail(c, MRT_ERR_INVALID_ARG, "mrt_debug_world_hit: ray %zu is not finite or beyond 2e7", i / 6);
    HIP_TRY(c, hipSetDevice(c->device));
    MRT_TRY(mrt::wait_all(c, __func__));
    // rays become the texels of an 8-pixel-wide virtual image (one 8x8 tile per 64 rays), padded with copies of ray 0
    const size_t n_pad = (n + 63) / 64 * 64;
    const size_t words = need_words ? need_words : 1;
    std::vector<float> host_rays(6 * n_pad);
    std::memcpy(host_rays.data(), rays, 6 * n * sizeof(float));
    for (size_t i = n; i < n_pad; i++) std::memcpy(host_rays.data() + 6 * i, rays, 6 * sizeof(float));
    float* d_rays = nullptr; int32_t* d_hit = nullptr; uint32_t* d_cand = nullptr; uint32_t* d_queue = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_rays); (void)hipFree(d_hit); (void)hipFree(d_cand); (void)hipFree(d_queue); };
    hipError_t e = hipMalloc((void**)&d_rays, host_rays.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&d_hit, n_pad * 2 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_cand, n_pad * words * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_queue, 64);
    if (e == hipSuccess) e = hipMemcpy(d_rays, host_rays.data(), host_rays.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_cand, 0, n_pad * words * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(d_hit, 0xFF, n_pad * 2 * sizeof(int32_t));
    if (e != hipSuccess) { cleanup(); return fail(c, MRT_ERR_HIP, "mrt_debug_world_hit: %s", hipGetErrorString(e)); }
    mrt::KParams p;
    std::memset(&p, 0, sizeof p);
    p.locals = c->locals;
    p.locals.shape[0] = 8; p.locals.shape[1] = (uint32_t)(n_pad / 8);
    p.locals.samples_per_frame = 1; p.locals.ray_depth = 1;
    fill_scene_params(c, p);
    p.shard_rank = 0; p.shard_world = 1;
    p.tiles_x = 1; p.n_tiles = (uint32_t)(n_pad / 64);
    p.tile_queue = d_queue;
    p.n_blocks = 1; p.pix_stride = 0; p.queue_layers = 1; p.lane_frames = 1;
    p.dbg_rays = d_rays; p.dbg_hit = d_hit; p.dbg_cand = d_cand; p.dbg_words = (uint32_t)words;
    int le = mrt::launch_debug_world_hit(p, c->n_waves, c->stream);
    int ws = MRT_OK;
    if (le == 0) ws = mrt::wait_stream(c, c->stream, "mrt_debug_world_hit");
    if (ws != MRT_OK) { cleanup(); return ws; }
    if (le != 0 || e != hipSuccess) { cleanup(); return fail(c, MRT_ERR_HIP, "mrt_debug_world_hit: launch failed: %s", hipGetErrorString(le ? (hipError_t)le : e)); }
    std::vector<int32_t> hits(n_pad * 2);
    e = hipMemcpy(hits.data(), d_hit, hits.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) std::memcpy(hit_out, hits.data(), n * 2 * sizeof(int32_t));
    if (e == hipSuccess && cand_out) {
        std::vector<uint32_t> cand(n_pad * words);
        e = hipMemcpy(cand.data(), d_cand, cand.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (size_t i = 0; i < n; i++) {
                std::memset(cand_out + i * cand_words, 0, cand_words * sizeof(uint32_t));
                std::memcpy(cand_out + i * cand_words, cand.data() + i * words, need_words * sizeof(uint32_t));
            }
    }
    cleanup();
    if (e != hipSuccess) return fail(c, MRT_ERR_HIP, "mrt_debug_world_hit: read-back failed: %s", hipGetErrorString(e));
    return MRT_OK;
}

int mrt_debug_last_launch(mrt_ctx* c, uint32_t out[2]) {
    if (!c || !out) return MRT_ERR_INVALID_ARG;
    out[0] = c->last_launch[0]; out[1] = c->last_launch[1];
    c->last_launch[1] = 0xFFFFFFFFu;            // a pilot launch is reported once
    return MRT_OK;
}

int mrt_debug_set_boxes(mrt_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 2) return MRT_ERR_INVALID_ARG;
    c->boxes_mode = mode;
    return MRT_OK;
}

int mrt_debug_sweep_variant(mrt_ctx* c) {
    if (!c || !c->have_world) return 0;
    return use_matrix_core_sweep(c) ? 2 : 1;
}

int mrt_debug_mfma_scale(double reach, float scale_out[4], uint32_t* neg_k2_bf16_pair_out) {
    if (!scale_out || !neg_k2_bf16_pair_out || !(reach >= 0.0) || !std::isfinite(reach)) return MRT_ERR_INVALID_ARG;
    mfma_scales(reach, scale_out, neg_k2_bf16_pair_out);
    return MRT_OK;
}

int mrt_debug_set_hierarchy(mrt_ctx* c, uint32_t max_levels, uint32_t top_target) {
    if (!c || max_levels < 1 || max_levels > mrt::kMaxLevels) return MRT_ERR_INVALID_ARG;      // top_target 0 = automatic
    c->max_levels = max_levels;
    c->top_target = top_target;
    return MRT_OK;
}

// diagnostic / A-B switch: 0 = tile queue in index order instead of heaviest-first
int mrt_debug_set_tile_sort(mrt_ctx* c, int enabled) {
    if (!c) return MRT_ERR_INVALID_ARG;
    c->lpt_enabled = enabled != 0;
    return MRT_OK;
}

// diagnostic / tuning: pilot spp, waves per CU (0 = automatic).
// Call before rendering.
int mrt_debug_set_schedule(mrt_ctx* c, uint32_t pilot_spp, int waves_per_cu) {
    if (!c) return MRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    MRT_TRY(mrt::wait_all(c, __func__));
    c->pilot_spp = pilot_spp ? pilot_spp : 1;
    c->waves_per_cu_override = waves_per_cu;
    const uint32_t frames = c->frames_done;
    if (frames != 0) return fail(c, MRT_ERR_STATE, "mrt_debug_set_schedule: frames already rendered");
    return alloc_frame_buffers(c);
}

int mrt_debug_lds_layout(uint32_t n_members, uint32_t n_nodes, uint32_t levels, uint32_t n_top_padded, uint32_t out[3]) {
    if (!out || levels < 1 || levels > mrt::kMaxLevels) return MRT_ERR_INVALID_ARG;
    mrt::KParams p;
    std::memset(&p, 0, sizeof p);
    p.n_members = n_members; p.n_nodes = n_nodes; p.levels = levels; p.n_padded = n_top_padded;
    { const uint32_t ch = (n_top_padded + mrt::kChunk - 1) / mrt::kChunk; p.mask_chunks = ch < 16u ? ch : 16u; }
    p.box_lds_count = n_members <= 1024u ? 0u : mrt::large_scene_box_lds_count(n_top_padded, levels, p.mask_chunks, mrt::kBoxLdsCap);
    p.gen_cap = n_members <= 1024u ? 576u : mrt::large_scene_stack_cap(p.mask_chunks, p.box_lds_count);
    uint32_t lay[2];
    mrt::render_lds_layout(p, lay);
    out[0] = lay[0]; out[1] = lay[1]; out[2] = p.gen_cap;
    return MRT_OK;
}

int mrt_set_wait_timeout(mrt_ctx* c, double seconds) {
    if (!c || !(seconds >= 0.0) || !std::isfinite(seconds)) return MRT_ERR_INVALID_ARG;
    c->wait_timeout_s = seconds;
    return MRT_OK;
}

int mrt_get_schedule(mrt_ctx* c, uint32_t out[6]) {
    if (!c || !out) return MRT_ERR_INVALID_ARG;
    out[0] = c->width.div; out[1] = c->width.mult; out[2] = c->width.settled;
    out[3] = c->width.div ? mrt::width_frames_in_flight(c->width.div, c->width.mult, c->max_slots) : c->frame_slots;
    out[4] = c->last_launch_div;
    out[5] = c->slots_probed ? c->max_slots : 0u;        // 0 = not measured yet (no setting has asked for more than two frames)
    return MRT_OK;
}

int mrt_set_schedule_hint(mrt_ctx* c, uint32_t div, uint32_t mult) {
    if (!c) return MRT_ERR_INVALID_ARG;
    if (div == 0 && mult == 0) {
        c->hint_div = c->hint_mult = 0;
    } else {
        if (div < 1 || div > mrt::kMaxWidthDiv || mult < 1 || mult > 8 || std::max(2u, div) * mult > mrt_ctx::kMaxFrameSlots)
            return fail(c, MRT_ERR_INVALID_ARG, "mrt_set_schedule_hint: div %u x mult %u (div 1..8, mult 1..8, max(2, div) x mult <= 16)", div, mult);
        c->hint_div = div; c->hint_mult = mult;
    }
    c->width.div = 0;                   // the next redraw takes the hint (or starts measuring again)
    return MRT_OK;
}

int mrt_debug_width_policy(int op, const uint32_t workload[6], uint32_t state[7], double util, double rate) {
    if (!workload || !state || op < 0 || op > 2) return MRT_ERR_INVALID_ARG;
    mrt::WidthWorkload w;
    w.n_tiles = workload[0]; w.n_waves = workload[1]; w.max_slots = workload[2]; w.spp = workload[3]; w.n_members = workload[4]; w.counter = workload[5];
    mrt::WidthState s;
    float pr;
    std::memcpy(&pr, &state[6], 4);
    s.div = state[0]; s.mult = state[1]; s.prev_div = state[2]; s.prev_mult = state[3]; s.low_windows = state[4]; s.settled = state[5]; s.prev_rate = pr;
    if (op == 0) mrt::width_policy_start(s, w);
    else if (op == 1) { mrt::WidthWindow m; m.util = util; m.rate = rate; mrt::width_policy_step(s, w, m); }
    else { state[0] = mrt::width_launch_div(s.div, (uint32_t)util); return MRT_OK; }
    pr = (float)s.prev_rate;
    state[0] = s.div; state[1] = s.mult; state[2] = s.prev_div; state[3] = s.prev_mult; state[4] = s.low_windows; state[5] = s.settled;
    std::memcpy(&state[6], &pr, 4);
    return MRT_OK;
}

int mrt_debug_stream_concurrency(mrt_ctx* c, uint32_t streams, float* out) {
    if (!c || !out || streams < 2 || streams > mrt_ctx::kMaxFrameSlots) return MRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    return probe_stream_concurrency(c, streams, out);
}

int mrt_debug_set_frames_in_flight(mrt_ctx* c, int slots) {
    if (!c || slots < 0 || slots > (int)mrt_ctx::kMaxFrameSlots) return MRT_ERR_INVALID_ARG;
    c->frame_slots_override = slots;
    return MRT_OK;
}

// div_unscaled / sqrt_unscaled (kernels.hip) against hipcc's `/` and sqrtf(), on the device, over whole operand ranges
int mrt_debug_arith(mrt_ctx* c, int mode, const uint32_t bits_range[4], uint64_t count, uint64_t seed, uint64_t out[3]) {
    if (!c || !bits_range || !out || mode < 0 || mode > 2) return MRT_ERR_INVALID_ARG;
    if (bits_range[0] > bits_range[1] || (mode != 0 && bits_range[2] > bits_range[3]))
        return fail(c, MRT_ERR_INVALID_ARG, "mrt_debug_arith: empty range");
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long* d_out = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d_out, 3 * sizeof(unsigned long long)));
    const unsigned long long init[3] = {0ull, 0ull, ~0ull};
    hipError_t e = hipMemcpyAsync(d_out, init, sizeof init, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = (hipError_t)mrt::launch_arith_check(mode, bits_range, count, seed, d_out, c->stream);
    unsigned long long host[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(host, d_out, sizeof host, hipMemcpyDeviceToHost, c->stream);
    int ws = MRT_OK;
    if (e == hipSuccess) ws = mrt::wait_stream(c, c->stream, "mrt_debug_arith");
    if (ws != MRT_OK) return ws;            // (stalled: d_out is left to the process)
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, MRT_ERR_HIP, "mrt_debug_arith failed: %s", hipGetErrorString(e));
    for (int k = 0; k < 3; k++) out[k] = host[k];
    return MRT_OK;
}

int mrt_debug_arith_pairs(mrt_ctx* c, const float* x, const float* y, size_t n, uint32_t* out) {
    if (!c || !x || !y || !out || n > (1u << 28)) return MRT_ERR_INVALID_ARG;
    if (n == 0) return MRT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    float* d_xy = nullptr;
    uint32_t* d_o = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d_xy, 2 * n * sizeof(float)));
    hipError_t e = hipMalloc((void**)&d_o, 6 * n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpyAsync(d_xy, x, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_xy + n, y, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = (hipError_t)mrt::launch_arith_pairs(d_xy, d_xy + n, (uint32_t)n, d_o, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_o, 6 * n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    int ws = MRT_OK;
    if (e == hipSuccess) ws = mrt::wait_stream(c, c->stream, "mrt_debug_arith_pairs");
    if (ws != MRT_OK) return ws;
    (void)hipFree(d_xy);
    if (d_o) (void)hipFree(d_o);
    if (e != hipSuccess) return fail(c, MRT_ERR_HIP, "mrt_debug_arith_pairs failed: %s", hipGetErrorString(e));
    return MRT_OK;
}

int mrt_sync(mrt_ctx* c) {
    if (!c) return MRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    MRT_TRY(mrt::wait_all(c, __func__));
    return MRT_OK;
}

int mrt_reset(mrt_ctx* c) {
    if (!c) return MRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    MRT_TRY(mrt::wait_all(c, __func__));
    const size_t bytes = local_texels(c) * 4 * sizeof(float);
    HIP_TRY(c, hipMemsetAsync(c->d_fb[0], 0, bytes, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_fb[1], 0, bytes, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, 16 * sizeof(unsigned long long), c->stream));
    for (auto& S : c->slot)          // the tile queues' counters (normally left at zero by every finalize pass)
        if (S.d_sort_scratch) HIP_TRY(c, hipMemsetAsync(S.d_sort_scratch + 1024, 0, sizeof(uint32_t), c->stream));
    c->inputs_dirty = true;      // the next redraw's side stream waits for these memsets (ev_inputs)
    const uint32_t spp = c->locals.samples_per_frame, mode = c->locals.rng_mode;
    reset_locals(c);
    c->locals.samples_per_frame = spp;
    c->locals.rng_mode = mode;
    c->target = 0;
    return MRT_OK;
}

int mrt_get_locals(mrt_ctx* c, mrt_locals* out) {
    if (!c || !out) return MRT_ERR_INVALID_ARG;
    *out = c->locals;
    return MRT_OK;
}

int mrt_set_rng_shuffle(mrt_ctx* c, const uint32_t s[4]) {
    if (!c || !s) return MRT_ERR_INVALID_ARG;
    std::memcpy(c->locals.rng_shuffle, s, 16);
    c->shuffle_overridden = true;          // the next frame is rendered on its own (mrt_render does not batch it)
    return MRT_OK;
}

int mrt_set_rng_mode(mrt_ctx* c, uint32_t mode) {
    if (!c || mode > MRT_RNG_COUNTER) return MRT_ERR_INVALID_ARG;
    c->locals.rng_mode = mode;
    c->width.div = 0;
    return MRT_OK;
}

int mrt_set_draw_counting(mrt_ctx* c, int enabled) {
    if (!c) return MRT_ERR_INVALID_ARG;
    c->count_draws = enabled != 0;
    return MRT_OK;
}

int mrt_set_samples_per_frame(mrt_ctx* c, uint32_t spp) {
    if (!c) return MRT_ERR_INVALID_ARG;
    c->locals.samples_per_frame = spp;
    c->width.div = 0;
    return MRT_OK;
}

uint32_t mrt_frames_done(mrt_ctx* c) { return c ? c->frames_done : 0; }

void* mrt_framebuffer_device_ptr(mrt_ctx* c) {
    if (!c) return nullptr;
    return c->d_fb[c->target ^ 1];   // after the swap, the last render target is "secondary"
}

int mrt_read_framebuffer(mrt_ctx* c, float* out, size_t cap) {
    if (!c || !out) return MRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    float* src = c->d_fb[c->target ^ 1];
    if (c->shard_world == 1) {
        const size_t n = (size_t)c->args.width * c->args.height * 4;
        if (cap < n) return fail(c, MRT_ERR_TOO_SMALL, "mrt_read_framebuffer: need %zu floats", n);
        return copy_rows(c, src, out, 16, false);
    }
    const size_t n = local_texels(c) * 4;
    if (cap < n) return fail(c, MRT_ERR_TOO_SMALL, "mrt_read_framebuffer: need %zu floats", n);
    HIP_TRY(c, hipMemcpyAsync(out, src, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    MRT_TRY(mrt::wait_stream(c, c->stream, __func__));
    return MRT_OK;
}

int mrt_read_counters(mrt_ctx* c, mrt_counters* out) {
    if (!c || !out) return MRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long h[5];
    HIP_TRY(c, hipMemcpyAsync(h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    MRT_TRY(mrt::wait_stream(c, c->stream, __func__));
    out->samples = h[0]; out->world_hit_calls = h[1]; out->rng_draws = h[2]; out->lane_slots = h[3];
    out->member_tests = h[4];
    out->sweep_records = c->n_padded;
    return MRT_OK;
}

// diagnostic: all 16 raw counter slots (slots 4.. are only written by -DMRT_STAMPS builds)
int mrt_debug_read_counters(mrt_ctx* c, uint64_t out[16]) {
    if (!c || !out) return MRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(out, c->d_counters, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    MRT_TRY(mrt::wait_stream(c, c->stream, __func__));
    return MRT_OK;
}

// diagnostic: enable / read the per-wave log {t_start, t_end (100 MHz ticks), trips, bounces}
// that -DMRT_STAMPS builds write; out == NULL just (re)allocates it for the current shard.  The log is a ring over the last
// kWaveLogFrames frames (frames in flight overlap): `back` = 0 reads the most recent frame's, 1 the one before, ...
int mrt_debug_wave_log_frame(mrt_ctx* c, uint32_t back, uint64_t* out, size_t cap_waves, size_t* n_waves) {
    if (!c || back >= mrt_ctx::kWaveLogFrames) return MRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = c->n_waves;
    if (n_waves) *n_waves = n;
    if (!c->d_wave_log || c->wave_log_waves != n) {
        MRT_TRY(mrt::wait_all(c, __func__));
        if (c->d_wave_log) (void)hipFree(c->d_wave_log);
        c->d_wave_log = nullptr;
        HIP_TRY(c, hipMalloc(&c->d_wave_log, mrt_ctx::kWaveLogFrames * n * 4 * sizeof(uint64_t)));
        HIP_TRY(c, hipMemset(c->d_wave_log, 0, mrt_ctx::kWaveLogFrames * n * 4 * sizeof(uint64_t)));
        c->wave_log_waves = n;
    }
    if (out) {
        if (cap_waves < n) return fail(c, MRT_ERR_TOO_SMALL, "mrt_debug_wave_log: need %zu waves", n);
        if (c->frame_seq <= back) return fail(c, MRT_ERR_STATE, "mrt_debug_wave_log: frame %u back has not been rendered", back);
        MRT_TRY(mrt::wait_all(c, __func__));
        const unsigned long long* src = c->d_wave_log + (size_t)((c->frame_seq - 1 - back) % mrt_ctx::kWaveLogFrames) * n * 4;
        HIP_TRY(c, hipMemcpyAsync(out, src, n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        MRT_TRY(mrt::wait_stream(c, c->stream, __func__));
    }
    return MRT_OK;
}
int mrt_debug_wave_log(mrt_ctx* c, uint64_t* out, size_t cap_waves, size_t* n_waves) {
    return mrt_debug_wave_log_frame(c, 0, out, cap_waves, n_waves);
}

int mrt_kernel_ms_history(mrt_ctx* c, float* ms, size_t cap, size_t* n_out) {
    if (!c || !ms || !n_out) return MRT_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    size_t n = c->frame_seq < mrt_ctx::kEventRing ? (size_t)c->frame_seq : mrt_ctx::kEventRing;
    if (n > cap) n = cap;
    for (size_t i = 0; i < n; i++) {          // ms[0] = oldest of the n most recent redraws
        const uint32_t slot = (uint32_t)((c->frame_seq - n + i) % mrt_ctx::kEventRing);
        MRT_TRY(mrt::wait_event(c, c->ev_stop[slot], "mrt_kernel_ms_history: the render kernel's stop event"));
        HIP_TRY(c, hipEventElapsedTime(&ms[i], c->ev_start[slot], c->ev_stop[slot]));
    }
    *n_out = n;
    return MRT_OK;
}

int mrt_last_kernel_ms(mrt_ctx* c, float* ms) {
    if (!c || !ms) return MRT_ERR_INVALID_ARG;
    if (!c->frame_seq) return fail(c, MRT_ERR_STATE, "mrt_last_kernel_ms: nothing rendered yet");
    size_t n = 0;
    return mrt_kernel_ms_history(c, ms, 1, &n);
}

}  // extern "C"
