// Host-side state behind the opaque mrt_ctx of include/myraytracer_amd.h, shared by api.cpp (the frame loop)
// and multi_gpu.cpp (the gather).  Internal: not installed.
#pragma once
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <string>
#include <vector>

#include "mrt_internal.h"
#include "width_policy.h"

struct mrt_ctx {
    int device = 0;
    mrt_args args{};
    uint64_t seed = 0;
    mrt_locals locals{};
    uint32_t frames_done = 0;          // State::sample_count (lib.rs:213, 300)

    uint32_t shard_rank = 0, shard_world = 1;
    uint32_t local_bands = 0;

    mrt_world world{};
    bool have_world = false;
    uint32_t n_spheres = 0, n_padded = 0;
    mrt_camera_raw cam_raw{};

    // device memory (all owned)
    mrt::SphereRec* d_spheres = nullptr;
    mrt::SphereRec* d_clusters = nullptr;  // bounding spheres the sweep tests (up to kClusterK spheres each)
    uint16_t* d_top_mfma = nullptr;        // the top-level records as the MFMA A operand (build_top_mfma)
    bool mfma_scene_ok = false;            // the expanded test's extra slack is negligible for this scene
    double mfma_r2_ref = 0.0;              // median R^2 of the top level (camera check at launch)
    float mfma_origin[3] = {0.0f, 0.0f, 0.0f};   // the matrix-core sweep works in coordinates relative to this point
    double mfma_reach = 0.0;                      // max over ALL spheres of |centre - mfma_origin| + |radius|: no hit point lies further out
    int sweep_mode = 0;                    // 0 automatic, 1 SGPR-fed VALU sweep, 2 matrix-core sweep (mrt_debug_set_sweep)
    float* d_shade = nullptr;              // 8 floats per sphere: centre, radius, material colour, fuzz | ior
    mrt::SphereRec* d_nodes = nullptr;     // hierarchy levels below the top: members (kClusterK per cluster), clusters, ...
    uint32_t* d_member_index = nullptr;    // their indices in the reference's sphere order
    // large scenes' walk: the axis-aligned boxes of the hierarchy's nodes in the kernel's top-down numbering (KParams::boxes),
    // and the same array with every real box opened wide (a box test that never rejects: mrt_debug_set_boxes(0))
    mrt::BoxRec* d_boxes = nullptr;
    mrt::BoxRec* d_boxes_open = nullptr;
    uint32_t box_cluster_first = 0, box_cluster_parent_first = 0;
    bool box_quad = false;
    float box_kc = 0.0f;
    int boxes_mode = 1;                    // mrt_debug_set_boxes: 0 = the boxes never reject (diagnostic), 1 / 2 = they do
    float cluster_factor = 8.0f;           // grow a cluster while its enclosing radius <= factor * largest member radius
    // hierarchy depth rule (build_hierarchy): levels are added while the top has more than top_target records; 0 = automatic
    // (256, or 128 for scenes whose walk tests boxes)
    uint32_t max_levels = mrt::kMaxLevels, top_target = 0;
    uint32_t levels = 1, n_nodes = 0, n_members = 0;
    uint32_t level_base[mrt::kMaxLevels] = {0, 0, 0, 0};
    uint32_t n_direct = 0, direct_first = 0;
    mrt::SphereRec direct[mrt::kMaxDirect] = {};
    uint32_t direct_index[mrt::kMaxDirect] = {};
    float* d_vec4 = nullptr;
    float* d_f32 = nullptr;
    int32_t* d_i32 = nullptr;
    uint32_t* d_seeds = nullptr;
    float* d_fb[2] = {nullptr, nullptr};   // [target, secondary] ping-pong (lib.rs:505-543)
    int target = 0;                        // index of the buffer the NEXT redraw writes
    unsigned long long* d_counters = nullptr;
    bool count_draws = true;               // mrt_set_draw_counting
    uint32_t last_launch[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};   // mrt_debug_last_launch: the last render / pilot instantiation
    // Up to frame_slots frames may be in flight: frame n's render kernel (sort, pilot) runs on side
    // stream n % frame_slots and only its finalize pass -- the one step that needs frame n-1's framebuffer -- runs
    // on the caller's stream.  The next frame's heavy tiles thus start while this frame's last
    // pixels drain (a pixel is one sequential chain, so every frame ends on a thinning chip).
    // How many: 2 for launches that fill the chip (re-measured in round 2, LDS-free sort: C3 9,799 / 9,709 / 9,380 Msamples/s
    // with 2 / 3 / 4).  A PIXEL-STARVED shard (fewer than two pixels per resident lane, long sample chains: an 8-GPU share of
    // C5) is different: its launch is as long as its heaviest pixel's chain while most of its waves end much earlier, and a
    // wave instruction costs the same with 40 % of its lanes active as with all -- so it runs more frames at once, each on
    // fewer, better packed waves (redraw_frames; round 4).  frame_slots is the count in use, slot[] the capacity.
    static constexpr uint32_t kMaxFrameSlots = 16;
    uint32_t frame_slots = 2;
    uint32_t last_slot = 0;                         // the slot of the most recent redraw
    int frame_slots_override = 0;                   // mrt_debug_set_frames_in_flight: 0 = automatic
    struct FrameSlot {
        hipStream_t stream = nullptr;
        hipEvent_t render_done = nullptr, finalize_done = nullptr;
        void* d_pix_acc = nullptr;             // per-pixel (x per-block, counter mode) colour sums + costs, render -> finalize
        size_t pix_acc_layers = 0;             // capacity in layers of local_texels entries
        uint32_t cost_first_layer = 0, cost_layers = 1;   // the layers holding the slot's most recent frame
        uint32_t* d_tile_cost = nullptr;       // written by this slot's finalize, orders its next queue
        uint32_t* d_tile_order = nullptr;
        uint32_t* d_sort_scratch = nullptr;    // 1024 u32 of sort workspace + the queue counter
        bool cost_valid = false;               // d_tile_cost holds a usable estimate for the current scene
        // The tile queue's counter is left at zero by every finalize pass of the slot.  If anything between a render launch
        // and its last finalize launch fails, it is not: the next launch on this slot resets it itself.
        bool queue_dirty = false;
        // launch-width controller: the context's cumulative {world_hit calls, lane slots} copied to pinned host memory right
        // after this slot's render kernel (h_stats[3 slot ..]: counters 1 .. 3 in one copy), the event that says the copy has landed, the frame it was
        hipEvent_t stats_ready = nullptr;
        uint64_t stats_seq = 0;
        bool stats_pending = false;
        bool render_pending = false;           // a render kernel of this slot has been launched and not yet been seen complete
        uint64_t render_seq = 0;               // its frame (diagnostics of a stalled wait)
    } slot[kMaxFrameSlots];
    // Launch width (redraw_frames; the policy itself: width_policy.h): a frame is launched on 1 / width.div of the persistent
    // waves the chip holds and max(2, width.div) x width.mult frames are in flight, so that the chip stays full.  Narrow launches
    // pack the lanes better (more pixels per lane in sequence: the launch's tail, in which lanes idle until their wave's longest
    // pixel ends, is the same length but a smaller share) at the price of a longer frame latency -- and they do not always pay
    // (C3 / C4, 0.98 / 0.99 lane utilisation at full width, lose 1-4 % at a half; C2 loses 4 % at a half and gains 18 % at a
    // quarter).  So the setting is MEASURED: trials while the utilisation is low, kept only if the frame rate rises by 3 %.
    // Scheduling only: the images do not change.
    mrt::WidthState width;                          // div 0 = not chosen yet for the current workload
    uint32_t hint_div = 0, hint_mult = 0;           // mrt_set_schedule_hint: the caller's setting (0 = the controller decides)
    uint32_t max_slots = kMaxFrameSlots;            // frames that can really run side by side (probe_stream_concurrency)
    bool slots_probed = false;
    uint32_t last_launch_div = 1, last_frames_running = 0;   // mrt_get_schedule: what the most recent launch was issued with
    uint32_t running_seen[kMaxFrameSlots] = {}, running_seen_n = 0;    // frames seen queued or running at the last calls (schedule_frame)
    uint32_t nothing_running_calls = 0;             // consecutive calls that found no earlier frame queued or running
    // settled settings by workload, so that a change of camera / samples per frame / scene and back does not start the trials
    // over (and a viewer that moves its camera every frame still reaches one)
    struct WidthMemo { uint32_t n_tiles, spp, large, counter, n_spheres, div, mult; };
    std::vector<WidthMemo> width_memo;
    uint64_t width_valid_from = 0;                  // frame_seq from which samples and timings belong to the current setting
    bool width_timing = false;                      // a measurement window is open: since frame width_t0_seq, at width_t0
    uint64_t width_t0_seq = 0;
    std::chrono::steady_clock::time_point width_t0;
    struct LaneStat { uint64_t seq = 0, hits = 0, slots = 0; bool valid = false; } stat_base, stat_last;
    unsigned long long* h_stats = nullptr;          // pinned, 3 x kMaxFrameSlots (+ 2 x kMaxFrameSlots: the concurrency probe's stamps)
    hipEvent_t ev_inputs = nullptr;            // scene / seeds uploads on the caller's stream
    bool inputs_dirty = true;
    uint64_t frame_seq = 0;
    uint32_t tiles_x = 0, n_tiles = 0, n_waves = 0, cus = 0;
    uint32_t pilot_spp = 2;
    int waves_per_cu_override = 0;
    bool lpt_enabled = true;
    static constexpr uint32_t kWaveLogFrames = 32;
    unsigned long long* d_wave_log = nullptr;   // diagnostic, see mrt_debug_wave_log: a ring of kWaveLogFrames frames' logs
    size_t wave_log_waves = 0;

    // multi-GPU gather (multi_gpu.cpp): on the root, the full frame assembled from every shard's bands
    // (total_bands_padded x 8 rows x W RGBA32F, row 0 = bottom) and, for the RCCL variant, the rank-major
    // receive staging; ev_gather marks "this shard's bands have been copied out" on its stream
    float* d_gather = nullptr;
    float* d_gather_stage = nullptr;
    size_t gather_bytes = 0, gather_stage_bytes = 0;
    hipEvent_t ev_gather = nullptr;
    // on the root: "everything queued on the root's stream before this gather" -- a reader of the previous frame's
    // d_gather among it -- which every shard's stream waits for before it overwrites d_gather (write-after-read)
    hipEvent_t ev_gather_root = nullptr;
    bool gather_per_band = false;              // mrt_debug_set_gather_per_band: the cross-device copy loop on one device

    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // ring of HIP event pairs around the render kernel of the most recent redraws, frame f at f % kEventRing
    static constexpr uint32_t kEventRing = 64;
    hipEvent_t ev_start[kEventRing] = {}, ev_stop[kEventRing] = {};

    bool shuffle_overridden = false;       // mrt_set_rng_shuffle since the last frame
    bool batch_frames = true;              // mrt_render may render several frames per launch (mrt_debug_set_frame_batching)
    int batch_form = 0;                    // 0 automatic, 1 always "a lane keeps its pixel for the batch's frames", 2 always queue layers
    float set_world_ms = 0.0f;             // host time of the last scene upload (hierarchy build + copies)

    // Every blocking host wait of the library polls with this deadline (seconds; 0 = no deadline) and fails with
    // MRT_ERR_STALLED, naming the wait, instead of hanging (mrt_set_wait_timeout).  A context that has stalled once stays
    // failed: its queued work may never finish, so mrt_destroy then releases what it can without waiting.
    double wait_timeout_s = 120.0;
    bool stalled = false;

    std::string err;
};

namespace mrt {

// records the message behind mrt_last_error (ctx == NULL: the thread's global message) and returns `status`
int fail(mrt_ctx* ctx, int status, const char* fmt, ...) __attribute__((format(printf, 3, 4)));

inline uint32_t total_bands(uint32_t height) { return (height + kBandRows - 1) / kBandRows; }
inline size_t local_texels(const mrt_ctx* c) { return (size_t)c->local_bands * kBandRows * c->args.width; }

// Bounded host waits (api.cpp): poll the event / stream until it is complete or the context's deadline has passed; `what`
// names the wait in the error message.  Return an mrt_status.
int wait_event(mrt_ctx* c, hipEvent_t ev, const char* what);
int wait_stream(mrt_ctx* c, hipStream_t s, const char* what);
// everything this context has in flight: the side streams, then the caller's stream
int wait_all(mrt_ctx* c, const char* what);

}  // namespace mrt

#define MRT_TRY(expr)                      \
    do {                                   \
        const int st_ = (expr);            \
        if (st_ != MRT_OK) return st_;     \
    } while (0)

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return mrt::fail(ctx, MRT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
