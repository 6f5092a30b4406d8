/*
 * myraytracer_amd.h -- C ABI of the MI355X-native backend for the per-pixel render loop
 * of zetanumbers/myraytracer.
 *
 * The reference has no FFI / plugin seam for this path: the WGSL shader is
 * include_str!-embedded (raytracer/src/lib.rs:1016) and run by a wgpu draw
 * (lib.rs:262-267).  Each entry point below therefore names the reference item it
 * replaces; INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add to
 * raytracer/src/lib.rs to route `State` through this library.
 *
 * Conventions
 *   - every function returns an mrt_status (0 = OK) unless stated; nothing aborts or
 *     throws across the ABI (the reference panics via expect/unwrap, lib.rs:147,270,...).
 *   - one mrt_ctx = one GPU = one caller thread at a time (the reference's State is
 *     single-threaded and !Send, lib.rs:206-215).
 *   - the caller owns every host array passed in (copied during the call) and every
 *     output buffer; the ctx owns all device memory.
 *   - framebuffer rows are bottom-up: row 0 is the BOTTOM of the picture
 *     (shader.wgsl:26 vs sample_framebuffer.wgsl:24).
 *   - all structs are plain C, 4-byte fields, no padding surprises (static_asserts in
 *     csrc/api.cpp).
 */
#ifndef MYRAYTRACER_AMD_H
#define MYRAYTRACER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: round 4 -- the diagnostic entry points (mrt_debug_*) moved to myraytracer_amd_debug.h; mrt_build_id added; since 2 the
 * library also gained mrt_set_draw_counting and the boxes / frame-batching diagnostics, and mrt_debug_set_hierarchy /
 * mrt_debug_build_hierarchy accept top_target 0 = automatic ("ABI history"). */
/* 4: round 5 -- MRT_ERR_STALLED and mrt_set_wait_timeout (every blocking host wait has a deadline), mrt_set_schedule_hint /
 * mrt_get_schedule (the launch schedule a run settled at can be read and pinned); mrt_create no longer touches the process
 * environment (GPU_MAX_HW_QUEUES is the host's to set: INTEGRATION.md 2a). */
#define MRT_ABI_VERSION 4

typedef enum {
    MRT_OK = 0,
    MRT_ERR_INVALID_ARG = 1,   /* null pointer, zero size, bad enum */
    MRT_ERR_NO_DEVICE = 2,     /* no HIP device / wrong arch: the product never falls back to CPU */
    MRT_ERR_HIP = 3,           /* a HIP runtime call failed; see mrt_last_error */
    MRT_ERR_NO_SCENE = 4,      /* redraw before set_world */
    MRT_ERR_BAD_SCENE = 5,     /* index out of range, non-finite or out-of-range geometry */
    MRT_ERR_TOO_SMALL = 6,     /* caller buffer too small */
    MRT_ERR_STATE = 7,         /* call not allowed in this state (e.g. reshard after first frame) */
    MRT_ERR_IO = 8,
    MRT_ERR_STALLED = 9        /* a wait for the GPU passed its deadline (mrt_set_wait_timeout); mrt_last_error names the wait.
                                  The context stays failed: destroy it (mrt_destroy does not wait for a stalled context) */
} mrt_status;

/* ---- raytracer::Args, lib.rs:18-37; flags of native-runner/src/main.rs:20-31 ---- */
typedef struct {
    uint32_t width;                 /* default 0 */
    uint32_t height;                /* default 0 */
    uint32_t samples_per_frame;     /* default 1 */
    uint32_t ray_depth;             /* default 50 */
    float    max_framebuffer_weight;/* default 1.0 */
} mrt_args;

/* Args::default(), lib.rs:27-37 */
void mrt_args_default(mrt_args* out);
/* Size rule of App::resumed, lib.rs:113-134,149-154: both 0 -> default window size
 * (MRT_DEFAULT_WIDTH x MRT_DEFAULT_HEIGHT here, there is no window); exactly one 0 ->
 * square of the other. */
#define MRT_DEFAULT_WIDTH 800
#define MRT_DEFAULT_HEIGHT 600
void mrt_args_resolve_size(mrt_args* inout);

/* ---- Locals uniform, lib.rs:368-377 / shader.wgsl:8-17 (48 bytes) ---- */
typedef struct {
    uint32_t shape[2];
    uint32_t samples_per_frame;
    uint32_t ray_depth;
    uint32_t rng_shuffle[4];
    float    framebuffer_weight;
    uint32_t rng_mode;          /* the reference's first padding word: 0 = its per-pixel stream (MRT_RNG_*) */
    uint32_t _padding[2];
} mrt_locals;

/* RNG modes.  0 is the reference: one sequential Xoshiro128+ stream per pixel per frame
 * (shader.wgsl:377-382).  1 is an extension (north_star's "counter-based RNG per lane"): every sample
 * starts from a hash of (seed texel ^ rng_shuffle, sample index), so samples are independent of how
 * many draws earlier samples consumed; within a sample the draw order is the reference's. */
enum { MRT_RNG_PIXEL_STREAM = 0, MRT_RNG_COUNTER = 1 };
/* In the counter mode a pixel's colour sum is DEFINED blockwise: S_b = the sequential sum of samples [64 b, 64 b + 64),
 * colour = ((S_0 + S_1) + S_2) ... -- up to 64 samples per frame the same expression as the stream mode's.  Blocks of
 * one pixel may then be rendered by different lanes, which is what keeps a small shard of a many-spp frame (an 8-GPU
 * share of 1920x1080 at 4,096 spp) from starving the GPU. */
#define MRT_COUNTER_BLOCK 64

/* ---- raw::World, lib.rs:641-685 / shader.wgsl:109-124,165-182 (64 bytes) plus the
 *      Dielectric extension appended after MetalRange (80 bytes total) ---- */
typedef struct {
    int32_t center_base_idx, radius_base_idx, material_ty_base_idx, material_idx_base_idx;
    int32_t length, _padding[3];
} mrt_sphere_range;
typedef struct { int32_t albedo_base_idx, length, _padding[2]; } mrt_lambertian_range;
typedef struct { int32_t albedo_base_idx, fuzz_base_idx, length, _padding; } mrt_metal_range;
typedef struct { int32_t ior_base_idx, length, _padding[2]; } mrt_dielectric_range;  /* extension */
typedef struct {
    mrt_sphere_range     spheres;
    mrt_lambertian_range lambertians;
    mrt_metal_range      metals;
    mrt_dielectric_range dielectrics;
} mrt_world;

/* raw::MaterialTy, lib.rs:644-648 / shader.wgsl:126-127; 3 is the extension */
enum { MRT_LAMBERTIAN = 1, MRT_METAL = 2, MRT_DIELECTRIC = 3 };

/* ---- api::Sphere / api::DynMaterial, lib.rs:611-639, flattened to one POD ----
 * albedo is used by Lambertian and Metal; param = fuzz (Metal) or index of refraction
 * (Dielectric). 36 bytes. */
typedef struct {
    float   center[3];
    float   radius;
    int32_t material_ty;
    float   albedo[3];
    float   param;
} mrt_sphere;

/* ---- camera (extension; mode 0 is the reference's fixed pinhole, shader.wgsl:360-381) */
typedef struct {
    int32_t mode;                  /* 0 = reference pinhole, 1 = look-at thin lens */
    float   lookfrom[3], lookat[3], vup[3];
    float   vfov_deg, defocus_angle_deg, focus_dist;
} mrt_camera;
typedef struct {                   /* what the kernel consumes; derived on the host in double */
    int32_t mode, defocus;
    float   origin[3], su[3], sv[3], fw[3], ru[3], rv[3];
} mrt_camera_raw;

typedef struct {
    uint64_t samples;              /* camera rays started */
    uint64_t world_hit_calls;      /* = bounces; sphere tests = world_hit_calls * spheres.length */
    uint64_t rng_draws;            /* xoshiro128+ outputs consumed */
    uint64_t lane_slots;           /* 64 x trips of each wave's bounce loop: world_hit_calls / lane_slots
                                      = SIMD lane utilisation of the kernel (diagnostic, not in the oracle) */
    uint64_t member_tests;         /* per-sphere discriminants evaluated (members of candidate clusters + directly tested spheres) */
    uint64_t sweep_records;        /* top-level bound records the sweep tests per world_hit (not accumulated): executed
                                      bound tests of the sweep = world_hit_calls * sweep_records */
} mrt_counters;

typedef struct mrt_ctx mrt_ctx;

/* ------------------------------------------------------------------ lifecycle */

/* App::new + State::new (lib.rs:77,217-234): allocates the seed texture (Subject::new,
 * lib.rs:389-415; seeded deterministically from `seed` instead of entropy), the two
 * ping-pong framebuffers (DoubleFramebuffers::new, lib.rs:514-538) and the Locals.
 * `device` is the HIP device ordinal.  Fails with MRT_ERR_NO_DEVICE if there is none. */
int mrt_create(const mrt_args* args, uint64_t seed, int device, mrt_ctx** out);
/* Drop of State's wgpu handles */
void mrt_destroy(mrt_ctx* ctx);

/* Deadline, in seconds, of every wait for the GPU inside this library (the back-pressure of mrt_redraw, mrt_sync, read-backs,
 * re-allocations, mrt_destroy): a wait that lasts longer fails with MRT_ERR_STALLED and names itself ("back-pressure of slot 3,
 * frame 17, 8 frames in flight") instead of hanging -- the reference's frame loop would block in wgpu's present forever
 * (lib.rs:270).  Default 120 s (or the environment's MRT_WAIT_TIMEOUT_S at mrt_create); 0 = no deadline.  Must exceed the
 * longest single frame the caller renders. */
int mrt_set_wait_timeout(mrt_ctx* ctx, double seconds);

/* Tile sharding for multi-GPU (no reference counterpart): this ctx renders the 8-row
 * bands b with b % world == rank.  Must be called before the first redraw.  Seeds are
 * keyed by global pixel index, so any sharding yields the same image. */
int mrt_set_shard(mrt_ctx* ctx, uint32_t rank, uint32_t world);
/* hipStream_t to launch on (e.g. torch's current stream); NULL = the ctx's own stream. */
int mrt_set_stream(mrt_ctx* ctx, void* hip_stream);

/* ------------------------------------------------------------------ scene */

/* Object::new's upload (lib.rs:768-863): the raw::World index block and the three SoA
 * arrays, verbatim.  `world` points to world_bytes bytes: MRT_WORLD_BYTES_REFERENCE (64) = the
 * reference's raw::World exactly as lib.rs:676-684 lays it out (no dielectrics), or
 * sizeof(mrt_world) (80) = with the DielectricRange extension; nothing beyond world_bytes is
 * read.  vec4_data: n_vec4 x 4 floats, f32_data: n_f32 floats, i32_data: n_i32 ints.  All
 * base+length ranges are validated. */
#define MRT_WORLD_BYTES_REFERENCE 64
int mrt_set_world_raw(mrt_ctx* ctx, const void* world, size_t world_bytes,
                      const float* vec4_data, size_t n_vec4,
                      const float* f32_data, size_t n_f32,
                      const int32_t* i32_data, size_t n_i32);
/* api::World { spheres } (lib.rs:611-639) -> packs with mrt_pack_world, then uploads */
int mrt_set_world(mrt_ctx* ctx, const mrt_sphere* spheres, size_t n);
/* The AoS -> SoA packing of lib.rs:722-799 (host only, no GPU needed).  Capacities are
 * in elements (vec4: 4 floats each); returns MRT_ERR_TOO_SMALL if any is short.
 * Needs at most 2n vec4, 2n f32, 2n i32. */
int mrt_pack_world(const mrt_sphere* spheres, size_t n, mrt_world* world,
                   float* vec4_data, size_t cap_vec4, size_t* n_vec4,
                   float* f32_data, size_t cap_f32, size_t* n_f32,
                   int32_t* i32_data, size_t cap_i32, size_t* n_i32);

int mrt_set_camera(mrt_ctx* ctx, const mrt_camera* cam);
int mrt_camera_derive(const mrt_camera* cam, mrt_camera_raw* out);   /* host only */

/* Replace the seed texture (Rgba32Uint W x H of lib.rs:397-415): seeds = W*H*4 u32,
 * row 0 = bottom.  Optional: mrt_create already fills it from `seed`. */
int mrt_set_seeds(mrt_ctx* ctx, const uint32_t* seeds, size_t n_u32);
int mrt_read_seeds(mrt_ctx* ctx, uint32_t* out, size_t cap_u32);    /* this shard's rows, packed */

/* ------------------------------------------------------------------ frame loop */

/* State::redraw (lib.rs:241-307) minus the present pass: one raytrace pass of
 * samples_per_frame spp into framebuffers.target blended with .secondary, swap,
 * sample_count += 1, framebuffer_weight = min(max_w, n/(n+1)), new rng_shuffle, Locals
 * update.  Asynchronous on the ctx's stream, like a swap chain: consecutive frames overlap on the GPU (2 to 16 in flight,
 * see mrt_get_schedule), and the call returns at once unless that many frames
 * are already queued.
 * BACK-PRESSURE: in that case the call blocks the HOST (polling, with the deadline of mrt_set_wait_timeout) until the render
 * kernel of the oldest frame in flight has completed.  That kernel runs on a side stream of the library and waits only for
 * work queued by EARLIER calls (its slot's previous blend on the ctx's stream, the scene / seed uploads), never for anything
 * the caller submits later -- but a caller that gates the ctx's stream on an event it records only AFTER further mrt_redraw
 * calls can deadlock itself here, and the call must not be made while the ctx's stream is being captured into a graph.
 * The next call's blend is ordered behind this one's on the ctx's stream. */
int mrt_redraw(mrt_ctx* ctx);
/* The launch schedule (no reference counterpart: lib.rs:241-307 has one frame after the other).  A frame's render kernel runs
 * on 1 / div of the persistent waves the chip holds and max(2, div) x mult frames are in flight -- with mult 2, twice the
 * launches the chip holds: frames end out of order, and a queued launch takes every workgroup slot the moment it frees; the
 * library measures its way to a setting over the first frames of a workload.  mrt_get_schedule: out[0] = div, out[1] = mult, out[2] = 1
 * once the setting is final for the current workload (or pinned), out[3] = frames in flight, out[4] = the share the most recent
 * launch really got (1 / out[4]: never narrower than the frames the caller really keeps in flight), out[5] = frames that can run
 * side by side in this process (hardware queues: GPU_MAX_HW_QUEUES; 0 = not measured yet).
 * mrt_set_schedule_hint pins (div, mult) -- e.g. what an earlier run of the same workload settled at, or rank 0's setting on
 * every rank of a multi-GPU run -- so that no trial runs and two runs schedule alike; (0, 0) returns to the measured
 * setting.  div 1..8, mult 1..8, max(2, div) x mult <= 16; the frames in flight are held to out[5] where that has been measured
 * (a process with too few hardware queues).  Takes effect at the next redraw (a change waits for the frames
 * under way); the images are the same whatever the schedule. */
int mrt_get_schedule(mrt_ctx* ctx, uint32_t out[6]);
int mrt_set_schedule_hint(mrt_ctx* ctx, uint32_t div, uint32_t mult);
/* `frames` x mrt_redraw: the same images.  Frames are independent until their blend, so when the (shard of the) image has
 * fewer than about two pixels per GPU lane -- a pixel is one sequential chain of samples, lib.rs:299-306's remedy for that is
 * more frames -- up to 32 consecutive frames of the stream mode share one render launch (also when a frame is very short).
 * Memory: every frame of such a launch parks its colour sums in a layer of its own (16 B per pixel); a launch is held to 1 GiB
 * of them per frame in flight (two), e.g. 32 frames of 1920x1080 or 8 of 3840x2160; the buffers are kept until mrt_destroy. */
int mrt_render(mrt_ctx* ctx, uint32_t frames);
int mrt_sync(mrt_ctx* ctx);
/* Restart accumulation: zero framebuffers, frame counter 0, weight 0, shuffle [0;4]. */
int mrt_reset(mrt_ctx* ctx);

/* Current Locals (what the NEXT redraw will use) */
int mrt_get_locals(mrt_ctx* ctx, mrt_locals* out);
/* Override the next frame's rng_shuffle (the reference draws it from thread_rng, lib.rs:305) */
int mrt_set_rng_shuffle(mrt_ctx* ctx, const uint32_t shuffle[4]);
int mrt_set_samples_per_frame(mrt_ctx* ctx, uint32_t spp);
int mrt_set_rng_mode(mrt_ctx* ctx, uint32_t mode);          /* MRT_RNG_*; takes effect at the next redraw */
uint32_t mrt_frames_done(mrt_ctx* ctx);

/* host-only helpers exposing the schedule of lib.rs:300-305 */
float mrt_frame_weight(uint32_t frames_done, float max_framebuffer_weight);
void  mrt_frame_shuffle(uint64_t seed, uint32_t frame, uint32_t out[4]);
void  mrt_pixel_seed(uint64_t seed, uint64_t pixel_index, uint32_t out[4]);

/* ------------------------------------------------------------------ output */

/* Geometry of this shard's packed framebuffer: local_rows rows of `width` RGBA f32
 * texels; local row r is global row ((r/8)*world + rank)*8 + r%8 (rows >= height are
 * padding and hold zeros). */
int mrt_shard_info(mrt_ctx* ctx, uint32_t* rank, uint32_t* world, uint32_t* local_rows, uint32_t* width);
/* Device pointer of the most recently rendered framebuffer (local_rows*width*4 floats),
 * for zero-copy hand-off to RCCL / torch.  Valid until the next redraw/destroy. */
void* mrt_framebuffer_device_ptr(mrt_ctx* ctx);
/* Read-back (no reference counterpart; the reference only presents, lib.rs:270-297).
 * world == 1: the full image, height*width*4 floats, row 0 = bottom.
 * world  > 1: this shard's packed rows, local_rows*width*4 floats. */
int mrt_read_framebuffer(mrt_ctx* ctx, float* rgba_out, size_t cap_floats);
int mrt_read_counters(mrt_ctx* ctx, mrt_counters* out);   /* accumulated since create/reset */
/* mrt_counters::rng_draws is a per-lane counter in the kernel's rejection loop (a few VALU instructions per trip); the other
 * counters are wave totals kept on the scalar side.  0 launches the instantiation without it (rng_draws then stops
 * advancing); 1 (default) counts.  Takes effect at the next redraw; the images are the same. */
int mrt_set_draw_counting(mrt_ctx* ctx, int enabled);

/* ------------------------------------------------------------------ multi-GPU (no reference counterpart)
 *
 * The reference drives one adapter (lib.rs:329-335).  The caller that owns its frame loop (State::new /
 * State::redraw, lib.rs:217-234, :241-307) uses N GPUs by holding one mrt_ctx per GPU, ctxs[i] created on
 * device i with mrt_set_shard(ctxs[i], i, N), calling mrt_redraw on each and then ONE gather per frame. */

/* One process, N contexts.  Copies every shard's most recent framebuffer into its interleaved place in the
 * root's full-frame buffer: device-to-device (peer-to-peer over xGMI) copies of the shard's bands, issued on
 * that shard's own stream (so it follows its redraw), all links at once; the root's stream then waits for all
 * of them.  Asynchronous; read the result with mrt_read_gathered / mrt_gathered_device_ptr on ctxs[root].
 * Requires ctxs[i] to be shard i of n of the same width x height. */
int mrt_gather(mrt_ctx* const* ctxs, uint32_t n, uint32_t root);
/* One process per GPU (shard rank = RCCL rank): grouped ncclSend / ncclRecv of the packed bands on
 * `nccl_comm` (an ncclComm_t the caller created with ITS librccl; this library resolves RCCL's entry points at
 * run time and does not link it), one message per peer straight to the root, then the un-permute on the
 * root, all on the ctx's stream.  Collective: every rank calls it once per frame. */
int mrt_gather_rccl(mrt_ctx* ctx, void* nccl_comm, uint32_t root);
/* Full frame on the root after a gather: height*width*4 floats, row 0 = bottom (device pointer valid until the
 * next gather on this ctx; order further work after the ctx's stream: the next gather's copies wait for
 * everything queued on the root's stream before it, so an asynchronous reader queued there is never overtaken). */
void* mrt_gathered_device_ptr(mrt_ctx* root_ctx);
int mrt_read_gathered(mrt_ctx* root_ctx, float* rgba_out, size_t cap_floats);   /* synchronises */
/* host-only index math of the interleave: local row r of shard (rank, world) is this global row; rows per
 * shard; and the un-permute of a rank-major [world][local_rows][width][4] array into [height][width][4] */
uint32_t mrt_shard_global_row(uint32_t local_row, uint32_t rank, uint32_t world);
uint32_t mrt_shard_local_rows(uint32_t height, uint32_t world);
int mrt_unshard_rows(const float* gathered, uint32_t world, uint32_t width, uint32_t height, float* out);
/* Elapsed GPU time (ms) of the most recent redraw's render kernel, from HIP events on
 * the launch stream.  Synchronises on the stop event. */
int mrt_last_kernel_ms(mrt_ctx* ctx, float* ms);
/* The same for the n <= min(cap, 64) most recent redraws, oldest first. */
int mrt_kernel_ms_history(mrt_ctx* ctx, float* ms, size_t cap, size_t* n_out);

const char* mrt_last_error(mrt_ctx* ctx);       /* never NULL; ctx may be NULL */
/* Identifies the binary: the first 16 hex digits of the sha256 over the library's sources (the Makefile computes it at
 * build time; bench.py prints it next to the hash of the sources on disk and refuses a headline from a stale build). */
const char* mrt_build_id(void);
const char* mrt_status_string(int status);
int mrt_abi_version(void);

/* ------------------------------------------------------------------ scenes (host only) */

/* The shipped 4-sphere scene, lib.rs:687-720.  Returns the sphere count (4) or a
 * negative mrt_status; writes min(cap, n) spheres. */
int mrt_scene_default(mrt_sphere* out, size_t cap);
/* RTIOW cover scene (extension): ground + 22x22 grid + 3 big spheres.
 * dielectric == 0 emits glass slots as Metal(0.9,0.9,0.9; fuzz 0) (config C2),
 * dielectric != 0 as Dielectric(1.5) (C3/C4).  cam_out (optional) gets the matching
 * camera (defocus only when dielectric != 0). */
int mrt_scene_cover(uint64_t scene_seed, int dielectric, mrt_sphere* out, size_t cap, mrt_camera* cam_out);
/* 10k-sphere stress scene (C5): ground + n_side x n_side jittered grid, 80/15/5 % L/M/D */
int mrt_scene_stress(uint64_t scene_seed, uint32_t n_side, mrt_sphere* out, size_t cap, mrt_camera* cam_out);

/* Scenes as data: a line-oriented text file holding what api::World (lib.rs:611-639,
 * hard-coded in the reference, lib.rs:687-720) and the camera extension hold:
 *     # comment
 *     camera pinhole
 *     camera lookat <from xyz> <at xyz> <up xyz> <vfov deg> <defocus angle deg> <focus dist>
 *     sphere <centre xyz> <radius> lambertian <albedo rgb>
 *     sphere <centre xyz> <radius> metal <albedo rgb> <fuzz>
 *     sphere <centre xyz> <radius> dielectric <ior>
 *     sphere <centre xyz> <radius> material <ty> <albedo rgb> <param>     (any other MaterialTy: absorbs)
 * Sphere order = the reference's sphere index order (it decides ties, shader.wgsl:291-296).  Numbers are
 * written with 9 significant digits, so save -> load reproduces every f32 bit.
 * mrt_scene_load returns the number of spheres in the file (writes min(cap, n)) or a negative mrt_status
 * (-MRT_ERR_IO, -MRT_ERR_BAD_SCENE; mrt_last_error(NULL) names the line); *has_camera = 1 if the file has a
 * camera line (cam_out filled), else 0 (cam_out = the reference's pinhole). */
int mrt_scene_save(const char* path, const mrt_sphere* spheres, size_t n, const mrt_camera* cam);
int mrt_scene_load(const char* path, mrt_sphere* out, size_t cap, mrt_camera* cam_out, int* has_camera);

/* ------------------------------------------------------------------ image output (host only) */

/* rgba: height*width*4 floats, row 0 = bottom (as read back).  PFM keeps linear floats
 * (bottom-up is PFM's native order).  PPM is what the reference's present pass puts on screen: the
 * linear value (sample_framebuffer.wgsl:38-41) stored to the sRGB surface (lib.rs:349-351, :1133), i.e.
 * clamp to [0,1], the sRGB OETF (12.92 c below 0.0031308, else 1.055 c^(1/2.4) - 0.055), round to 8 bits,
 * rows flipped to top-down (sample_framebuffer.wgsl:24). */
int mrt_write_pfm(const char* path, const float* rgba, uint32_t width, uint32_t height);
int mrt_write_ppm(const char* path, const float* rgba, uint32_t width, uint32_t height);
int mrt_write_png(const char* path, const float* rgba, uint32_t width, uint32_t height);   /* 8-bit RGB, same encoding as the PPM, + sRGB chunk */
uint8_t mrt_srgb8(float linear);      /* the per-channel conversion mrt_write_ppm / mrt_write_png apply */

#ifdef __cplusplus
}
#endif
#endif /* MYRAYTRACER_AMD_H */
