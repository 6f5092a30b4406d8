/*
 * myraytracer_amd_debug.h -- diagnostic and tuning entry points of libmyraytracer_amd.so.
 *
 * NOT part of the drop-in boundary: nothing here maps to a seam of the reference (raytracer/src/lib.rs has no
 * counterpart for any of it).  The product header is myraytracer_amd.h; this one exists for the parity tests
 * (candidate sets of the sweep + walk, the hand-rolled division / square root against hipcc's, hierarchy
 * inspection) and for the A/B switches the measurements in DESIGN_HISTORY.md were taken with.  The symbols are
 * exported by the same library; tests/test_abi.py checks both headers against it.
 */
#ifndef MYRAYTRACER_AMD_DEBUG_H
#define MYRAYTRACER_AMD_DEBUG_H

#include "myraytracer_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic: make mrt_gather on this root use the cross-device form of the copy (one hipMemcpyPeerAsync per
 * band) even when a shard shares the root's device, so that its indexing runs on a one-GPU box. */
int mrt_debug_set_gather_per_band(mrt_ctx* root_ctx, int enabled);
/* Diagnostic: the 16 raw u64 counter slots  (0..4 = mrt_counters; 6.. are phase cycle sums
 * written only by the -DMRT_STAMPS profiling build). */
int mrt_debug_read_counters(mrt_ctx* ctx, uint64_t out[16]);
/* Diagnostic: per-pixel cost (bounce-loop trips) of the last frame, local_rows*width u32. */
int mrt_debug_read_pixel_costs(mrt_ctx* ctx, uint32_t* out, size_t cap);
/* Diagnostic / tuning: 0 = one sphere per cluster record, > 0 = clusters of up to 4 spheres (the value itself
 * is no longer used); takes effect at the next mrt_set_world* call. */
int mrt_debug_set_cluster_factor(mrt_ctx* ctx, float factor);
/* Diagnostic / tuning: depth of the bounding-sphere hierarchy built by the next mrt_set_world* call:
 * levels are added (up to max_levels, 1..4) while the top level has more than top_target records (0 = automatic: 256, or
 * 128 for scenes beyond 4,096 member slots, whose walk tests boxes below the top). */
int mrt_debug_set_hierarchy(mrt_ctx* ctx, uint32_t max_levels, uint32_t top_target);
/* Diagnostic / tuning: which variant of the conservative sweep runs: 0 = automatic (matrix cores where the
 * expanded test's slack is negligible for the scene and camera), 1 = SGPR-fed VALU sweep, 2 = matrix cores.
 * Either way the image is the same; takes effect at the next redraw. */
int mrt_debug_set_sweep(mrt_ctx* ctx, int mode);
/* Diagnostic, host only (no context, no device): builds the bounding-sphere hierarchy that mrt_set_world would
 * upload for these spheres and returns it for inspection (tests/test_host_logic.py checks its invariants):
 *   top_out    n_top x (cx, cy, cz, -R^2)          the swept level, padded to a multiple of 32 with never-hit records
 *   nodes_out  n_nodes x (cx, cy, cz, -R^2 | -r^2) levels 0 .. levels-1, level k at info[6 + k]
 *   member_index_out  n_members sphere indices (level 0; padding slots hold 0 and a never-hit record)
 *   mfma_out   n_top / 32 tiles x 512 bf16         the A operand of the matrix-core sweep, origin in mfma_origin_out
 *   info[10] = {levels, n_top, n_nodes, n_members, n_direct, direct_first, level_base[0..3]}
 * Any output pointer may be NULL (sizes are still returned in info); returns MRT_ERR_TOO_SMALL if a capacity
 * (in records / indices / bf16 values) is insufficient. */
int mrt_debug_build_hierarchy(const mrt_sphere* spheres, size_t n, uint32_t max_levels, uint32_t top_target,
                              float* top_out, size_t top_cap, float* nodes_out, size_t nodes_cap,
                              uint32_t* member_index_out, size_t member_cap, uint16_t* mfma_out, size_t mfma_cap,
                              float mfma_origin_out[3], uint32_t info[10]);
/* The same build, returning the axis-aligned boxes of the hierarchy's nodes (the walk's second bound for scenes beyond 1,024
 * member slots): boxes_out = n_boxes x (cx, cy, cz, ex, ey, ez, kc, kpad) -- centre, half extents, and the coefficients of
 * the test's slack K = kc X + kpad, X = |p|^2 (info[2] = 1) or |p|_1 (info[2] = 0) of the ray origin relative to the
 * centre; a never-hit box has extents -3e38.  info[8] = {levels, n_boxes, quadratic?, box_base[0..4]}: the boxes of level
 * k (1 <= k <= levels, the swept top last) start at box_base[k], parallel to that level's records. */
int mrt_debug_build_boxes(const mrt_sphere* spheres, size_t n, uint32_t max_levels, uint32_t top_target, float* boxes_out,
                          size_t boxes_cap, uint32_t info[8]);
/* The same boxes in the order the kernel walks them (what mrt_set_world uploads for a large scene): numbered top-down over the
 * complete 4-ary tree below the n_top swept records -- depth t at o_t = n_top (4^t - 1) / 3, so that the children of node g,
 * whatever its depth, are 4 g + n_top .. + 3; slots without a node hold never-hit boxes.  open != 0: every real box opened
 * wide (the form mrt_debug_set_boxes(0) selects).  info[5] = {levels, n_boxes, n_top, first cluster-level node, first node
 * whose children are clusters}. */
int mrt_debug_build_boxes_top_down(const mrt_sphere* spheres, size_t n, uint32_t max_levels, uint32_t top_target, int open,
                                   float* boxes_out, size_t boxes_cap, uint32_t info[5]);
/* Host-side diagnostic, no GPU needed: the ray-side factors of the matrix-core sweep for a scene and camera that keep every
 * ray origin and every bound within `reach` of the sweep's origin: scale_out[4] = {stretch K, 2 K^2,
 * -(1 - 2^-13) K^2, (4 reach)^2} with K the power of two for which |K oc.ds| <= 1/2 for every admitted ray, and
 * *neg_k2_bf16_pair_out = -K^2 as two bf16.  What mrt_redraw passes to the kernel (tests/test_host_logic.py). */
int mrt_debug_mfma_scale(double reach, float scale_out[4], uint32_t* neg_k2_bf16_pair_out);
/* Diagnostic: ONE world_hit (shader.wgsl:314-329, range [0.001, 1e4)) for each of n caller-supplied rays -- rays[6 i ..] =
 * origin xyz, direction xyz; directions of unit length to 1e-5, as every ray of the render loop is -- through the very sweep +
 * walk the render kernel runs (the same kernel, instantiated to take its rays from this array), with the current scene,
 * hierarchy and sweep variant.  hit_out[2 i] = index of the closest sphere or -1, hit_out[2 i + 1] = the bits of its t;
 * candidates_out (optional): cand_words_per_ray >= ceil(spheres/32) words per ray, bit s set iff sphere s reached the root
 * tests, i.e. passed the conservative sweep, every level of the walk AND the exact discriminant test.  The conservativeness
 * claim of DESIGN.md 4 is that this set equals {s : discriminant_s >= 0} for every ray (tests/test_gpu_superset.py). */
int mrt_debug_world_hit(mrt_ctx* ctx, const float* rays, size_t n, int32_t* hit_out, uint32_t* candidates_out,
                        size_t cand_words_per_ray);
/* Diagnostic: the render kernel's own forms of division and square root -- the correctly rounded expansions of `/` and
 * sqrtf() WITHOUT their operand-scaling steps, used at every root, hit normal and normalize (shader.wgsl:286-299, :354,
 * :381) where the operands cannot need those steps -- against hipcc's `/` and sqrtf() on the device, bit for bit (two NaNs
 * count as equal):
 *   mode 0: the square root of EVERY f32 with bit pattern in [bits_range[0], bits_range[1]];
 *   mode 1: `count` quotients n / d, |n| a bit pattern drawn uniformly from [bits_range[0], bits_range[1]], |d| from
 *           [bits_range[2], bits_range[3]] (SplitMix64 of seed and index), n of either sign; mode 2: d of either sign too.
 * out[0] = operands tested, out[1] = operands whose results differ, out[2] = the smallest differing operand (mode 0: the
 * bits of x; else bits(n) | bits(d) << 32; ~0 if none). */
int mrt_debug_arith(mrt_ctx* ctx, int mode, const uint32_t bits_range[4], uint64_t count, uint64_t seed, uint64_t out[3]);
/* The same for n caller-supplied operand pairs: out[6 i ..] = bits(x / y), bits(unscaled quotient), bits(sqrtf(x)),
 * bits(unscaled root), and the render kernel's two per-wave operand tests evaluated on the pair -- hit normal: x a component
 * of (at - centre), y the radius; normalize: x the squared length, y a component -- 1 = unscaled forms, 0 = the wave takes
 * the literal `/` and sqrtf(). */
int mrt_debug_arith_pairs(mrt_ctx* ctx, const float* x, const float* y, size_t n, uint32_t* out);
/* Diagnostic: which instantiation of the render kernel the most recent redraw launched (out[0]) and, if it was preceded by
 * a cost-estimating pilot launch not yet reported, which one that was (out[1], else 0xFFFFFFFF): bit 0 = with the RNG draw
 * counter, 1 = pilot, 2 = counter-RNG mode, 3 = small-scene layout, 4 = matrix-core sweep, 5 = large-scene layout with the
 * quadratic form of the box test's slack (else the linear form)
 * (tests/test_gpu_parity.py::test_every_render_kernel_instantiation_against_the_oracle). */
int mrt_debug_last_launch(mrt_ctx* ctx, uint32_t out[2]);
/* Diagnostic A/B switch (large scenes, > 1,024 member slots, whose walk tests the axis-aligned box of every node): 0 opens
 * every box wide, so that the box tests never reject and the walk reaches every member below the swept candidates; 1 (default)
 * and 2: the real boxes.  Either way the image is the same.  Takes effect at the next redraw. */
int mrt_debug_set_boxes(mrt_ctx* ctx, int mode);
/* Which variant the next redraw will run with the current scene, camera and mode: 1 or 2 (0 before a scene is set). */
int mrt_debug_sweep_variant(mrt_ctx* ctx);
/* Diagnostic A/B switch: 0 makes mrt_render launch every frame on its own; 1 = automatic (default); 2 / 3 force the form a
 * batch takes -- 2: the lane that takes a pixel renders it for every frame of the batch (what short frames of a large image
 * get), 3: the frames are layers of the tile queue (what a pixel-starved shard gets).  The images are the same. */
int mrt_debug_set_frame_batching(mrt_ctx* ctx, int enabled);
/* Diagnostic A/B switch: 0 queues tiles in index order instead of heaviest-first. */
int mrt_debug_set_tile_sort(mrt_ctx* ctx, int enabled);
/* Diagnostic / tuning: pilot samples per pixel, waves per CU (0 = automatic).  Before the first
 * redraw only. */
int mrt_debug_set_schedule(mrt_ctx* ctx, uint32_t pilot_spp, int waves_per_cu);
/* Host-side diagnostic, no GPU needed: the LDS footprint of the render kernel for a scene with this hierarchy (the sizes
 * mrt_debug_build_hierarchy reports): out[3] = {bytes per workgroup of 4 waves, workgroups resident per CU (160 KB of LDS;
 * large scenes: at most 4, the occupancy their kernels are built for), entries of a wave's top queue (small scenes) / work
 * stack (large scenes)}.  tests/test_hierarchy_host.py pins the residency of C3's and C5's layouts with it. */
int mrt_debug_lds_layout(uint32_t n_members, uint32_t n_nodes, uint32_t levels, uint32_t n_top_padded, uint32_t out[3]);
/* Diagnostic / tuning: how many frames may be in flight (each on a side stream of its own; 1..8), 0 = automatic: 2, more for
 * pixel-starved shards.  A change waits for the frames under way.  The images are the same. */
int mrt_debug_set_frames_in_flight(mrt_ctx* ctx, int slots);
/* Host only, no GPU needed: the launch-width controller's policy (csrc/width_policy.h) on synthetic input, for its CPU unit
 * tests.  workload[6] = {n_tiles, n_waves, max_slots, spp, n_members, counter-RNG?}; state[7] in / out = {div, mult, prev_div,
 * prev_mult, low_windows, settled, prev_rate as the bits of an f32}.  op 0: what is known up front (state out only);
 * op 1: one measurement window has closed with lane utilisation `util` and `rate` frames / s; op 2: state[0] out = the share a
 * launch gets under setting state[0] when `util` (as an integer) earlier frames are still queued or running. */
int mrt_debug_width_policy(int op, const uint32_t workload[6], uint32_t state[7], double util, double rate);
/* Diagnostic: how many of `streams` (2..8) side streams of this context really run side by side in this process -- identical
 * short clock-bounded kernels, one per stream, each stamping its start and end on the device's clock: *out = the most of them
 * resident at one instant (hardware queues are shared round-robin: HIP's default of 4 per process gives 4).  What caps the
 * frames in flight (mrt_get_schedule). */
int mrt_debug_stream_concurrency(mrt_ctx* ctx, uint32_t streams, float* out);
/* Diagnostic: per-wave log {t_start, t_end (100 MHz ticks), loop trips, bounces}, 4 u64 per 8x8
 * persistent wave, written only by the -DMRT_STAMPS build.  out == NULL allocates the log. */
int mrt_debug_wave_log(mrt_ctx* ctx, uint64_t* out, size_t cap_waves, size_t* n_waves);
/* The same for the frame `back` (0..31) redraws before the most recent one: the log is a ring over the last 32 frames, so that
 * the frames in flight of a narrow schedule can be laid side by side (scripts/shard_occupancy.py).  Entries of waves a launch
 * did not have are zero. */
int mrt_debug_wave_log_frame(mrt_ctx* ctx, uint32_t back, uint64_t* out, size_t cap_waves, size_t* n_waves);
/* Host wall time (ms) the most recent mrt_set_world* call spent building and uploading the bounding-sphere
 * hierarchy (a one-off per scene, outside the per-frame metric). */
int mrt_debug_last_set_world_ms(mrt_ctx* ctx, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* MYRAYTRACER_AMD_DEBUG_H */
