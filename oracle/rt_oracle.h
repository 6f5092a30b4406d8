/*
 * rt_oracle.h -- CPU ORACLE for the per-pixel render loop of zetanumbers/myraytracer.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (myraytracer_amd/) never
 * includes, links or calls anything in this directory.
 *
 * It restates, in plain C, the WGSL fragment shader raytracer/src/shader.wgsl and the
 * host-side data contract of raytracer/src/lib.rs (citations on every function in
 * rt_oracle.c).  The reference cannot be built or run here (no Rust toolchain, and it
 * is a wgpu fragment shader without read-back or fixed seed), and it ships no tests or
 * golden vectors, so:
 *
 *   PARITY STATUS
 *   - pinned:   Xoshiro128+ integer stream (published vectors), u32->f32 conversion,
 *               scene packing indices of the shipped 4-sphere scene, analytic
 *               sphere_hit / color_sky cases (tests/test_oracle_kat.py).
 *   - UNPINNED: everything floating point.  WGSL leaves the precision/fusion of dot,
 *               normalize, reflect, mix, sqrt and '/' to the backend; this oracle fixes
 *               one legal reading (the "MRT-F32" rules below) that the HIP kernels
 *               reproduce bit for bit.
 *   - extensions beyond the reference (Dielectric, look-at/defocus camera, seeded
 *               RNG expansion) have no reference counterpart at all.
 *
 * MRT-F32 arithmetic rules (all IEEE-754 binary32, round-to-nearest-even, no FTZ):
 *   dot3(a,b)      = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   normalize(v)   = v / sqrt(dot3(v,v))          (three correctly rounded divisions)
 *   reflect(d,n)   = d - (2*dot3(n,d))*n          (mul and sub rounded separately)
 *   mix(a,b,t)     = a*(1-t) + b*t                (no fma)
 *   f32(u32)       = RNE conversion, then * 2^-32 (exact)
 *   sphere test    : oc=o-c; b=dot3(oc,dir); cc=fma(oc.z,oc.z,fma(oc.y,oc.y,fma(oc.x,oc.x,-(r*r))));
 *                    d=fma(b,b,-(a*cc))           (a = dot3(dir,dir))
 *   everything else is evaluated exactly as written in rt_oracle.c, one rounding per
 *   operator; the file must be compiled with -ffp-contract=off.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* raw::World of lib.rs:641-685 (64 B) + DielectricRange extension (16 B) = 80 B */
typedef struct {
    int32_t center_base_idx, radius_base_idx, material_ty_base_idx, material_idx_base_idx;
    int32_t length, _pad[3];
} orc_sphere_range;
typedef struct { int32_t albedo_base_idx, length, _pad[2]; } orc_lambertian_range;
typedef struct { int32_t albedo_base_idx, fuzz_base_idx, length, _pad; } orc_metal_range;
typedef struct { int32_t ior_base_idx, length, _pad[2]; } orc_dielectric_range;
typedef struct {
    orc_sphere_range spheres;
    orc_lambertian_range lambertians;
    orc_metal_range metals;
    orc_dielectric_range dielectrics;
} orc_world;

/* Locals uniform of lib.rs:368-377 / shader.wgsl:8-17 (48 B) */
typedef struct {
    uint32_t shape[2];
    uint32_t samples_per_frame;
    uint32_t ray_depth;
    uint32_t rng_shuffle[4];
    float framebuffer_weight;
    uint32_t rng_mode;          /* first padding word of the reference's Locals: 0 = its per-pixel stream */
    uint32_t _pad[2];
} orc_locals;
enum { ORC_RNG_PIXEL_STREAM = 0, ORC_RNG_COUNTER = 1 };
#define ORC_COUNTER_BLOCK 64u   /* counter mode: samples are summed in blocks of 64, the blocks' sums in index order */

/* user-level camera (extension; mode 0 = the fixed pinhole of shader.wgsl:360-381) */
typedef struct {
    int32_t mode;            /* 0 pinhole (reference), 1 look-at thin lens */
    float lookfrom[3], lookat[3], vup[3];
    float vfov_deg, defocus_angle_deg, focus_dist;
} orc_camera;

/* derived camera actually consumed per sample (all f32) */
typedef struct {
    int32_t mode; int32_t defocus;   /* defocus != 0 -> sample the lens disk */
    float origin[3], su[3], sv[3], fw[3], ru[3], rv[3];
} orc_camera_raw;

typedef struct {
    uint64_t samples, world_hit_calls, sphere_tests, rng_draws;
    uint64_t scatter_lambertian, scatter_metal, scatter_dielectric;
    uint64_t paths_missed, paths_absorbed, paths_exhausted;
} orc_counters;

/* material types, lib.rs:644-648 / shader.wgsl:126-127 (+3 extension) */
enum { ORC_LAMBERTIAN = 1, ORC_METAL = 2, ORC_DIELECTRIC = 3 };

/* ---- RNG (shader.wgsl:36-94) ---- */
uint32_t orc_xoshiro128plus_next(uint32_t s[4]);
float    orc_u32_to_f32(uint32_t u);
uint64_t orc_splitmix64_at(uint64_t seed, uint64_t k);
void     orc_pixel_seed(uint64_t seed, uint64_t pixel_index, uint32_t out[4]);
void     orc_fill_seeds(uint64_t seed, uint32_t w, uint32_t h, uint32_t* seeds /* w*h*4 */);
void     orc_frame_shuffle(uint64_t seed, uint32_t frame, uint32_t out[4]);
float    orc_frame_weight(uint32_t frames_done, float max_w);
void     orc_sample_state(const uint32_t base[4], uint32_t sample, uint32_t out[4]);

/* ---- unit-testable pieces ---- */
/* returns 1 on hit; out = {at[3], t, normal[3], front_face, ty, idx} */
typedef struct { float at[3]; float t; float normal[3]; int32_t front_face, ty, idx; } orc_hit;
int  orc_sphere_hit(const orc_world* w, const float* vec4, const float* f32, const int32_t* i32,
                    int32_t idx, const float orig[3], const float dir[3], float t_min, float t_sup,
                    orc_hit* out);
int  orc_world_hit(const orc_world* w, const float* vec4, const float* f32, const int32_t* i32,
                   const float orig[3], const float dir[3], float t_min, float t_sup,
                   orc_hit* out, int32_t* hit_sphere);
void orc_color_sky(float y, float out[3]);
/* world_hit over [0.001, 1e4) for n rays (6 floats each: origin, direction): hit_sphere[r] = winner or -1, hit_t[r] = its t
 * (1e4 on a miss); disc_ge0 (optional, n x spheres.length bytes): bit 0 = sphere_hit's discriminant is not < 0, bit 1 = that AND
 * the sphere is not entirely behind the ray's origin (b >= +0 and c >= +0 of :277-281, where both roots are <= 0) */
void orc_world_hit_batch(const orc_world* w, const float* vec4, const float* f32, const int32_t* i32,
                         const float* rays, int64_t n, int32_t* hit_sphere, float* hit_t, uint8_t* disc_ge0, int nthreads);
void orc_camera_derive(const orc_camera* cam, orc_camera_raw* out);

/* AoS -> SoA packing of lib.rs:722-799.  spheres: n x {cx,cy,cz,r, ty, p0,p1,p2,p3}
 * (p = albedo rgb in p0..p2; p3 = fuzz for metal / ior for dielectric).
 * Output arrays must hold: vec4 (n + n)*4 floats worst case, f32 2n floats, i32 2n ints.
 * Returns lengths through n_vec4/n_f32/n_i32. */
typedef struct { float center[3]; float radius; int32_t ty; float p[4]; } orc_sphere_aos;
void orc_pack_world(const orc_sphere_aos* spheres, int32_t n, orc_world* world,
                    float* vec4, int32_t* n_vec4, float* f32, int32_t* n_f32,
                    int32_t* i32, int32_t* n_i32);

/* ---- the render pass (shader.wgsl:371-386 over every pixel of rows [y0,y1)) ----
 * seeds: W*H x [u32;4] (Rgba32Uint texture of lib.rs:397-415), row 0 = bottom.
 * prev / out: W*H x rgba f32, row 0 = bottom (sample_framebuffer.wgsl:24 flips on present).
 * counters may be NULL.  nthreads <= 0 -> all cores. */
void orc_render_rows(const orc_locals* locals, const orc_world* world,
                     const float* vec4, const float* f32, const int32_t* i32,
                     const orc_camera_raw* cam, const uint32_t* seeds,
                     const float* prev, float* out,
                     uint32_t y0, uint32_t y1, int nthreads, orc_counters* counters);

/* the same over the pixel rectangle [x0,x1) x [y0,y1) only (other texels of `out` are left untouched);
 * threads share the rectangle's pixels, so a single row also uses every core */
void orc_render_rect(const orc_locals* locals, const orc_world* world,
                     const float* vec4, const float* f32, const int32_t* i32,
                     const orc_camera_raw* cam, const uint32_t* seeds,
                     const float* prev, float* out,
                     uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, int nthreads, orc_counters* counters);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
