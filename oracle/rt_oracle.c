/*
 * rt_oracle.c -- CPU ORACLE (test infrastructure only; see rt_oracle.h header comment).
 *
 * A function-by-function restatement of /root/reference/raytracer/src/shader.wgsl and
 * of the host data contract in /root/reference/raytracer/src/lib.rs, under the MRT-F32
 * arithmetic rules of rt_oracle.h.  Compile with: gcc -O2 -ffp-contract=off -mfma -fopenmp
 */
#include "rt_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(float s, v3 a) { return v3_make(s * a.x, s * a.y, s * a.z); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline v3 v3_divs(v3 a, float s) { return v3_make(a.x / s, a.y / s, a.z / s); }

/* WGSL dot(); MRT-F32: fma chain x, then y, then z */
/* Which legal reading of WGSL's arithmetic the oracle takes.  Default = "MRT-F32" (DESIGN.md 3): dot() and the discriminant
 * as fma chains.  -DORC_READING_NOFMA builds the OTHER common lowering -- dot(a, b) = a.x*b.x + a.y*b.y + a.z*b.z with every
 * product and sum rounded separately, no fused operation anywhere (shader.wgsl:277-280 leave this to naga's backend) -- as
 * librt_oracle_nofma.so: never a parity target, only the yardstick of scripts/reading_spread.py, which measures how far two
 * legal readings of the reference sit from each other. */
#ifdef ORC_READING_NOFMA
#define ORC_FMA(a, b, c) ((a) * (b) + (c))
#else
#define ORC_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#endif
static inline float dot3(v3 a, v3 b) {
    return ORC_FMA(a.z, b.z, ORC_FMA(a.y, b.y, a.x * b.x));
}
/* WGSL normalize(e) = e / length(e) */
static inline v3 normalize3(v3 v) { return v3_divs(v, sqrtf(dot3(v, v))); }
/* WGSL reflect(e1,e2) = e1 - 2*dot(e2,e1)*e2   (shader.wgsl:230) */
static inline v3 reflect3(v3 d, v3 n) {
    float k = 2.0f * dot3(n, d);
    return v3_make(d.x - k * n.x, d.y - k * n.y, d.z - k * n.z);
}
/* WGSL mix(e1,e2,e3) = e1*(1-e3) + e2*e3 */
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

/* ------------------------------------------------------------------ RNG */

/* shader.wgsl:36-38 */
static inline uint32_t rotl_u32(uint32_t x, uint32_t k) { return (x << k) | (x >> (32u - k)); }

typedef struct { uint32_t s[4]; uint64_t draws; } rng_t;

/* shader.wgsl:49-64 xoshiro128plus_random_u32 */
uint32_t orc_xoshiro128plus_next(uint32_t s[4]) {
    uint32_t result = s[0] + s[3];
    uint32_t t = s[1] << 9;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl_u32(s[3], 11);
    return result;
}

/* shader.wgsl:66-69: f32(i) / 4294967296.0 ; division by 2^32 == exact scaling */
float orc_u32_to_f32(uint32_t u) { return (float)u * 0x1p-32f; }

static inline float rand_f32(rng_t* r) { r->draws++; return orc_u32_to_f32(orc_xoshiro128plus_next(r->s)); }

/* shader.wgsl:77-82 (x, then y, then z) */
static inline v3 rand_vec3(rng_t* r) {
    float x = rand_f32(r); float y = rand_f32(r); float z = rand_f32(r);
    return v3_make(x, y, z);
}

/* shader.wgsl:84-90: rejection, loop while dot(v,v) > 1.0 */
static inline v3 rand_unit_ball(rng_t* r) {
    v3 q = rand_vec3(r);
    v3 v = v3_make(2.0f * q.x - 1.0f, 2.0f * q.y - 1.0f, 2.0f * q.z - 1.0f);
    while (dot3(v, v) > 1.0f) {
        q = rand_vec3(r);
        v = v3_make(2.0f * q.x - 1.0f, 2.0f * q.y - 1.0f, 2.0f * q.z - 1.0f);
    }
    return v;
}
/* shader.wgsl:92-94 */
static inline v3 rand_unit_sphere(rng_t* r) { return normalize3(rand_unit_ball(r)); }

/* Seeding.  The reference draws W*H x [u32;4] from an entropy-seeded SplitMix64
 * (lib.rs:389-395) -- unreproducible by design.  Build-defined replacement: SplitMix64
 * used as a counter-based generator, 2 outputs per pixel, keyed by the global pixel
 * index (so any shard of the image derives identical seeds). */
uint64_t orc_splitmix64_at(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1u) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void orc_pixel_seed(uint64_t seed, uint64_t pixel_index, uint32_t out[4]) {
    uint64_t a = orc_splitmix64_at(seed, 2u * pixel_index);
    uint64_t b = orc_splitmix64_at(seed, 2u * pixel_index + 1u);
    out[0] = (uint32_t)a; out[1] = (uint32_t)(a >> 32);
    out[2] = (uint32_t)b; out[3] = (uint32_t)(b >> 32);
    /* lib.rs:393 filters the all-zero state (a fixed point of xoshiro) */
    if ((out[0] | out[1] | out[2] | out[3]) == 0u) {
        out[0] = 0x9E3779B9u; out[1] = 0x7F4A7C15u; out[2] = 0xBF58476Du; out[3] = 0x1CE4E5B9u;
    }
}
void orc_fill_seeds(uint64_t seed, uint32_t w, uint32_t h, uint32_t* seeds) {
    uint64_t n = (uint64_t)w * h;
    for (uint64_t p = 0; p < n; p++) orc_pixel_seed(seed, p, seeds + 4 * p);
}
/* lib.rs:305 draws rng_shuffle from thread_rng() after every frame; frame 0 uses
 * [0;4] (lib.rs:419-426).  Build-defined deterministic replacement. */
void orc_frame_shuffle(uint64_t seed, uint32_t frame, uint32_t out[4]) {
    if (frame == 0) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    uint64_t s2 = seed ^ 0xD1B54A32D192ED03ull;
    uint64_t a = orc_splitmix64_at(s2, 2ull * frame);
    uint64_t b = orc_splitmix64_at(s2, 2ull * frame + 1u);
    out[0] = (uint32_t)a; out[1] = (uint32_t)(a >> 32);
    out[2] = (uint32_t)b; out[3] = (uint32_t)(b >> 32);
}
/* EXTENSION (north_star's "counter-based RNG per lane"; no reference counterpart): in mode
 * ORC_RNG_COUNTER every sample starts from a Xoshiro128+ state that is a hash of the pixel's
 * frame state (seed texel ^ rng_shuffle) and the sample's index, so samples do not depend on how
 * many draws earlier samples consumed.  Within a sample the draw order is the reference's. */
static inline uint32_t fmix32(uint32_t z) {
    z ^= z >> 16; z *= 0x85EBCA6Bu; z ^= z >> 13; z *= 0xC2B2AE35u; z ^= z >> 16;
    return z;
}
void orc_sample_state(const uint32_t base[4], uint32_t sample, uint32_t out[4]) {
    for (uint32_t j = 0; j < 4; j++) out[j] = fmix32(base[j] + 0x9E3779B9u * (4u * sample + j + 1u));
    if ((out[0] | out[1] | out[2] | out[3]) == 0u) {
        out[0] = 0x9E3779B9u; out[1] = 0x7F4A7C15u; out[2] = 0xBF58476Du; out[3] = 0x1CE4E5B9u;
    }
}

/* lib.rs:300-304: weight used by the NEXT frame after `frames_done` frames */
float orc_frame_weight(uint32_t frames_done, float max_w) {
    if (frames_done == 0) return 0.0f;   /* lib.rs:424 initial value */
    float w = (float)frames_done / (float)(frames_done + 1u);
    return max_w < w ? max_w : w;        /* f32::min */
}

/* ------------------------------------------------------------------ scene access */

typedef struct {
    const orc_world* w; const float* vec4; const float* f32; const int32_t* i32;
} scene_t;

/* shader.wgsl:254-268 */
static inline v3 sphere_load_center(const scene_t* s, int32_t idx) {
    const float* p = s->vec4 + 4 * (size_t)(s->w->spheres.center_base_idx + idx);
    return v3_make(p[0], p[1], p[2]);
}
static inline float sphere_load_radius(const scene_t* s, int32_t idx) {
    return s->f32[s->w->spheres.radius_base_idx + idx];
}

typedef struct { v3 at; float t; v3 normal; int front_face; int32_t ty, idx; } hit_t;

/* shader.wgsl:280: c = dot(oc, oc) - radius * radius.  MRT-F32 folds -r^2 into the head of the fma chain; the
 * no-fma reading evaluates the text as written */
static inline float sphere_c(v3 oc, float radius) {
#ifdef ORC_READING_NOFMA
    return dot3(oc, oc) - radius * radius;
#else
    return __builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, __builtin_fmaf(oc.x, oc.x, -(radius * radius))));
#endif
}

/* shader.wgsl:274-282: a, b and the discriminant d of sphere_hit (what `if d < 0 { return false }` tests) */
static inline float sphere_discriminant(v3 center, float radius, v3 orig, v3 dir, float* a_out, float* b_out) {
    v3 oc = v3_sub(orig, center);
    float a = dot3(dir, dir);
    float b = dot3(oc, dir);
    /* c = dot(oc,oc) - radius*radius, MRT-F32 form (see header) */
    float c = sphere_c(oc, radius);
    *a_out = a; *b_out = b;
#ifdef ORC_READING_NOFMA
    return b * b - a * c;                               /* :282 as written */
#else
    return __builtin_fmaf(b, b, -(a * c));
#endif
}

/* shader.wgsl:270-312 sphere_hit */
static inline int sphere_hit(const scene_t* s, int32_t idx, v3 orig, v3 dir, float t_min, float t_sup, hit_t* out) {
    v3 center = sphere_load_center(s, idx);
    float radius = sphere_load_radius(s, idx);

    float a, b;
    float d = sphere_discriminant(center, radius, orig, dir, &a, &b);

    if (d < 0.0f) return 0;

    float d_sqrt = sqrtf(d);
    float t = (-b - d_sqrt) / a;
    if (t < t_min || t_sup <= t) t = (-b + d_sqrt) / a;
    if (t < t_min || t_sup <= t) return 0;

    v3 at = v3_add(orig, v3_scale(t, dir));            /* shader.wgsl:103-105 */
    v3 normal = v3_divs(v3_sub(at, center), radius);
    int front_face = dot3(normal, dir) <= 0.0f;
    if (!front_face) normal = v3_neg(normal);

    out->at = at; out->t = t; out->normal = normal; out->front_face = front_face;
    out->ty = s->i32[s->w->spheres.material_ty_base_idx + idx];
    out->idx = s->i32[s->w->spheres.material_idx_base_idx + idx];
    return 1;
}

/* shader.wgsl:314-329 world_hit: linear scan, t_sup shrinks, ties keep the lower index */
static inline int world_hit(const scene_t* s, v3 orig, v3 dir, float t_min, float t_sup, hit_t* out, int32_t* which) {
    hit_t tmp; int result = 0;
    memset(&tmp, 0, sizeof tmp);
    for (int32_t i = 0; i < s->w->spheres.length; i++) {
        if (sphere_hit(s, i, orig, dir, t_min, t_sup, &tmp)) {
            t_sup = tmp.t; *out = tmp; result = 1;
            if (which) *which = i;
        }
    }
    return result;
}

/* shader.wgsl:331-334 */
static inline v3 color_sky(float y_norm) {
    float t = 0.5f * y_norm + 0.5f;
    return v3_make(mixf(1.0f, 0.5f, t), mixf(1.0f, 0.7f, t), mixf(1.0f, 1.0f, t));
}

/* ------------------------------------------------------------------ materials */

typedef struct { v3 attenuation; v3 orig, dir; } scatter_t;

/* shader.wgsl:198-216 */
static inline int lambertian_scatter(const scene_t* s, int32_t idx, rng_t* rng, const hit_t* hit, scatter_t* out) {
    const float* al = s->vec4 + 4 * (size_t)(s->w->lambertians.albedo_base_idx + idx);
    v3 dir = v3_add(hit->normal, rand_unit_sphere(rng));
    if (dot3(dir, dir) == 0.0f) dir = hit->normal;
    out->attenuation = v3_make(al[0], al[1], al[2]);
    out->orig = hit->at; out->dir = dir;
    return 1;
}
/* shader.wgsl:218-242 */
static inline int metal_scatter(const scene_t* s, int32_t idx, rng_t* rng, v3 ray_dir, const hit_t* hit, scatter_t* out) {
    v3 normal = hit->normal;
    v3 refl = reflect3(ray_dir, normal);
    float fuzz = s->f32[s->w->metals.fuzz_base_idx + idx];
    v3 ball = rand_unit_ball(rng);
    v3 dir = v3_make(refl.x + fuzz * ball.x, refl.y + fuzz * ball.y, refl.z + fuzz * ball.z);
    if (dot3(dir, normal) <= 0.0f) return 0;
    const float* al = s->vec4 + 4 * (size_t)(s->w->metals.albedo_base_idx + idx);
    out->attenuation = v3_make(al[0], al[1], al[2]);
    out->orig = hit->at; out->dir = dir;
    return 1;
}
/* EXTENSION (no reference counterpart; SURVEY.md 8a "conventions to preserve"):
 * material type 3, RTIOW dielectric with Schlick reflectance, exactly one RNG draw. */
static inline int dielectric_scatter(const scene_t* s, int32_t idx, rng_t* rng, v3 d, const hit_t* hit, scatter_t* out) {
    float ior = s->f32[s->w->dielectrics.ior_base_idx + idx];
    v3 n = hit->normal;
    float ri = hit->front_face ? (1.0f / ior) : ior;
    float cos_t = dot3(v3_neg(d), n);
    cos_t = (cos_t < 1.0f) ? cos_t : 1.0f;
    float sin_t = sqrtf(1.0f - cos_t * cos_t);
    int cannot_refract = (ri * sin_t) > 1.0f;
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    float x = 1.0f - cos_t;
    float x2 = x * x; float x4 = x2 * x2; float x5 = x4 * x;
    float reflectance = r0 + (1.0f - r0) * x5;
    float u = rand_f32(rng);
    v3 dir;
    if (cannot_refract || reflectance > u) {
        dir = reflect3(d, n);
    } else {
        v3 perp = v3_make(ri * (d.x + cos_t * n.x), ri * (d.y + cos_t * n.y), ri * (d.z + cos_t * n.z));
        float k = -sqrtf(fabsf(1.0f - dot3(perp, perp)));
        dir = v3_make(perp.x + k * n.x, perp.y + k * n.y, perp.z + k * n.z);
    }
    out->attenuation = v3_make(1.0f, 1.0f, 1.0f);
    out->orig = hit->at; out->dir = dir;
    return 1;
}
/* shader.wgsl:244-252 */
static inline int dyn_material_scatter(const scene_t* s, rng_t* rng, v3 ray_dir, const hit_t* hit, scatter_t* out, orc_counters* c) {
    if (hit->ty == ORC_LAMBERTIAN) { c->scatter_lambertian++; return lambertian_scatter(s, hit->idx, rng, hit, out); }
    else if (hit->ty == ORC_METAL) { c->scatter_metal++; return metal_scatter(s, hit->idx, rng, ray_dir, hit, out); }
    else if (hit->ty == ORC_DIELECTRIC) { c->scatter_dielectric++; return dielectric_scatter(s, hit->idx, rng, ray_dir, hit, out); }
    return 0;
}

/* shader.wgsl:336-358 color_world */
static inline v3 color_world(const scene_t* s, v3 orig, v3 dir, uint32_t depth, rng_t* rng, orc_counters* c) {
    v3 attenuation = v3_make(1.0f, 1.0f, 1.0f);
    for (uint32_t i = depth; i > 0u; i--) {
        hit_t hit; memset(&hit, 0, sizeof hit);          /* hit_nil(), shader.wgsl:142-144 */
        c->world_hit_calls++;
        c->sphere_tests += (uint64_t)s->w->spheres.length;
        if (!world_hit(s, orig, dir, 0.001f, 1.0e4f, &hit, 0)) {
            c->paths_missed++;
            return v3_mul(attenuation, color_sky(dir.y));
        }
        scatter_t sc;
        if (!dyn_material_scatter(s, rng, dir, &hit, &sc, c)) {
            c->paths_absorbed++;
            return v3_make(0.0f, 0.0f, 0.0f);
        }
        attenuation = v3_mul(attenuation, sc.attenuation);
        orig = sc.orig;
        dir = normalize3(sc.dir);
    }
    c->paths_exhausted++;
    return v3_make(0.0f, 0.0f, 0.0f);
}

/* ------------------------------------------------------------------ camera */

/* EXTENSION: look-at thin-lens camera.  Derived in double, rounded once to f32. */
void orc_camera_derive(const orc_camera* cam, orc_camera_raw* out) {
    memset(out, 0, sizeof *out);
    out->mode = cam->mode;
    if (cam->mode == 0) return;
    double lf[3], la[3], up[3], w[3], u[3], v[3];
    for (int i = 0; i < 3; i++) { lf[i] = cam->lookfrom[i]; la[i] = cam->lookat[i]; up[i] = cam->vup[i]; }
    for (int i = 0; i < 3; i++) w[i] = lf[i] - la[i];
    double wl = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int i = 0; i < 3; i++) w[i] /= wl;
    u[0] = up[1] * w[2] - up[2] * w[1];
    u[1] = up[2] * w[0] - up[0] * w[2];
    u[2] = up[0] * w[1] - up[1] * w[0];
    double ul = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int i = 0; i < 3; i++) u[i] /= ul;
    v[0] = w[1] * u[2] - w[2] * u[1];
    v[1] = w[2] * u[0] - w[0] * u[2];
    v[2] = w[0] * u[1] - w[1] * u[0];
    const double deg = 3.14159265358979323846 / 180.0;
    double focus = cam->focus_dist;
    double s = tan(0.5 * (double)cam->vfov_deg * deg) * focus;
    double r = tan(0.5 * (double)cam->defocus_angle_deg * deg) * focus;
    out->defocus = cam->defocus_angle_deg > 0.0f;
    for (int i = 0; i < 3; i++) {
        out->origin[i] = cam->lookfrom[i];
        out->su[i] = (float)(s * u[i]);
        out->sv[i] = (float)(s * v[i]);
        out->fw[i] = (float)(focus * w[i]);
        out->ru[i] = (float)(r * u[i]);
        out->rv[i] = (float)(r * v[i]);
    }
}

/* Ray generation for one sample.  mode 0: shader.wgsl:379-381 exactly.  mode 1: the
 * same (vx,vy) viewport coordinates pushed through the derived look-at basis. */
static inline void camera_ray(const orc_camera_raw* cam, float vx, float vy, rng_t* rng, v3* orig, v3* dir) {
    if (cam->mode == 0) {
        *orig = v3_make(0.0f, 0.0f, 0.0f);
        *dir = normalize3(v3_make(vx, vy, -1.0f));
        return;
    }
    v3 p = v3_make((vx * cam->su[0] + vy * cam->sv[0]) - cam->fw[0],
                   (vx * cam->su[1] + vy * cam->sv[1]) - cam->fw[1],
                   (vx * cam->su[2] + vy * cam->sv[2]) - cam->fw[2]);
    v3 o = v3_make(cam->origin[0], cam->origin[1], cam->origin[2]);
    if (cam->defocus) {
        /* unit disk by rejection, 2 draws per try, accept |p|^2 <= 1 (mirrors shader.wgsl:86) */
        float px, py;
        do {
            float qx = rand_f32(rng); float qy = rand_f32(rng);
            px = 2.0f * qx - 1.0f; py = 2.0f * qy - 1.0f;
        } while (ORC_FMA(py, py, px * px) > 1.0f);
        v3 off = v3_make(px * cam->ru[0] + py * cam->rv[0],
                         px * cam->ru[1] + py * cam->rv[1],
                         px * cam->ru[2] + py * cam->rv[2]);
        *orig = v3_add(o, off);
        *dir = normalize3(v3_sub(p, off));
    } else {
        *orig = o;
        *dir = normalize3(p);
    }
}

/* ------------------------------------------------------------------ fs_main */

/* shader.wgsl:371-386 for the pixel (px,py), py counted from the BOTTOM row */
static void shade_pixel(const orc_locals* L, const scene_t* s, const orc_camera_raw* cam,
                        const uint32_t* seeds, const float* prev, float* out,
                        uint32_t px, uint32_t py, orc_counters* c) {
    const uint32_t W = L->shape[0], H = L->shape[1];
    const size_t p = (size_t)py * W + px;
    const float pixel_pos_x = (float)px + 0.5f, pixel_pos_y = (float)py + 0.5f;   /* vs_main :26 */
    const float pixel_side = 2.0f / (float)H;                                     /* :373 */
    const float base_x = (pixel_pos_x - 0.5f * (float)W) * pixel_side;             /* :374 */
    const float base_y = (pixel_pos_y - 0.5f * (float)H) * pixel_side;

    rng_t rng;                                                                    /* :44-47 */
    for (int k = 0; k < 4; k++) rng.s[k] = seeds[4 * p + k] ^ L->rng_shuffle[k];
    rng.draws = 0;

    uint32_t base[4];
    for (int k = 0; k < 4; k++) base[k] = rng.s[k];
    /* The reference's stream mode adds the samples one by one (:376-382).  In the counter mode (extension) the samples
     * are independent, and the sum is DEFINED blockwise so that a pixel's blocks can be rendered by different lanes:
     * colour = (..((S_0 + S_1) + S_2) ..), S_b = the sequential sum of samples [b*B, min(spp, (b+1)*B)), B = ORC_COUNTER_BLOCK.
     * Up to B samples per frame that is the same expression as the stream mode's. */
    const int counter = L->rng_mode == ORC_RNG_COUNTER;
    v3 color = v3_make(0.0f, 0.0f, 0.0f), block = v3_make(0.0f, 0.0f, 0.0f);
    for (uint32_t i = 0; i < L->samples_per_frame; i++) {                          /* :378 */
        if (counter) orc_sample_state(base, i, rng.s);                             /* extension, see header */
        float u = rand_f32(&rng); float v = rand_f32(&rng);                        /* :71-75 */
        float vx = base_x + u * pixel_side;                                        /* :379-380 */
        float vy = base_y + v * pixel_side;
        v3 orig, dir;
        camera_ray(cam, vx, vy, &rng, &orig, &dir);
        const v3 sample = color_world(s, orig, dir, L->ray_depth, &rng, c);        /* :381 */
        if (!counter) {
            color = v3_add(color, sample);
        } else {
            block = v3_add(block, sample);
            if ((i + 1u) % ORC_COUNTER_BLOCK == 0u || i + 1u == L->samples_per_frame) {
                color = (i < ORC_COUNTER_BLOCK) ? block : v3_add(color, block);    /* the first block IS the running sum */
                block = v3_make(0.0f, 0.0f, 0.0f);
            }
        }
        c->samples++;
    }
    float n = (float)L->samples_per_frame;
    color = v3_make(color.x / n, color.y / n, color.z / n);                        /* :383 */
    c->rng_draws += rng.draws;

    const float w = L->framebuffer_weight;                                         /* :385 */
    const float* q = prev + 4 * p;
    float* o = out + 4 * p;
    o[0] = mixf(color.x, q[0], w);
    o[1] = mixf(color.y, q[1], w);
    o[2] = mixf(color.z, q[2], w);
    o[3] = mixf(1.0f, q[3], w);
}

static void counters_add(orc_counters* a, const orc_counters* b) {
    uint64_t* x = (uint64_t*)a; const uint64_t* y = (const uint64_t*)b;
    for (size_t i = 0; i < sizeof(orc_counters) / sizeof(uint64_t); i++) x[i] += y[i];
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_render_rect(const orc_locals* locals, const orc_world* world,
                     const float* vec4, const float* f32, const int32_t* i32,
                     const orc_camera_raw* cam, const uint32_t* seeds,
                     const float* prev, float* out,
                     uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, int nthreads, orc_counters* counters) {
    scene_t s = {world, vec4, f32, i32};
    orc_counters total; memset(&total, 0, sizeof total);
    if (x1 <= x0 || y1 <= y0) return;
    const int64_t rw = (int64_t)(x1 - x0), n = rw * (int64_t)(y1 - y0);
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
    {
        orc_counters local; memset(&local, 0, sizeof local);
        /* pixels are independent (own stream, own texel), so any order gives the same image and the same
         * (integer) counter totals */
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int64_t k = 0; k < n; k++)
            shade_pixel(locals, &s, cam, seeds, prev, out, x0 + (uint32_t)(k % rw), y0 + (uint32_t)(k / rw), &local);
#ifdef _OPENMP
#pragma omp critical
#endif
        counters_add(&total, &local);
    }
    if (counters) counters_add(counters, &total);
}

void orc_render_rows(const orc_locals* locals, const orc_world* world,
                     const float* vec4, const float* f32, const int32_t* i32,
                     const orc_camera_raw* cam, const uint32_t* seeds,
                     const float* prev, float* out,
                     uint32_t y0, uint32_t y1, int nthreads, orc_counters* counters) {
    orc_render_rect(locals, world, vec4, f32, i32, cam, seeds, prev, out, 0u, locals->shape[0], y0, y1, nthreads, counters);
}

/* ------------------------------------------------------------------ exported unit hooks */

int orc_sphere_hit(const orc_world* w, const float* vec4, const float* f32, const int32_t* i32,
                   int32_t idx, const float orig[3], const float dir[3], float t_min, float t_sup,
                   orc_hit* out) {
    scene_t s = {w, vec4, f32, i32};
    hit_t h; memset(&h, 0, sizeof h);
    int r = sphere_hit(&s, idx, v3_make(orig[0], orig[1], orig[2]), v3_make(dir[0], dir[1], dir[2]), t_min, t_sup, &h);
    if (r && out) {
        out->at[0] = h.at.x; out->at[1] = h.at.y; out->at[2] = h.at.z; out->t = h.t;
        out->normal[0] = h.normal.x; out->normal[1] = h.normal.y; out->normal[2] = h.normal.z;
        out->front_face = h.front_face; out->ty = h.ty; out->idx = h.idx;
    }
    return r;
}
int orc_world_hit(const orc_world* w, const float* vec4, const float* f32, const int32_t* i32,
                  const float orig[3], const float dir[3], float t_min, float t_sup,
                  orc_hit* out, int32_t* hit_sphere) {
    scene_t s = {w, vec4, f32, i32};
    hit_t h; memset(&h, 0, sizeof h);
    int32_t which = -1;
    int r = world_hit(&s, v3_make(orig[0], orig[1], orig[2]), v3_make(dir[0], dir[1], dir[2]), t_min, t_sup, &h, &which);
    if (r && out) {
        out->at[0] = h.at.x; out->at[1] = h.at.y; out->at[2] = h.at.z; out->t = h.t;
        out->normal[0] = h.normal.x; out->normal[1] = h.normal.y; out->normal[2] = h.normal.z;
        out->front_face = h.front_face; out->ty = h.ty; out->idx = h.idx;
    }
    if (hit_sphere) *hit_sphere = which;
    return r;
}
/* world_hit (range [0.001, 1e4), shader.wgsl:340) for a batch of rays, plus -- for the tests of the HIP path's conservative
 * sweep -- which spheres have a discriminant that is not < 0 (the set sphere_hit goes on to take roots of). */
void orc_world_hit_batch(const orc_world* w, const float* vec4, const float* f32, const int32_t* i32,
                         const float* rays, int64_t n, int32_t* hit_sphere, float* hit_t_out, uint8_t* disc_ge0, int nthreads) {
    scene_t s = {w, vec4, f32, i32};
    const int32_t ns = w->spheres.length;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 64)
#endif
    for (int64_t r = 0; r < n; r++) {
        const v3 o = v3_make(rays[6 * r], rays[6 * r + 1], rays[6 * r + 2]), d = v3_make(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]);
        hit_t h; memset(&h, 0, sizeof h);
        int32_t which = -1;
        const int got = world_hit(&s, o, d, 0.001f, 1.0e4f, &h, &which);
        hit_sphere[r] = got ? which : -1;
        hit_t_out[r] = got ? h.t : 1.0e4f;
        if (disc_ge0)
            for (int32_t i = 0; i < ns; i++) {
                float a, b;
                const v3 ctr = sphere_load_center(&s, i);
                const float rad = sphere_load_radius(&s, i);
                const float disc = sphere_discriminant(ctr, rad, o, d, &a, &b);
                /* bit 0: discriminant not < 0.  bit 1: additionally the sphere is not entirely behind the origin, i.e. NOT
                 * (b >= +0 and c >= +0) -- with both non-negative, sqrt(d) <= b and neither root of :290-292 reaches t_min. */
                const v3 oc = v3_sub(o, ctr);
                const float c = sphere_c(oc, rad);
                uint32_t bb, cb; memcpy(&bb, &b, 4); memcpy(&cb, &c, 4);
                const int ge0 = !(disc < 0.0f), ahead = ((bb | cb) >> 31) != 0u;
                disc_ge0[r * ns + i] = (uint8_t)(ge0 | ((ge0 && ahead) << 1));
            }
    }
}

void orc_color_sky(float y, float out[3]) { v3 c = color_sky(y); out[0] = c.x; out[1] = c.y; out[2] = c.z; }

/* lib.rs:722-799: SoA packing.  vec4 = [centers(x,y,z,1) | lambertian albedos(r,g,b,1) |
 * metal albedos]; f32 = [radii | metal fuzz | dielectric ior]; i32 = [material ty | material idx] */
void orc_pack_world(const orc_sphere_aos* sp, int32_t n, orc_world* world,
                    float* vec4, int32_t* n_vec4, float* f32, int32_t* n_f32,
                    int32_t* i32, int32_t* n_i32) {
    memset(world, 0, sizeof *world);
    int32_t nl = 0, nm = 0, nd = 0;
    for (int32_t i = 0; i < n; i++) {
        if (sp[i].ty == ORC_LAMBERTIAN) nl++; else if (sp[i].ty == ORC_METAL) nm++; else if (sp[i].ty == ORC_DIELECTRIC) nd++;
    }
    int32_t v = 0, f = 0, k = 0;
    world->spheres.center_base_idx = v;
    for (int32_t i = 0; i < n; i++) { vec4[4 * v] = sp[i].center[0]; vec4[4 * v + 1] = sp[i].center[1]; vec4[4 * v + 2] = sp[i].center[2]; vec4[4 * v + 3] = 1.0f; v++; }
    world->spheres.radius_base_idx = f;
    for (int32_t i = 0; i < n; i++) f32[f++] = sp[i].radius;
    world->spheres.material_ty_base_idx = k;
    for (int32_t i = 0; i < n; i++) i32[k++] = sp[i].ty;
    world->spheres.material_idx_base_idx = k;
    { int32_t cl = 0, cm = 0, cd = 0;
      for (int32_t i = 0; i < n; i++) {
          if (sp[i].ty == ORC_LAMBERTIAN) i32[k++] = cl++;
          else if (sp[i].ty == ORC_METAL) i32[k++] = cm++;
          else if (sp[i].ty == ORC_DIELECTRIC) i32[k++] = cd++;
          else i32[k++] = 0;
      } }
    world->spheres.length = n;
    world->lambertians.albedo_base_idx = v;
    for (int32_t i = 0; i < n; i++) if (sp[i].ty == ORC_LAMBERTIAN) { vec4[4 * v] = sp[i].p[0]; vec4[4 * v + 1] = sp[i].p[1]; vec4[4 * v + 2] = sp[i].p[2]; vec4[4 * v + 3] = 1.0f; v++; }
    world->lambertians.length = nl;
    world->metals.albedo_base_idx = v;
    for (int32_t i = 0; i < n; i++) if (sp[i].ty == ORC_METAL) { vec4[4 * v] = sp[i].p[0]; vec4[4 * v + 1] = sp[i].p[1]; vec4[4 * v + 2] = sp[i].p[2]; vec4[4 * v + 3] = 1.0f; v++; }
    world->metals.fuzz_base_idx = f;
    for (int32_t i = 0; i < n; i++) if (sp[i].ty == ORC_METAL) f32[f++] = sp[i].p[3];
    world->metals.length = nm;
    world->dielectrics.ior_base_idx = f;
    for (int32_t i = 0; i < n; i++) if (sp[i].ty == ORC_DIELECTRIC) f32[f++] = sp[i].p[3];
    world->dielectrics.length = nd;
    *n_vec4 = v; *n_f32 = f; *n_i32 = k;
}
