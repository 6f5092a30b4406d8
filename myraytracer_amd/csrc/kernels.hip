// HIP kernels for gfx950 (MI355X): the per-pixel render loop of zetanumbers/myraytracer
// (raytracer/src/shader.wgsl fs_main and everything it calls), rebuilt for CDNA4.
//
// Shape of the kernels (DESIGN.md §4 has the measurements behind each choice):
//   * one lane = one pixel for the whole frame, as in the fragment shader, so the
//     reference's one-Xoshiro128+-stream-per-pixel draw order is preserved exactly;
//   * the sample loop and the bounce loop (shader.wgsl:378, :339) are flattened into ONE
//     per-lane state machine: every trip of the wave's loop is one `world_hit` for every
//     live lane, whichever sample / bounce that lane is on;
//   * `world_hit` (shader.wgsl:314-329) is split into (1) a branch-free, conservative sweep over
//     the top level of a 4-ary hierarchy of bounding spheres (clusters of <= 4 neighbouring spheres,
//     and for large scenes bounds of bounds) -- where the scene admits it as two bf16-split GEMMs per 32
//     records on the matrix cores, which also drop the bounds that lie behind the ray's origin (1 multiply
//     + 1 v_alignbit per ray and record on the VALU); otherwise from wave-uniform records fetched by scalar
//     loads into SGPRs (10 fp32 VALU ops + 1 v_alignbit); either way the signs land in per-lane bitmasks
//     kept in LDS -- and (2) a COOPERATIVE walk: the candidates of all 64 rays become work items (owner lane, node)
//     in small LDS queues and every round 64 lanes take 64 items, whoever owns them: node rounds
//     evaluate the reference's discriminant for the 4 members of a cluster (large scenes, above it: the line
//     against the axis-aligned boxes of the 4 children of an inner node, all inner levels on ONE work stack),
//     root rounds its sqrt / divide / range tests
//     (shader.wgsl:286-296) for members with disc >= 0 and merge them into the owner's slot by a
//     64-bit LDS minimum over (t, sphere index), which is what the reference's index-order scan yields;
//   * persistent waves pull 8x8 tiles from one global heaviest-first queue and a lane that
//     finishes its pixel takes the next waiting one (render_kernel); the queue may hold several LAYERS of the image --
//     the frames of a batch (mrt_render), or, in the counter-RNG mode, the blocks of 64 samples of a frame -- so that a
//     shard with few pixels, or a very short frame, still fills the chip;
//   * an iteration is: world_hit for every lane with a ray; shading; release / refill / acquire of pixels; the next
//     sample's camera ray for the lanes that need one, sharing the scatter's normalize;
//   * finalize_kernel turns the per-pixel colour sums into the framebuffer: one coalesced
//     RGBA32F store per pixel per frame, whole 128-byte lines per 8x8 tile.
//
// Arithmetic follows the "MRT-F32" rules: fma only where written, no
// contraction (-ffp-contract=off), correctly rounded sqrt and divide (hipcc's default expansions, or those
// expansions minus the operand-scaling steps where the operands cannot need them: div_unscaled, sqrt_unscaled).
// The CPU oracle under oracle/ implements the same rules independently; tests require
// bit-identical framebuffers.

#include <hip/hip_runtime.h>
#include "mrt_internal.h"
#include "mrt_device.h"

#include <type_traits>

namespace mrt {
namespace {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }

// WGSL dot(): x*x first, then fma in y, then fma in z
__device__ __forceinline__ float dot3(V3 a, V3 b) {
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}
// WGSL normalize(e) = e / length(e)
__device__ __forceinline__ V3 normalize3(V3 v) { return v / __builtin_sqrtf(dot3(v, v)); }
// WGSL reflect(e1, e2) = e1 - 2*dot(e2, e1)*e2  (shader.wgsl:230)
__device__ __forceinline__ V3 reflect3(V3 d, V3 n) {
    float k = 2.0f * dot3(n, d);
    return v3(d.x - k * n.x, d.y - k * n.y, d.z - k * n.z);
}
// WGSL mix(e1, e2, e3) = e1*(1-e3) + e2*e3
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

// ---- Xoshiro128+ (shader.wgsl:36-94) -------------------------------------------------
struct Rng { uint32_t s0, s1, s2, s3; uint32_t draws; };

__device__ __forceinline__ uint32_t rng_next(Rng& r) {          // shader.wgsl:49-64
    uint32_t result = r.s0 + r.s3;
    uint32_t t = r.s1 << 9;
    r.s2 ^= r.s0;
    r.s3 ^= r.s1;
    r.s1 ^= r.s2;
    r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = (r.s3 << 11) | (r.s3 >> 21);                          // rotl_u32(.., 11), :36-38
    return result;
}
__device__ __forceinline__ float rng_f32(Rng& r) {               // shader.wgsl:66-69
    r.draws++;
    return (float)rng_next(r) * 0x1p-32f;                        // == f32(i) / 4294967296.0
}
__device__ __forceinline__ uint32_t fmix32(uint32_t z) {         // MurmurHash3 finaliser (counter mode)
    z ^= z >> 16; z *= 0x85EBCA6Bu; z ^= z >> 13; z *= 0xC2B2AE35u; z ^= z >> 16;
    return z;
}
// 2.0 * random_f32() - 1.0 (shader.wgsl:86) in one rounding: f32(i) * 2^-32 and the doubling are exact
// (powers of two, no underflow), so the reference's value is fl(f32(i) * 2^-31 - 1) = this fma, bit for bit
__device__ __forceinline__ float rng_pm1(Rng& r) {
    r.draws++;
    return __builtin_fmaf((float)rng_next(r), 0x1p-31f, -1.0f);
}
// The reference's literal acceptance test, for the index-ordered loop over ALL spheres that rays
// with a non-finite or non-unit direction take: NaN compares false, so a NaN root is accepted.
__device__ __forceinline__ void literal_test(const SphereRec s, uint32_t idx, V3 o, V3 d, float a,
                                             float& t_sup, int32_t& best) {
    V3 oc = v3(o.x - s.cx, o.y - s.cy, o.z - s.cz);
    float b = dot3(oc, d);
    float c = __builtin_fmaf(oc.z, oc.z, __builtin_fmaf(oc.y, oc.y, __builtin_fmaf(oc.x, oc.x, s.neg_r2)));
    float disc = __builtin_fmaf(b, b, -(a * c));
    if (!(disc < 0.0f)) {
        float d_sqrt = __builtin_sqrtf(disc);
        const float t_min = 0.001f;
        float t = (-b - d_sqrt) / a;
        if (t < t_min || t_sup <= t) t = (-b + d_sqrt) / a;
        if (!(t < t_min || t_sup <= t)) { t_sup = t; best = (int32_t)idx; }
    }
}

// ---- the discriminant sweep over wave-uniform sphere records --------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
// four SphereRec = 16 dwords = one s_load_dwordx16 from the constant address space
typedef const f32x16 __attribute__((address_space(4)))* SphQuadPtr;
struct Sph8 { f32x16 lo, hi; };

// Scalar loads are issued and waited for by hand (inline asm): hipcc schedules every
// s_load of an unrolled body first and then spills the SGPRs, and its waitcnt pass can only
// emit lgkmcnt(0) -- scalar loads return out of order -- which would also wait for a
// prefetch.  An asm load is invisible to that pass, so each group is tied to its own wait:
// smem_wait() "redefines" the group, and every use of the group therefore follows the wait.
__device__ __forceinline__ void smem_load8(Sph8& g, SphQuadPtr quads, uint32_t first_sphere) {
    const SphQuadPtr p = quads + first_sphere / 4u;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40"
                 : "=&s"(g.lo), "=&s"(g.hi) : "s"(p));
}
// One statement = "group `cur` has landed; start fetching group `nxt`".  `bits` (produced by
// the previous group's tests) rides along so that those tests are scheduled BEFORE this
// point and the tests of `cur` after it, i.e. while the loads of `nxt` are in flight.
__device__ __forceinline__ void smem_wait_then_load8(Sph8& cur, Sph8& nxt, SphQuadPtr quads, uint32_t first_sphere,
                                                     uint32_t& bits) {
    const SphQuadPtr p = quads + first_sphere / 4u;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx16 %2, %5, 0x0\n\ts_load_dwordx16 %3, %5, 0x40"
                 : "+s"(cur.lo), "+s"(cur.hi), "=&s"(nxt.lo), "=&s"(nxt.hi), "+v"(bits) : "s"(p));
}
// The sweep's CONSERVATIVE line-vs-bounding-sphere test (10 fp32 VALU + 1 v_alignbit): with `ds` the
// ray direction stretched by kBoundStretch (1 + 1e-4), S = (oc.ds)^2 - (oc.oc - R^2) is >= 0 whenever the
// reference's discriminant b*b - a*c (shader.wgsl:277-282) of ANY sphere inside the bound is >= 0:
// the stretch adds >= 1.9e-4*|oc|^2 of slack against <= 1.2e-4*|oc|^2 of accumulated rounding error
// and R is 1.5 % larger than the enclosing radius (proof sketch: DESIGN.md §4).  False positives only
// cost a discriminant evaluation; a false negative cannot happen.  sign(S) is shifted into `bits`.
__device__ __forceinline__ void test1(float cx, float cy, float cz, float neg_R2, V3 o, V3 ds, uint32_t& bits) {
    const float ocx = o.x - cx, ocy = o.y - cy, ocz = o.z - cz;
    const float b = __builtin_fmaf(ocz, ds.z, __builtin_fmaf(ocy, ds.y, ocx * ds.x));
    const float c = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, neg_R2)));
    const float S = __builtin_fmaf(b, b, -c);
    bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(S), 31);       // oldest record ends in the top bit
}
__device__ __forceinline__ void test4(const f32x16 q, V3 o, V3 ds, uint32_t& bits) {
    test1(q[0], q[1], q[2], q[3], o, ds, bits);
    test1(q[4], q[5], q[6], q[7], o, ds, bits);
    test1(q[8], q[9], q[10], q[11], o, ds, bits);
    test1(q[12], q[13], q[14], q[15], o, ds, bits);
}
__device__ __forceinline__ void test8(const Sph8& g, V3 o, V3 ds, uint32_t& bits) {
    test4(g.lo, o, ds, bits);
    test4(g.hi, o, ds, bits);
}

// ---- the walk's second bound for large scenes: the axis-aligned box of the member spheres under a node -----------------
// A kd-built group of spheres on a plane fills its box, not its bounding sphere: over C5's 100 x 100 grid a ray's LINE touches
// 7.2 + 7.4 + 1.4 bounding spheres of the three levels but 1.9 + 1.9 + 0.9 boxes (experiments/bound_stats.py).  The test is the
// line against the box grown by K on every side, through the three separating axes d x e_i:
//     |p_j d_k - p_k d_j| <= e_j |d_k| + e_k |d_j| + K        p = o - centre, (i, j, k) cyclic
// (necessary and sufficient for a line and a box; the parts of the line behind the origin are left to the sphere tests).
// K = kc X + kpad, X = |p|^2 or |p|_1 (per scene), is the slack that makes it CONSERVATIVE against the reference's own
// rounding: a member whose computed discriminant is >= 0 has the line within sqrt(r^2 + 14 eps |oc|^2 / a) of its centre,
// i.e. up to min(14 eps |oc|^2 / (2 r), sqrt(14 eps) |oc|) beyond its surface, hence beyond its box; the host (api.cpp,
// build_boxes) sets kc per scene and kpad per box so that K covers 1.4143 x that for every member under the node, plus the
// test's own rounding (4 eps |p|_1; the right-hand side's three roundings are in the extents).  The kernel reads kpad FOLDED
// INTO THE EXTENTS (e + kpad: on the axis d x e_i that is a slack of kpad (|d_j| + |d_k|), which covers what "+ kpad" covered:
// api.cpp, pack_boxes) and kc from its arguments, so a box is 24 bytes.  A never-hit box has extents
// -3e38: some axis' right-hand side is then hugely negative (a unit direction has a component >= 0.57).
// 24 VALU: 3 + 3 (X) + 1 (K) + 3 x 5 + 2.
// (c, e): a BoxRec -- the centre and the half extents with kpad folded in (mrt_internal.h); kc: the scene's coefficient of X.
template <bool QUAD>
__device__ __forceinline__ uint32_t box_separated_bits(const V3 c, const V3 e, const float kc, V3 o, V3 d) {
    const float px = o.x - c.x, py = o.y - c.y, pz = o.z - c.z;
    const float X = QUAD ? __builtin_fmaf(pz, pz, __builtin_fmaf(py, py, px * px))
                         : (__builtin_fabsf(px) + __builtin_fabsf(py)) + __builtin_fabsf(pz);
    const float K = kc * X;
    const float ex = e.x, ey = e.y, ez = e.z;
    const float adx = __builtin_fabsf(d.x), ady = __builtin_fabsf(d.y), adz = __builtin_fabsf(d.z);
    const float sx = __builtin_fmaf(ey, adz, __builtin_fmaf(ez, ady, K)) - __builtin_fabsf(__builtin_fmaf(-pz, d.y, py * d.z));
    const float sy = __builtin_fmaf(ez, adx, __builtin_fmaf(ex, adz, K)) - __builtin_fabsf(__builtin_fmaf(-px, d.z, pz * d.x));
    const float sz = __builtin_fmaf(ex, ady, __builtin_fmaf(ey, adx, K)) - __builtin_fabsf(__builtin_fmaf(-py, d.x, px * d.y));
    // separated on some axis <=> some difference is negative (finite operands: never NaN): the sign bit of the result
    return __float_as_uint(sx) | __float_as_uint(sy) | __float_as_uint(sz);
}
template <bool QUAD>
__device__ __forceinline__ bool box_may_touch(const V3 c, const V3 e, const float kc, V3 o, V3 d) {
    return (int32_t)box_separated_bits<QUAD>(c, e, kc, o, d) >= 0;
}

// ---- the same conservative test on the matrix cores -------------------------------------------------
// Expanding S = (oc.ds)^2 - (oc.oc - R^2) with oc = o - C turns its two dot products into products of a
// per-record vector with a per-ray vector:
//     -(oc.ds) = C.ds - o.ds                 S = (oc.ds)^2 - o.o - U
//     U        = -2 o.C + (C.C - R^2)
// i.e. two [32 records] x [32 rays] GEMMs per tile.  The f32 MFMA runs on the vector FMA units (measured:
// no overlap with VALU work), so the GEMMs run in bf16 on the matrix cores proper, with every f32 factor
// split into bf16 pieces x = hi + lo (+ mid) and the cross products laid out along K = 16:
//     k  0..2   C_hi (x,y,z)     . v_hi        v = K ds for the first GEMM, 2 K^2 o for the second
//     k  3..5   C_hi             . v_lo
//     k  6..8   C_lo             . v_hi
//     k  9..11  (1, 1, 1)        . (-K o.ds | -K^2 o.o (minus its slack)), each as hi, mid, lo
//     k 12..14  Ck (hi, mid, lo) . (0, 0, 0 | -K^2 x (1, 1, 1))       Ck = C.C - R^2 (minus its slack)
// so ONE A operand per tile serves both; the first GEMM's result g = -K oc.ds, squared where it is positive (the
// record's centre ahead of the origin), is the C input of the second, which therefore delivers
// K^2 (max(-oc.ds, 0)^2 - U - o.o) -- K^2 S for a centre ahead, K^2 (R^2 - |oc|^2) otherwise, which drops the bounds
// that lie entirely behind the origin: one multiply and one alignbit per (ray, record).  K, a power of two chosen by
// the host so that |g| <= 1/2 (KParams::mfma_scale), only makes the multiply's clamp to [0, 1] act as max(g, 0)^2; a
// power of two changes no rounding.  What the split drops (C_lo v_lo and the remainders: 3 x 2^-18 of
// every product) and the f32 accumulation err by at most 2.5e-5 o.o + 5e-5 C.C in S; the
// test gives away 2^-13 = 1.2e-4 of o.o + C.C + R^2: o.o is scaled by 1 - 2^-13 (in mfma_scale[2]) and the host
// lowers Ck by 2^-13 (C.C + R^2) (api.cpp, build_top_mfma).  o and C are taken relative to the centre of the
// records' bounding box (P.mfma_origin; the rounding of o - origin is relative to the difference), so the
// slack does not depend on where the scene sits, only on its extent against R: the host selects this
// variant only where it is small against R^2; elsewhere the SGPR-fed sweep above runs.
// Operand layout (lane l, r = l & 31, h = l >> 5): A[record r][k = 8h + j], B[k = 8h + j][ray r], j = 0..7;
// result register i of lane l is record (i&3) + 8(i>>2) + 4h for ray r.  v_permlane32_swap(a, b) =
// {(a.lo, b.lo), (a.hi, b.hi)} builds the B operands of both 32-ray halves from a lane's own k 0..7 and
// k 8..15 words and, applied to the two halves' sign words, hands every lane the signs of its OWN ray:
// r[0] = the records with (row & 4) == 0, r[1] = the others.  The host stores the records of a tile in that
// order, so r[0] / r[1] are the masks of chunks 2t / 2t+1.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct MfmaRay { u32x4 bp[2], bu[2]; };
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {          // two round-to-nearest conversions
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_round(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ void swap32(uint32_t a, uint32_t b, uint32_t& r0, uint32_t& r1) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    r0 = r[0];
    r1 = r[1];
}
__device__ __forceinline__ void mfma_pack_ray(V3 v, float w0, float w1, float w2, uint32_t y2, uint32_t y3, u32x4 out[2]) {
    const V3 h = v3(bf16_round(v.x), bf16_round(v.y), bf16_round(v.z));
    const V3 l = v3(v.x - h.x, v.y - h.y, v.z - h.z);           // exact; rounded to bf16 by the packing below
    const uint32_t x0 = pk_bf16(h.x, h.y), x1 = pk_bf16(h.z, l.x), x2 = pk_bf16(l.y, l.z);      // k 0..5, 6..7 = x0
    const uint32_t y0 = pk_bf16(h.z, w0), y1 = pk_bf16(w1, w2);                                   // k 8..11
    uint32_t a0, a1, a2, a3, b0, b1, b2, b3;
    swap32(x0, y0, a0, b0);
    swap32(x1, y1, a1, b1);
    swap32(x2, y2, a2, b2);
    swap32(x0, y3, a3, b3);
    out[0] = u32x4{a0, a1, a2, a3};
    out[1] = u32x4{b0, b1, b2, b3};
}
// `dsk` = K x the stretched direction, o2 = o.o, s2k2 = 2 K^2, nsk2 = -(1 - 2^-13) K^2, nk2 = -K^2 as a bf16 pair, K the
// power of two of KParams::mfma_scale: the first GEMM comes out as K (C.ds - o.ds) = -K oc.ds, the second as
// -K^2 (U + o.o) with o.o lowered by its slack.  Scaling by a power of two changes no rounding.
__device__ __forceinline__ MfmaRay mfma_ray_operands(V3 o, V3 dsk, float o2, float s2k2, float nsk2, uint32_t nk2) {
    MfmaRay m;
    const float nk0 = -dot3(o, dsk);
    const float n0 = bf16_round(nk0), n1 = bf16_round(nk0 - n0), n2 = (nk0 - n0) - n1;    // hi + mid + lo, each difference exact
    mfma_pack_ray(dsk, n0, n1, n2, 0u, 0u, m.bp);
    const float k1p = o2 * nsk2;
    const float q0 = bf16_round(k1p), q1 = bf16_round(k1p - q0), q2 = (k1p - q0) - q1;
    mfma_pack_ray(v3(s2k2 * o.x, s2k2 * o.y, s2k2 * o.z), q0, q1, q2, nk2, nk2 & 0xFFFFu, m.bu);     // k 12..14: -K^2, k 15: 0
    return m;
}
// one tile of 32 records against the wave's 64 rays; `a` = this lane's 8 bf16 of the tile's A operand
// (api.cpp, build_top_mfma).  Returns the candidate mask of the tile's 32 records (record i at bit 31 - i) for this
// lane's own ray.
__device__ __forceinline__ uint32_t mfma_sweep_tile(const u32x4 a, const MfmaRay& m) {
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bf16x8 av = __builtin_bit_cast(bf16x8, a);
    uint32_t hb[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        // The first GEMM gives g = -K oc.ds: positive where the record's centre lies AHEAD of the origin.  Its square
        // becomes the C input of the second GEMM -- but only where g > 0: x |x| clamped to [0, 1] (the output modifier
        // of the same multiply; |g| <= 1/2 by the choice of K) is max(g, 0)^2 -- which then delivers
        // K^2 (max(-oc.ds, 0)^2 - U - o.o) in the same 16 registers.  For a centre ahead that is K^2 S, the stretched
        // discriminant, positive for a true candidate by the margin of the slack; for a centre not ahead it is
        // -K^2 c, c = |oc|^2 - R^2 (inflated, minus the slack): positive only if the origin lies inside the bound.
        // A bound with its centre not ahead and the origin outside lies entirely behind the origin -- no root of
        // anything inside it is positive (see the node rounds) -- and is no candidate: candidate = sign bit clear.
        // (A g that is not > 0 only through rounding has g^2 far below the slack.)
        f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, m.bp[h]), zero, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_fmed3f(acc[i] * __builtin_fabsf(acc[i]), 0.0f, 1.0f);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, m.bu[h]), acc, 0, 0, 0);
        uint32_t bb = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) bb = __builtin_amdgcn_alignbit(bb, __float_as_uint(acc[i]), 31);
        hb[h] = bb;
    }
    const auto r = __builtin_amdgcn_permlane32_swap(hb[0], hb[1], false, false);
    return ~((r[0] << 16) | (r[1] & 0xFFFFu));         // record i of the tile at bit 31 - i
}

// Candidate masks: per wave kBlockChunks / 2 x 64 lanes of u32 (one sign mask per 32 records -- two chunks, one
// matrix-core tile -- and lane), word-major so that the 64 lanes of one access touch 256 consecutive bytes.
// Diagnostic build only (-DMRT_STAMPS, scripts/phase_profile.py): s_memtime shares of the
// phases of the bounce loop, summed per wave into counters[4..].  Never in the product .so.
#ifdef MRT_STAMPS
#define MRT_STAMP(k)                                                        \
    do {                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                  \
        const uint64_t now_ = __builtin_amdgcn_s_memtime();                 \
        phase_[k] += now_ - last_;                                          \
        last_ = now_;                                                       \
        __builtin_amdgcn_sched_barrier(0);                                  \
    } while (0)
#else
#define MRT_STAMP(k) do { } while (0)
#endif

constexpr int kUnpackMore = 1;            // extra records unpacked per trip of the owners' loop (1, 2, 3 measured alike)
constexpr uint32_t kBlockChunks = 16;     // 16 chunks x 16 clusters x 4 = 1024 spheres per sweep block

// The cooperative walk: candidates found by the sweep become wave-wide work items
// (owner lane, node) in small LDS queues; every round 64 lanes take 64 items, whoever owns them.
// The bounds form a 4-ary hierarchy of P.levels levels above the member spheres (level 0);
// the sweep tests the top level's bounding spheres, the walk descends.  Small scenes (one level):
//   queue 1 : (owner, cluster)  written by the owners from their sweep masks
//   queue 0 : (owner, member)   members whose discriminant is >= 0, waiting for the root tests
// Large scenes: every owner filters its candidates against their boxes in the lane -> one stack of inner nodes of every
// level -> clusters -> members (render_kernel, "Large scenes").
// (kQueueCap, mrt_internal.h: < 64 left over + 4 x 64 pushed by one round; the small scenes' top queue: P.gen_cap)
// Large scenes keep ONE work stack of P.gen_cap entries for all inner levels (render_kernel) + this reserve: a round is sized
// so that its pushes fit (<= 4 per item), down to one item per round, which may exceed the capacity by 3 entries per level
// (kStackReserve, mrt_internal.h)
constexpr unsigned long long kNoHitKey = 0x461C4000FFFFFFFFull;   // (bits(1e4f) << 32) | -1

template <int N> struct IC { static constexpr int value = N; };
template <bool SMALL> struct Ent;
template <> struct Ent<true>  { typedef uint16_t type; static constexpr uint32_t id_bits = 10u; };
template <> struct Ent<false> { typedef uint32_t type; static constexpr uint32_t id_bits = 26u; };

// LDS of one wave: hit slots, rays, pixel FIFO, work queues, candidate masks
__host__ __device__ constexpr uint32_t lds_index_bytes(uint32_t n_members) { return (n_members * 2u + 15u) & ~15u; }     // small scenes' u16 sphere indices
__host__ __device__ constexpr uint32_t lds_off_rays() { return 0u; }                   // 64 x {ox,oy,oz,dx | dy,dz, u64 hit slot}
__host__ __device__ constexpr uint32_t lds_off_ring() { return 2048u; }                // kRingCap x u32
__host__ __device__ constexpr uint32_t lds_off_queues() { return 2048u + 512u; }
__host__ __device__ constexpr uint32_t lds_queue_bytes(bool small, uint32_t levels, uint32_t gen_cap) {
    return small ? (levels * kQueueCap + gen_cap) * 2u : (gen_cap + kStackReserve + 2u * kQueueCap) * 4u;     // large: cluster, root queues + the inner levels' stack
}
__host__ __device__ constexpr uint32_t lds_off_masks(bool small, uint32_t levels, uint32_t gen_cap) { return lds_off_queues() + lds_queue_bytes(small, levels, gen_cap); }
__host__ __device__ constexpr uint32_t lds_wave_bytes(bool small, uint32_t levels, uint32_t gen_cap, uint32_t mask_chunks) {
    return lds_off_masks(small, levels, gen_cap) + mask_chunks * 128u;
}

// inclusive prefix sum over the 64 lanes of a wave (all lanes active)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
#define MRT_DPP_ADD(ctrl, rmask) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rmask, 0xF, false)
    MRT_DPP_ADD(0x111, 0xF);      // row_shr:1
    MRT_DPP_ADD(0x112, 0xF);      // row_shr:2
    MRT_DPP_ADD(0x114, 0xF);      // row_shr:4
    MRT_DPP_ADD(0x118, 0xF);      // row_shr:8   -> scan within each row of 16
    MRT_DPP_ADD(0x142, 0xA);      // row_bcast:15 into rows 1 and 3
    MRT_DPP_ADD(0x143, 0xC);      // row_bcast:31 into rows 2 and 3
#undef MRT_DPP_ADD
    return x;
}
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }   // compiler-level only: LDS runs a wave's accesses in order
// lanes below `lane` whose bit is set in the 64-bit ballot `mask`
__device__ __forceinline__ uint32_t rank_in(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// What render_kernel leaves per pixel for finalize_kernel: the colour sum of the frame's samples
// and the pixel's cost (trips of the bounce loop).
struct alignas(16) PixAcc { float r, g, b; uint32_t cost; };

// Persistent waves and a global heaviest-first tile queue:
//  * the grid is as many single-wave workgroups as the chip holds; every wave starts at t = 0
//    and pulls 8x8 tiles from ONE global queue (an atomic counter over the tile list sorted by
//    estimated cost, heaviest first -- tile_order.hip) whenever one of its lanes would run dry;
//  * a lane owns one pixel for all of the frame's samples, as in the fragment shader -- the
//    reference's one sequential Xoshiro128+ stream per pixel (shader.wgsl:377-382) -- and when
//    the pixel is finished it takes the next pixel from the wave's small FIFO instead of idling
//    until the slowest pixel of its tile is done;
//  * finished colour sums are left in HBM; finalize_kernel turns them into the framebuffer with
//    coalesced whole-line stores.
constexpr uint32_t kCtrBlock = MRT_COUNTER_BLOCK;   // counter-RNG mode: samples per separately summed block
constexpr uint32_t kRingCap = 128;        // < 64 waiting + 64 from a new tile; power of two

// Kernel arguments that are only needed between sweeps (camera, material tables, queue and
// framebuffer pointers) are re-read from the kernarg segment where they are used: kept in SGPRs
// across the sweep they would crowd out its 64 sphere-record registers and be spilled to VGPR lanes.
typedef const KParams __attribute__((address_space(4)))* KArgPtr;
__device__ __forceinline__ KArgPtr cold_args() {
    KArgPtr p = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));      // opaque: keeps the loads at their use sites
    return p;
}

// A workgroup is kWavesPerGroup INDEPENDENT waves (each with its own FIFO, masks, queues); they share
// one thing: when the scene is small enough (SMALL), a read-only LDS copy of the member records that
// the walk gathers per item -- an L1 round trip per candidate cluster otherwise.  SMALL also means
// that every node id fits 10 bits, so the work items are u16.
constexpr uint32_t kWavesPerGroup = 4;
// DBG (mrt_debug_world_hit): the same sweep + walk for caller-supplied rays instead of camera rays -- a lane's "pixel"
// is ray number `texel` of P.dbg_rays, traced once; the winner goes to P.dbg_hit and every (ray, sphere) that
// reaches the root tests is recorded in P.dbg_cand.  Nothing else of the kernel changes.
// SC: the scene's layout -- 0 = SMALL (above), 1 / 2 = large, with the linear / the quadratic form of the box test's slack
// (api.cpp build_boxes; a compile-time choice: as a run-time flag it was two branches in every box test).
template <bool COUNT, bool PILOT, bool CTR, int SC, bool MFMA, bool DBG = false>
// Registers: small scenes run 5 workgroups per CU (their LDS footprint, 31.5 KB at C3) = 5 waves per SIMD = 96 VGPRs; large
// scenes (work queues of every level, u32 items: 33-36 KB per workgroup) fit 4 workgroups per CU whatever the kernel does, so
// they may use the 128 VGPRs of 4 waves per SIMD instead of spilling at 96.
__global__ void __launch_bounds__(64 * kWavesPerGroup) __attribute__((amdgpu_waves_per_eu(SC == 0 ? 5 : 4, 8))) render_kernel(const KParams P) {
    constexpr bool SMALL = SC == 0, QUAD = SC == 2;
    typedef typename Ent<SMALL>::type entry_t;
    constexpr uint32_t kIdBits = Ent<SMALL>::id_bits;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // small scenes always have one level (api.cpp)
    const uint32_t levels = SMALL ? 1u : P.levels;
    // (small scenes: the member records, then their sphere indices as u16 -- what a root round reads per item; from L2 the
    // index was a dependent global load in the middle of every root round)
    // (large scenes: the first P.box_lds_count boxes of the top-down numbering -- the swept top, and the level below it where it
    // fits -- which every owner's filter and the first inner rounds read: from L1 they were a third of the kernel's vector
    // memory instructions, and the texture path, not the VALU, is what a large scene's rounds wait for)
    const uint32_t nodes_bytes = SMALL ? P.n_nodes * (uint32_t)sizeof(SphereRec) + lds_index_bytes(P.n_members)
                                       : P.box_lds_count * (uint32_t)sizeof(BoxRec);
    unsigned char* const wlds = lds_raw + nodes_bytes + wave * lds_wave_bytes(SMALL, levels, P.gen_cap, P.mask_chunks);
    // lane l's 32 bytes: (ox,oy,oz,dx) (dy,dz) and the u64 slot its closest hit is min-ed into
    float4* const rays = reinterpret_cast<float4*>(wlds + lds_off_rays());
    unsigned long long* const best_slots = reinterpret_cast<unsigned long long*>(wlds + lds_off_rays()) + 3;   // slot of lane l at [4*l]
    uint32_t* const ring = reinterpret_cast<uint32_t*>(wlds + lds_off_ring());     // FIFO of waiting pixels: tile << 6 | lane-in-tile
    entry_t* const queues = reinterpret_cast<entry_t*>(wlds + lds_off_queues());   // queue k at k * kQueueCap
    uint32_t* const masks = reinterpret_cast<uint32_t*>(wlds + lds_off_masks(SMALL, levels, P.gen_cap)) + lane;   // records 32 w .. 32 w + 31 of the block at masks[w*64]
    const uint16_t* const index_lds = reinterpret_cast<const uint16_t*>(lds_raw + P.n_nodes * (uint32_t)sizeof(SphereRec));
    if (SMALL) {
        SphereRec* const dst = reinterpret_cast<SphereRec*>(lds_raw);
        const uint32_t n_rec = P.n_nodes;
        for (uint32_t i = threadIdx.x; i < n_rec; i += 64u * kWavesPerGroup) dst[i] = P.nodes[i];
        uint16_t* const di = reinterpret_cast<uint16_t*>(lds_raw + n_rec * (uint32_t)sizeof(SphereRec));
        for (uint32_t i = threadIdx.x; i < P.n_members; i += 64u * kWavesPerGroup) di[i] = (uint16_t)P.member_index[i];   // < 1,024 spheres
        __syncthreads();
    } else {
        float4* const dst = reinterpret_cast<float4*>(lds_raw);
        const float4* const src = reinterpret_cast<const float4*>(P.boxes);
        // (24-byte records, an even count: whole float4s)
        for (uint32_t i = threadIdx.x; i < (P.box_lds_count * (uint32_t)sizeof(BoxRec)) / 16u; i += 64u * kWavesPerGroup) dst[i] = src[i];
        __syncthreads();
    }

    const uint32_t H = P.locals.shape[1];
    const uint32_t spp = PILOT ? P.pilot_spp : P.locals.samples_per_frame;
    const uint32_t n_padded = P.n_padded;
    const uint32_t gen_cap = P.gen_cap;
    const SphereRec* __restrict__ spheres = P.spheres;
    const SphereRec* nodes = SMALL ? reinterpret_cast<const SphereRec*>(lds_raw) : P.nodes;   // levels 0 .. levels-1
    const uint32_t* __restrict__ member_index = P.member_index;
    const SphQuadPtr sph_quads = (SphQuadPtr)(uintptr_t)P.clusters;
    const float pixel_side = 2.0f / (float)H;                 // fs_main :373

    // pixel id q = (virtual tile << 6) | lane-in-tile -> coordinates.  A launch renders n_blocks LAYERS of the image and
    // virtual tile = layer * n_tiles + tile.  In the counter-RNG mode layer b = samples [64 b, 64 b + 64) of every pixel of
    // the frame, summed separately (the samples of a pixel are independent there); in the stream mode layer b = frame b
    // of a batch of consecutive frames (mrt_render: frames are independent until their blend), each with its own
    // rng_shuffle.  Usually there is one layer.
    auto locate = [&](KArgPtr C, uint32_t q, uint32_t& px, uint32_t& py, uint32_t& texel, uint32_t& layer) -> bool {
        uint32_t tile = q >> 6;
        const uint32_t l = q & 63u;
        layer = 0u;
        if (!PILOT && C->queue_layers > 1u) { layer = tile / C->n_tiles; tile -= layer * C->n_tiles; }
        const uint32_t tile_x = tile % C->tiles_x, band = tile / C->tiles_x;
        const uint32_t Wc = C->locals.shape[0], Hc = C->locals.shape[1];
        px = tile_x * kTileW + (l & 7u);
        py = (band * C->shard_world + C->shard_rank) * kBandRows + (l >> 3);  // global row, 0 = bottom
        texel = (band * kBandRows + (l >> 3)) * Wc + px;                      // row in this shard (< 2^32 texels)
        return px < Wc && py < Hc;
    };

    uint32_t head = 0, tail = 0, avail = 0;     // wave-uniform FIFO cursors (mod kRingCap) and fill
    bool queue_empty = false;                   // wave-uniform: the global tile queue is exhausted

    // ---- per-lane task state
    bool has_task = false, task_done = false;
    uint32_t s_done = 0, pix_trips = 0, texel = 0;
    float base_x = 0.0f, base_y = 0.0f;
    Rng rng; rng.draws = 0; rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
    uint32_t blk = 0;                                         // CTR: (layer << 7) | samples in this lane's block (<= 64)
    V3 color = v3(0.0f, 0.0f, 0.0f);
    V3 o = v3(0.0f, 0.0f, 0.0f), d = v3(0.0f, 0.0f, -1.0f), att = v3(1.0f, 1.0f, 1.0f);
    uint32_t depth_left = 0, trips = 0;
    unsigned long long started = 0, bounces = 0;     // wave totals, counted by ballot where the control flow is uniform
    unsigned long long mtests = 0;       // wave-uniform

#ifdef MRT_STAMPS
    uint64_t phase_[7] = {0, 0, 0, 0, 0, 0, 0};
    uint64_t rounds_a_ = 0, rounds_b_ = 0, items_a_ = 0, items_b_ = 0;     // wave-uniform
    uint64_t last_ = __builtin_amdgcn_s_memtime();
    const uint64_t wave_t0_ = __builtin_amdgcn_s_memrealtime();
#endif

    for (;;) {
        const bool live = has_task && !task_done;
        // One trip of the sample loop head, shader.wgsl:378-381: seeds the path (o, pre-normalised direction nd).
        // It runs at the TAIL of the iteration, for lanes whose path has just ended and for lanes that have just taken a
        // new pixel from the FIFO (a finished pixel is released and the next one acquired right before it), so that
        // the new camera ray and the scattered rays share one normalize (:354 / :381) and every lane enters the next
        // world_hit with a ray.  The wave's first iteration has nothing to trace and only acquires.  Per lane the draw
        // order is the reference's: jitter, lens, then the path's draws.
        // returns whether the lane still needs a point of the lens disk (new_sample_lens finishes the ray then)
        auto new_sample_head = [&](V3& nd, V3& p) -> bool {
            if (CTR) {      // extension: this sample's state = hash(pixel frame state, sample index)
                // (the pixel's frame state -- seed texel ^ shuffle, :44-47 -- is re-read per sample rather than held in
                // four more registers across the whole bounce loop)
                const KArgPtr Cs = cold_args();
                const uint4 sd = reinterpret_cast<const uint4*>(Cs->seeds)[texel];
                const uint32_t k4 = 4u * ((blk >> 7) * kCtrBlock + s_done);     // the sample's index within the frame
                rng.s0 = fmix32((sd.x ^ Cs->locals.rng_shuffle[0]) + 0x9E3779B9u * (k4 + 1u));
                rng.s1 = fmix32((sd.y ^ Cs->locals.rng_shuffle[1]) + 0x9E3779B9u * (k4 + 2u));
                rng.s2 = fmix32((sd.z ^ Cs->locals.rng_shuffle[2]) + 0x9E3779B9u * (k4 + 3u));
                rng.s3 = fmix32((sd.w ^ Cs->locals.rng_shuffle[3]) + 0x9E3779B9u * (k4 + 4u));
                if ((rng.s0 | rng.s1 | rng.s2 | rng.s3) == 0u) {
                    rng.s0 = 0x9E3779B9u; rng.s1 = 0x7F4A7C15u; rng.s2 = 0xBF58476Du; rng.s3 = 0x1CE4E5B9u;
                }
            }
            float u = rng_f32(rng); float v = rng_f32(rng);            // :71-75, x then y
            float vx = base_x + u * pixel_side;
            float vy = base_y + v * pixel_side;
            const KArgPtr C = cold_args();
            att = v3(1.0f, 1.0f, 1.0f);                                 // color_world :337
            depth_left = C->locals.ray_depth;
            if (C->cam.mode == 0) {
                o = v3(0.0f, 0.0f, 0.0f);                               // ORIGIN, :361
                nd = v3(vx, vy, -1.0f);                                 // :381 before normalize()
                return false;
            }
            // extension: look-at thin-lens camera over the same (vx, vy)
            p = v3((vx * C->cam.su[0] + vy * C->cam.sv[0]) - C->cam.fw[0],
                   (vx * C->cam.su[1] + vy * C->cam.sv[1]) - C->cam.fw[1],
                   (vx * C->cam.su[2] + vy * C->cam.sv[2]) - C->cam.fw[2]);
            o = v3(C->cam.origin[0], C->cam.origin[1], C->cam.origin[2]);
            nd = p;
            return C->cam.defocus != 0;
        };
        auto new_sample_lens = [&](V3& nd, V3 p, float lx, float ly) {      // (lx, ly): the accepted point of the unit disk
            const KArgPtr C = cold_args();
            const V3 off = v3(lx * C->cam.ru[0] + ly * C->cam.rv[0],
                              lx * C->cam.ru[1] + ly * C->cam.rv[1],
                              lx * C->cam.ru[2] + ly * C->cam.rv[2]);
            o = o + off;
            nd = p - off;
        };
        trips++;

        // ------------------------------------------------------------ world_hit, shader.wgsl:314-329
        const bool trace = live && depth_left != 0u;                        // lanes inside the loop of :339
        if (trace) pix_trips++;
        if (!PILOT) bounces += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(trace));
        float t_sup = 1.0e4f;                                               // :340
        int32_t best = -1;
        if (__any(trace)) {
            const float a = dot3(d, d);                                     // sphere_hit :277 (same for every sphere)
            // A ray with a non-finite component makes every discriminant NaN, which the
            // reference treats as "not < 0".  Such lanes take the literal loop below.
            // So does a direction that is not (nearly) unit length -- normalize() of an overflowed or
            // zero vector -- for which the sweep's conservative test has no proof.
            // One test covers both: a non-finite direction gives a non-finite `a`, and an origin is non-finite only
            // together with its direction -- origins are the camera's (validated: finite, |v| <= 1e7) or a hit point
            // o + t d of a finite ray and t < 1e4, and a hit point that is not finite makes the normal, hence the
            // scattered direction, NaN in the same iteration.
            // (matrix-core sweep: so does an origin further from the scene than the sweep's scaling admits -- 4 x the
            // distance of the camera or of the farthest sphere surface, api.cpp: no ray of a frame, but a caller's ray
            // under mrt_debug_world_hit may be)
            const V3 o_rel = MFMA ? v3(o.x - P.mfma_origin[0], o.y - P.mfma_origin[1], o.z - P.mfma_origin[2]) : o;
            const float o_rel2 = MFMA ? dot3(o_rel, o_rel) : 0.0f;
            const bool weird = !(a > 0.99999f && a < 1.00001f) || (MFMA && !(o_rel2 <= P.mfma_scale[3]));
            const bool usable = trace && !weird;
            const float stretch = MFMA ? P.mfma_scale[0] : kBoundStretch;
            const V3 ds = v3(d.x * stretch, d.y * stretch, d.z * stretch);
            // The few spheres far larger than the rest (a ground sphere) are outside the hierarchy: every ray does
            // sphere_hit (shader.wgsl:274-296) on them itself, record in SGPRs -- discriminant, and for lanes with
            // disc >= 0 (and the sphere not entirely behind the origin: see the node rounds) the roots and range tests
            // exactly as a root round would, sqrt and quotients by `a` in their unscaled forms for the same reason.
            // Their closest root is what the lane's hit slot starts from: a ground sphere is a candidate for most rays,
            // and as work items these were more than half of the root rounds' load.
            unsigned long long key0 = kNoHitKey;
            {
                const KArgPtr C = cold_args();
                const uint32_t n_direct = C->n_direct;
                const Divisor by_a = divisor_of(a);
                for (uint32_t j = 0; j < n_direct; j++) {
                    const float cx = C->direct[j].cx, cy = C->direct[j].cy, cz = C->direct[j].cz, nr2 = C->direct[j].neg_r2;
                    const uint32_t sidx = C->direct_index[j];
                    const float ocx = o.x - cx, ocy = o.y - cy, ocz = o.z - cz;
                    const float bq = __builtin_fmaf(ocz, d.z, __builtin_fmaf(ocy, d.y, ocx * d.x));
                    const float cq = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, nr2)));
                    const float disc = __builtin_fmaf(bq, bq, -(a * cq));
                    const bool cand = !(disc < 0.0f), ahead = (int32_t)(__float_as_uint(bq) | __float_as_uint(cq)) < 0;
                    const bool hq = usable && cand && ahead;
                    const float d_sqrt = sqrt_unscaled(disc);                     // :286
                    const float t_min = 0.001f;                                   // :340
                    const float t_near = div_unscaled(-bq - d_sqrt, by_a);        // :290
                    const float t_far = div_unscaled(-bq + d_sqrt, by_a);         // :292
                    const bool ok_near = !(t_near < t_min) && t_near < 1.0e4f;
                    const bool ok_far = !(t_far < t_min) && t_far < 1.0e4f;
                    const float t = ok_near ? t_near : t_far;
                    const unsigned long long key = (hq && (ok_near || ok_far)) ? (((unsigned long long)__float_as_uint(t) << 32) | sidx) : kNoHitKey;
                    key0 = key < key0 ? key : key0;
                    if (DBG && hq) atomicOr(P.dbg_cand + (size_t)texel * P.dbg_words + (sidx >> 5), 1u << (sidx & 31u));
                }
                if (!PILOT) mtests += (unsigned long long)n_direct * (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(usable));
            }
            // every lane leaves its ray where whoever picks up one of its work items finds it
            rays[2u * lane + 0u] = make_float4(o.x, o.y, o.z, d.x);
            rays[2u * lane + 1u] = make_float4(d.y, d.z, __uint_as_float((uint32_t)key0), __uint_as_float((uint32_t)(key0 >> 32)));
            lds_order();
            // Conservative sweep + cooperative walk, in blocks of kBlockChunks x kChunk cluster records.
            // The records are wave-uniform: they are fetched with scalar loads, 8 records (two
            // s_load_dwordx16 = 32 SGPRs) per group, double-buffered: wait for group g, issue the
            // loads of group g+1, then run the 8 x 11 VALU ops of group g while they fly.
            uint32_t bits = 0;      // running sign history; its low 16 (or 8) bits are the current chunk
            // MFMA variant of the sweep: per-ray operands of the two GEMMs, rays x records (see mfma_sweep_tile)
            MfmaRay mr;
            if (MFMA) mr = mfma_ray_operands(o_rel, ds, o_rel2, P.mfma_scale[1], P.mfma_scale[2], P.mfma_neg_k2_pair);
            for (uint32_t blk = 0; blk < n_padded; blk += kBlockChunks * kChunk) {
                const uint32_t blk_end = (blk + kBlockChunks * kChunk < n_padded) ? blk + kBlockChunks * kChunk : n_padded;
                uint32_t nz = 0;                                        // bit w: records 32 w .. 32 w + 31 of this block hold a candidate
                uint32_t rem = 0;                                       // this lane's candidate clusters in the block
                if (MFMA) {
                    // 32 records per tile = one mask word per lane and ray
                    uint32_t w = 0;
                    for (uint32_t i = blk; i < blk_end; i += 2u * kChunk, w++) {
                        const uint32_t m = mfma_sweep_tile(reinterpret_cast<const u32x4*>(P.top_mfma)[(size_t)(i / 32u) * 64u + lane], mr);
                        masks[w * 64u] = m;
                        nz |= (m < 1u ? m : 1u) << w;
                        rem += (uint32_t)__builtin_popcount(m);
                    }
                } else {
                // the 64 record SGPRs are live only during the block's sweep, not during its walk
                Sph8 ga, gb;
                smem_load8(ga, sph_quads, blk);
                asm volatile("" : "=s"(gb.lo), "=s"(gb.hi));   // defined (uniform) on every path to the block's final wait
                uint32_t c = 0;
                for (uint32_t i = blk; i < blk_end; i += kChunk, c++) {
                    // (the record count is a multiple of 32 -- api.cpp pads the top level to whole matrix-core tiles --
                    // so every chunk is full and chunks come in pairs)
                    smem_wait_then_load8(ga, gb, sph_quads, i + 8u, bits);  test8(ga, o, ds, bits);
                    const uint32_t nxt = (i + kChunk < n_padded) ? i + kChunk : 0u;   // next chunk, or a harmless reload
                    smem_wait_then_load8(gb, ga, sph_quads, nxt, bits);
                    test8(gb, o, ds, bits);
                    if (c & 1u) {
                        // 32 signs, record i - 16 (the even chunk's first) at bit 31; candidate = S >= 0
                        const uint32_t m = ~bits;
                        masks[(c >> 1) * 64u] = m;
                        nz |= (m < 1u ? m : 1u) << (c >> 1);
                        rem += (uint32_t)__builtin_popcount(m);
                    }
                }
                // the last prefetch is never consumed, but its destination SGPRs must stay reserved
                // until it has landed
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ga.lo), "+s"(ga.hi), "+s"(gb.lo), "+s"(gb.hi));
                }
                MRT_STAMP(1);
                if (!usable) { nz = 0; rem = 0; }         // idle lanes; "weird" lanes take the literal loop below
                lds_order();

                // ---- cooperative walk over this block's candidates
                uint32_t wm = 0, ebase = 0;               // owner side: see the unpacking loop
                uint32_t incl = wave_incl_scan(rem);
                uint32_t total_rem = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                // owners unpack their masks into `n_new` (owner, top record) items at their scanned positions behind dst[0];
                // wm: the 32-record mask word being unpacked (clz = record within the word); ebase: owner bits | first record
                // id of that word.  `room` >= n_new entries are free behind dst.
                auto unpack = [&](auto* const dst, const uint32_t room, const uint32_t owner_shift) -> uint32_t {
                    typedef typename std::remove_pointer<decltype(dst)>::type item_t;
                    const uint32_t excl = incl - rem;
                    const uint32_t n_new = total_rem < room ? total_rem : room;
                    item_t* wp = dst + excl;
                    auto refill = [&]() {
                        const uint32_t cc = (uint32_t)__builtin_ctz(nz);
                        nz &= nz - 1u;
                        wm = masks[cc * 64u];
                        ebase = (lane << owner_shift) | (blk + cc * 2u * kChunk);
                    };
                    if (total_rem <= room) {             // the usual case: everything fits, no bound to watch
                        while ((nz | wm) != 0u) {
                            if (wm == 0u) refill();
                            const uint32_t j = (uint32_t)__builtin_clz(wm);
                            wm ^= 0x80000000u >> j;
                            *wp++ = (item_t)(ebase + j);
                            // further records of the same word in the same trip: fewer trips, i.e. fewer taken branches
#pragma unroll
                            for (int more = 0; more < kUnpackMore; more++) {
                                if (wm != 0u) {
                                    const uint32_t j2 = (uint32_t)__builtin_clz(wm);
                                    wm ^= 0x80000000u >> j2;
                                    *wp++ = (item_t)(ebase + j2);
                                }
                            }
                        }
                    } else {
                        item_t* const wend = dst + n_new;
                        while ((nz | wm) != 0u && wp < wend) {
                            if (wm == 0u) refill();
                            const uint32_t j = (uint32_t)__builtin_clz(wm);
                            wm ^= 0x80000000u >> j;
                            *wp++ = (item_t)(ebase + j);
                        }
                    }
                    rem -= (uint32_t)(wp - (dst + excl));
                    total_rem -= n_new;
                    if (total_rem != 0u) incl = wave_incl_scan(rem);
                    lds_order();
                    MRT_STAMP(6);
                    return n_new;
                };
                // root round: the reference's sqrt / divide / range tests (shader.wgsl:286-296) for up to 64 (owner, member)
                // items.  Each member's root goes into its owner's slot by a 64-bit unsigned minimum of
                // (bits(t) << 32 | sphere index): t > 0, so this is the lexicographic minimum of (t, index) over spheres with
                // a root in [0.001, 1e4) -- what the reference's index-order scan with `t_sup <= t` (:291-296) ends with
                // (`near` is tried first, `far` only if `near` is out of range: near >= t_sup implies far >= t_sup).
                auto root_round = [&](const entry_t* const src, uint32_t& n) {
                    const uint32_t take = n < 64u ? n : 64u, start = n - take;
                    const bool act = lane < take;
                    const uint32_t it = src[act ? start + lane : 0u];
                    const uint32_t owner = it >> kIdBits, node = it & ((1u << kIdBits) - 1u);
                    const float4 r0 = rays[2u * owner];
                    const float2 r1 = *reinterpret_cast<const float2*>(rays + 2u * owner + 1u);
                    const V3 ro = v3(r0.x, r0.y, r0.z), rd = v3(r0.w, r1.x, r1.y);
                    const float ra = dot3(rd, rd);                      // the owner's `a`, same expression
                    const SphereRec sm = nodes[node];
                    const uint32_t sidx = SMALL ? (uint32_t)index_lds[node] : member_index[node];
                    const float ocx = ro.x - sm.cx, ocy = ro.y - sm.cy, ocz = ro.z - sm.cz;
                    const float bq = __builtin_fmaf(ocz, rd.z, __builtin_fmaf(ocy, rd.y, ocx * rd.x));
                    const float cq = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, sm.neg_r2)));
                    const float disc = __builtin_fmaf(bq, bq, -(ra * cq));
                    // sqrt and the two divisions by `ra` without operand scaling: ra is within 1e-5 of 1 (`weird` rays
                    // own no items), so scaling could only act on a numerator below 2^-102 or above 2^95 -- roots that
                    // fail the range test below whatever their last bits -- and on disc < 2^-96, a d_sqrt below 2^-47
                    // that changes -bq -+ d_sqrt only where that sum is below 2^-19 < t_min.
                    const float d_sqrt = sqrt_unscaled(disc);                     // :286
                    const float t_min = 0.001f;                                   // :340
                    const Divisor by_a = divisor_of(ra);
                    const float t_near = div_unscaled(-bq - d_sqrt, by_a);        // :290
                    const float t_far = div_unscaled(-bq + d_sqrt, by_a);         // :292
                    const bool ok_near = !(t_near < t_min) && t_near < 1.0e4f;
                    const bool ok_far = !(t_far < t_min) && t_far < 1.0e4f;
                    const float t = ok_near ? t_near : t_far;
                    // (unconditional, with a key that changes nothing for lanes without a root: no branch
                    // for the index load to hide behind)
                    const unsigned long long key = (act && (ok_near || ok_far)) ? (((unsigned long long)__float_as_uint(t) << 32) | sidx) : kNoHitKey;
                    atomicMin(best_slots + 4u * owner, key);
                    if (DBG) {      // this (ray, sphere) passed the sweep, the walk and the exact discriminant
                        const uint32_t owner_ray = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)texel);
                        if (act) atomicOr(P.dbg_cand + (size_t)owner_ray * P.dbg_words + (sidx >> 5), 1u << (sidx & 31u));
                    }
                    n = start;
#ifdef MRT_STAMPS
                    rounds_b_++; items_b_ += take;
#endif
                    lds_order();
                    MRT_STAMP(3);
                };
                // The reference's discriminant (shader.wgsl:274-282) for the 4 members of a cluster; members with disc >= 0
                // become (owner, member) items at dst[dn ..].  A member that lies entirely behind the ray's origin -- origin
                // outside it (cq >= 0) and its centre not ahead (bq >= 0): then sqrt(disc) <= bq, both roots of :290-292 are
                // <= 0 and fail t >= 0.001 (:291) -- can never be the hit and is dropped here.  Both conditions from sign
                // bits, in one three-input bit operation and one comparison: the discriminant of a finite ray against finite
                // geometry is never NaN, and never -0 (b*b - a*c cancels to +0), so "not < 0" is "sign bit clear".
                // `on`: the lanes whose item this is; returns the number of items pushed.
                auto members_round = [&](const SphereRec* const ch, const V3 ro, const V3 rd, const float ra, const bool on,
                                         const uint32_t e0, entry_t* const dst) -> uint32_t {
                    const SphereRec sr[4] = {ch[0], ch[1], ch[2], ch[3]};
                    bool h[4];
                    unsigned long long hm[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float ocx = ro.x - sr[q].cx, ocy = ro.y - sr[q].cy, ocz = ro.z - sr[q].cz;
                        const float bq = __builtin_fmaf(ocz, rd.z, __builtin_fmaf(ocy, rd.y, ocx * rd.x));
                        const float cq = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, sr[q].neg_r2)));
                        const float disc = __builtin_fmaf(bq, bq, -(ra * cq));
                        h[q] = (int32_t)((__float_as_uint(bq) | __float_as_uint(cq)) & ~__float_as_uint(disc)) < 0;
                        hm[q] = __builtin_amdgcn_ballot_w64(h[q]);
                    }
                    const unsigned long long on_mask = __builtin_amdgcn_ballot_w64(on);
                    const unsigned long long m0 = hm[0] & on_mask, m1 = hm[1] & on_mask, m2 = hm[2] & on_mask, m3 = hm[3] & on_mask;
                    // About one member in twenty passes (C3: 43 of 868 per world_hit), so a lane hardly ever holds two: when none
                    // does -- decided on the scalar side -- ONE ranked push serves the round instead of four.
                    const unsigned long long any = m0 | m1 | m2 | m3;
                    const unsigned long long twice = (m0 & m1) | ((m0 | m1) & m2) | ((m0 | m1 | m2) & m3);
                    if (twice == 0ull) {
                        const uint32_t q = h[1] ? 1u : h[2] ? 2u : h[3] ? 3u : 0u;
                        if ((h[0] || h[1] || h[2] || h[3]) && on) dst[rank_in(any)] = (entry_t)(e0 + q);
                        return (uint32_t)__popcll(any);
                    }
                    uint32_t pushed = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const unsigned long long mk = hm[q] & on_mask;
                        if (h[q] && on) dst[pushed + rank_in(mk)] = (entry_t)(e0 + (uint32_t)q);
                        pushed += (uint32_t)__popcll(mk);
                    }
                    return pushed;
                };
                if constexpr (SMALL) {
                    // Small scenes (one level): queue 1 = (owner, cluster) from the owners' masks, queue 0 = (owner, member)
                    // waiting for the root tests.  A full round at the deeper queue that has one; else refill queue 1; else
                    // a partial round at queue 1, then queue 0.
                    uint32_t qn0 = 0, qn1 = 0;
                    entry_t* const q0 = queues, * const q1 = queues + kQueueCap;
                    for (;;) {
                        int k = qn0 >= 64u ? 0 : qn1 >= 64u ? 1 : -1;
                        if (k < 0) {
                            if (total_rem != 0u) { qn1 += unpack(q1 + qn1, gen_cap - qn1, kIdBits); continue; }
                            k = qn1 != 0u ? 1 : qn0 != 0u ? 0 : -1;
                            if (k < 0) break;
                        }
                        if (k == 0) { root_round(q0, qn0); continue; }
                        // node round: 64 lanes take the last 64 (owner, cluster) items, whoever owns them
                        // (Round 1 rotated the four reads by node/4 to spread one read's 64 lanes over all LDS banks; the 8
                        // VALU of address arithmetic per round cost more than the bank conflicts they avoided: round 2.)
                        const uint32_t take = qn1 < 64u ? qn1 : 64u, start = qn1 - take;
                        const bool act = lane < take;
                        const uint32_t it = q1[act ? start + lane : 0u];
                        const uint32_t owner = it >> kIdBits, node = it & ((1u << kIdBits) - 1u);
                        const float4 r0 = rays[2u * owner];
                        const float2 r1 = *reinterpret_cast<const float2*>(rays + 2u * owner + 1u);
                        const V3 ro = v3(r0.x, r0.y, r0.z), rd = v3(r0.w, r1.x, r1.y);
                        qn0 += members_round(nodes + 4u * node, ro, rd, dot3(rd, rd), act, (owner << kIdBits) | (4u * node), q0 + qn0);
                        qn1 = start;
                        if (!PILOT) mtests += kClusterK * take;
#ifdef MRT_STAMPS
                        rounds_a_++; items_a_ += take;
#endif
                        lds_order();
                        MRT_STAMP(2);
                    }
                } else {
                    // Large scenes.  Every owner first FILTERS its own candidates -- does its line touch the record's BOX (the
                    // sweep tested its bounding sphere)? -- in the lane: one record of its masks per trip, its ray in registers,
                    // the record's box (from the group's LDS copy of the upper levels) one trip ahead.  (Round 3 wrote the raw candidates to a queue and read them back
                    // in filter rounds: two LDS round trips and a round's fixed costs per candidate more; C5 +7 %.)  The
                    // survivors feed three queues, deepest first:
                    //   inner    ONE stack for every level below the top: (owner, node g), g numbered top-down over the complete
                    //            4-ary tree (children of g = 4 g + P.n_padded .. + 3, whatever the level): test the 4 children's
                    //            boxes.  A round takes the last 64 items whatever their levels -- the code is the same.
                    //   cluster  (owner, cluster-level node g): the reference's discriminant for the cluster's 4 members
                    //            (level 0, members 4 m .. 4 m + 3, m = g - P.box_cluster_first)
                    //   root     (owner, member): the root tests
                    // A full round at the deepest queue that has one; else filter more candidates; else a partial round at the
                    // HIGHEST queue that holds anything (what it pushes may still fill the queues below).
                    constexpr uint32_t kIndexMask = (1u << 26) - 1u;
                    uint32_t sn = 0, cn = 0, rn = 0;
                    uint32_t* const clusterq = queues;                                // kQueueCap entries
                    uint32_t* const rootq = queues + kQueueCap;                       // kQueueCap
                    uint32_t* const stack = queues + 2u * kQueueCap;                  // gen_cap + kStackReserve
                    // its first box_lds boxes, in LDS (address-space qualified: as two generic pointers hipcc merged the owners' two
                    // sources into ONE flat load of a selected pointer, which goes down the texture path whatever it reads)
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    typedef const f32x4 __attribute__((address_space(3)))* LdsBoxPtr;
                    typedef const f32x4 __attribute__((address_space(1)))* GlobalBoxPtr;
                    typedef const f32x2 __attribute__((address_space(3)))* LdsBox2Ptr;          // one box = 3 x 8 bytes, 8-byte aligned
                    typedef const f32x2 __attribute__((address_space(1)))* GlobalBox2Ptr;
                    auto f4 = [](const f32x4 v) { return make_float4(v.x, v.y, v.z, v.w); };
                    const LdsBoxPtr bxs_lds = (LdsBoxPtr)lds_raw;
                    const GlobalBoxPtr bxs_g = (GlobalBoxPtr)P.boxes;
                    const LdsBox2Ptr bx2_lds = (LdsBox2Ptr)lds_raw;
                    const GlobalBox2Ptr bx2_g = (GlobalBox2Ptr)P.boxes;
                    const float box_kc = P.box_kc;
                    const uint32_t box_lds = P.box_lds_count;
                    const bool top_in_lds = box_lds >= n_padded;
                    const uint32_t cluster_parent_first = P.box_cluster_parent_first;
                    for (;;) {
                        int k = rn >= 64u ? 0 : cn >= 64u ? 1 : sn >= 64u ? 2 : -1;
                        if (k < 0) {
                            // the owners' filter: until two rounds' worth of survivors wait or nothing is left
                            if (__any((nz | wm) != 0u)) {
                                // (the queue's fill in a variable of its own: as a reference to `cn` or `sn` it made hipcc treat both
                                // -- and with them the whole round scheduling below -- as per-lane values in VGPRs)
                                uint32_t qn = levels == 1u ? cn : sn;
                                uint32_t* const dst = levels == 1u ? clusterq : stack;
                                const uint32_t owner_bits = lane << 26;
                                // the next record of this lane's masks with bit 31 set, or 0 when the lane has none left
                                auto next_candidate = [&]() -> uint32_t {
                                    uint32_t code = 0u;
                                    if ((nz | wm) != 0u) {
                                        if (wm == 0u) {
                                            const uint32_t cc = (uint32_t)__builtin_ctz(nz);
                                            nz &= nz - 1u;
                                            wm = masks[cc * 64u];
                                            ebase = blk + cc * 2u * kChunk;
                                        }
                                        const uint32_t j = (uint32_t)__builtin_clz(wm);
                                        wm ^= 0x80000000u >> j;
                                        code = 0x80000000u | (ebase + j);
                                    }
                                    return code;
                                };
                                auto fetch_box = [&](const uint32_t code, V3& bc, V3& be) {
                                    const uint32_t g = code & 0x7FFFFFFFu;
                                    f32x2 w0, w1, w2;
                                    if (top_in_lds) { w0 = bx2_lds[3u * g]; w1 = bx2_lds[3u * g + 1u]; w2 = bx2_lds[3u * g + 2u]; }
                                    else { w0 = bx2_g[3u * (size_t)g]; w1 = bx2_g[3u * (size_t)g + 1u]; w2 = bx2_g[3u * (size_t)g + 2u]; }
                                    bc = v3(w0.x, w0.y, w1.x);
                                    be = v3(w1.y, w2.x, w2.y);
                                };
                                // a candidate whose box the ray's line may touch goes on the queue (one comparison decides both: the
                                // separation's sign bit, or the missing candidate bit)
                                auto test_push = [&](const uint32_t code, const V3 bc, const V3 be) {
                                    const bool keep = (int32_t)(box_separated_bits<QUAD>(bc, be, box_kc, o, d) | ~code) >= 0;
                                    const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
                                    if (keep) dst[qn + rank_in(km)] = owner_bits | (code & 0x7FFFFFFFu);
                                    qn += (uint32_t)__popcll(km);
                                };
                                auto some_left = [&](const uint32_t code) -> bool { return __builtin_amdgcn_ballot_w64((int32_t)code < 0) != 0ull; };
                                // One box ahead: the next trip's box is requested before this trip's is tested.  Two trips per turn of
                                // the loop, on alternating register sets: carried over a single-trip loop the box fetched ahead had to be
                                // COPIED at the end of every trip, which also waited for it there
                                uint32_t ca = next_candidate(), cb, left;
                                V3 a0, a1, b0, b1;
                                fetch_box(ca, a0, a1);
                                for (;;) {
                                    cb = next_candidate();
                                    fetch_box(cb, b0, b1);
                                    test_push(ca, a0, a1);
                                    if (!(some_left(cb) && qn < 128u)) { left = cb; break; }
                                    ca = next_candidate();
                                    fetch_box(ca, a0, a1);
                                    test_push(cb, b0, b1);
                                    if (!(some_left(ca) && qn < 128u)) { left = ca; break; }
                                }
                                // (stopped with a candidate fetched but not tested: it goes back into the lane's mask word)
                                if ((int32_t)left < 0) wm |= 0x80000000u >> ((left & 0x7FFFFFFFu) - ebase);
                                if (levels == 1u) cn = qn; else sn = qn;
                                lds_order();
                                MRT_STAMP(6);
                                continue;
                            }
                            k = sn != 0u ? 2 : cn != 0u ? 1 : rn != 0u ? 0 : -1;
                            if (k < 0) break;
                        }
                        if (k == 0) { root_round(rootq, rn); continue; }
                        const uint32_t n = k == 1 ? cn : sn;
                        uint32_t take = n < 64u ? n : 64u;
                        if (k == 2) {
                            // an inner round pushes at most 4 items per item it pops: never beyond the stack (+ its reserve)
                            const uint32_t fit = gen_cap > sn ? (gen_cap - sn) / 3u : 0u;
                            take = take <= fit ? take : (fit ? fit : 1u);
                        }
                        const uint32_t start = n - take;
                        const bool act = lane < take;
                        const uint32_t it = (k == 1 ? clusterq : stack)[act ? start + lane : 0u];
                        const uint32_t owner = it >> 26, g = act ? it & kIndexMask : 0u;
                        const float4 r0 = rays[2u * owner];
                        const float2 r1 = *reinterpret_cast<const float2*>(rays + 2u * owner + 1u);
                        const V3 ro = v3(r0.x, r0.y, r0.z), rd = v3(r0.w, r1.x, r1.y);
                        if (k == 1) {
                            const uint32_t m4 = act ? 4u * (g - P.box_cluster_first) : 0u;
                            rn += members_round(nodes + m4, ro, rd, dot3(rd, rd), act, (owner << 26) | m4, rootq + rn);
                            cn = start;
                            if (!PILOT) mtests += kClusterK * take;
                        } else {
                            const uint32_t c0 = 4u * g + n_padded;                    // the first child, in the same numbering
                            // (from LDS when the children of every item of the round are among the boxes held there)
                            // (the four children: 4 x 24 = 96 contiguous bytes, 16-byte aligned -- c0 is a multiple of 4)
                            float4 b[6];
                            if (__builtin_amdgcn_ballot_w64(act && c0 + 4u > box_lds) == 0ull) {
                                const LdsBoxPtr bx = bxs_lds + 6u * (c0 >> 2);
#pragma unroll
                                for (int q = 0; q < 6; q++) b[q] = f4(bx[q]);
                            } else {
                                const GlobalBoxPtr bx = bxs_g + 6u * (size_t)(c0 >> 2);
#pragma unroll
                                for (int q = 0; q < 6; q++) b[q] = f4(bx[q]);
                            }
                            const V3 bc[4] = {v3(b[0].x, b[0].y, b[0].z), v3(b[1].z, b[1].w, b[2].x), v3(b[3].x, b[3].y, b[3].z), v3(b[4].z, b[4].w, b[5].x)};
                            const V3 be[4] = {v3(b[0].w, b[1].x, b[1].y), v3(b[2].y, b[2].z, b[2].w), v3(b[3].w, b[4].x, b[4].y), v3(b[5].y, b[5].z, b[5].w)};
                            bool h[4];
                            unsigned long long hm[4];
                            const unsigned long long act_mask = __builtin_amdgcn_ballot_w64(act);
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                h[q] = box_may_touch<QUAD>(bc[q], be[q], box_kc, ro, rd);
                                hm[q] = __builtin_amdgcn_ballot_w64(h[q]) & act_mask;
                            }
                            sn = start;
                            // the children of a node of the level above the clusters are clusters: they go to the cluster queue,
                            // the others back on the stack.  A round is mostly of one level (children arrive on top in bulk)
                            const bool to_c = g >= cluster_parent_first;
                            const unsigned long long m_c = __builtin_amdgcn_ballot_w64(to_c) & act_mask, m_s = act_mask & ~m_c;
                            const uint32_t e0 = (owner << 26) | c0;
                            auto push4 = [&](uint32_t* const dst, const unsigned long long who, const bool mine) -> uint32_t {
                                uint32_t pushed = 0;
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    const unsigned long long mk = hm[q] & who;
                                    if (h[q] && mine) dst[pushed + rank_in(mk)] = e0 + (uint32_t)q;
                                    pushed += (uint32_t)__popcll(mk);
                                }
                                return pushed;
                            };
                            if (m_s != 0ull) sn += push4(stack + sn, m_s, act && !to_c);
                            if (m_c != 0ull) cn += push4(clusterq + cn, m_c, act && to_c);
                        }
#ifdef MRT_STAMPS
                        rounds_a_++; items_a_ += take;
#endif
                        lds_order();
                        MRT_STAMP(2);
                    }
                }
            }
            if (trace) {
                if (weird) {
                    for (uint32_t idx = 0; idx < P.n_spheres; idx++)
                        literal_test(spheres[idx], idx, o, d, a, t_sup, best);
                } else {
                    const unsigned long long key = best_slots[4u * lane];
                    t_sup = __uint_as_float((uint32_t)(key >> 32));
                    best = (int32_t)(uint32_t)key;
                }
            }
        }
        MRT_STAMP(3);
        if (DBG && live) {
            P.dbg_hit[2u * texel] = best;
            P.dbg_hit[2u * texel + 1u] = (int32_t)__float_as_uint(t_sup);
            task_done = true;
        }

        // ---- paths that end without a scatter: loop :339 not entered (depth exhausted, :357) or a miss (sky, :343-345).
        // They are known before any shading, so these lanes release / start their next sample in this very iteration and
        // their camera-ray draws share the rejection loop below with the hit lanes' unit-ball draws.
        bool start_sample = false;
        V3 ndir = d;
        const bool hit = !DBG && live && depth_left != 0u && best >= 0;
        if (!DBG && live && !hit) {
            if (depth_left != 0u) {
                // color_sky, shader.wgsl:331-334, 343-345
                float t = 0.5f * d.y + 0.5f;
                color = color + att * v3(mixf(1.0f, 0.5f, t), mixf(1.0f, 0.7f, t), mixf(1.0f, 1.0f, t));     // :381
            }                                                                   // else: + vec3(0), :357
            s_done++;
            if (s_done < (CTR ? (blk & 127u) : spp)) start_sample = true;
            else if (!CTR && !PILOT && blk + 1u < cold_args()->lane_frames) {
                // Stream mode, a batch of SHORT frames (mrt_render): the lane keeps its pixel for all frames of the batch --
                // frame blk is done: park its colour sum in layer blk and start frame blk + 1 of the same pixel from the
                // seed texel and that frame's own rng_shuffle (xoshiro128plus_load :44-47), exactly the state a launch of its
                // own would have started from.  One acquisition (queue atomic, tile order, seed fetch from HBM) per pixel and
                // batch instead of one per pixel and frame.
                const KArgPtr C = cold_args();
                PixAcc sa; sa.r = color.x; sa.g = color.y; sa.b = color.z; sa.cost = pix_trips;
                reinterpret_cast<PixAcc*>(C->pix_acc)[(size_t)blk * C->pix_stride + texel] = sa;
                blk++;
                const uint4 sd = reinterpret_cast<const uint4*>(C->seeds)[texel];
                rng.s0 = sd.x ^ C->layer_shuffle[blk][0];
                rng.s1 = sd.y ^ C->layer_shuffle[blk][1];
                rng.s2 = sd.z ^ C->layer_shuffle[blk][2];
                rng.s3 = sd.w ^ C->layer_shuffle[blk][3];
                color = v3(0.0f, 0.0f, 0.0f);
                s_done = 0;
                pix_trips = 0;
                start_sample = true;
            } else task_done = true;
        }
        MRT_STAMP(4);

        // ---- release: the pixel's last sample is done -> leave its colour sum for finalize_kernel
        const bool release = has_task && task_done;
        if (release) {
            if (!DBG) {
                PixAcc sa; sa.r = color.x; sa.g = color.y; sa.b = color.z; sa.cost = pix_trips;
                const KArgPtr C = cold_args();
                // (counter mode: the block's layer; stream mode: 0, or the last frame of a batch the lane rendered in one go)
                reinterpret_cast<PixAcc*>(C->pix_acc)[(size_t)(CTR ? (blk >> 7) : blk) * C->pix_stride + texel] = sa;
            }
            has_task = false;
        }
        // ---- refill: lanes would run dry -> take the next (heaviest remaining) tile of the frame
        const unsigned long long need = __ballot(!has_task);
        if (!queue_empty && avail < (uint32_t)__popcll(need)) {
            const KArgPtr C = cold_args();
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(C->tile_queue, 1u);
            t = __builtin_amdgcn_readfirstlane(t);
            const uint32_t n_layers = PILOT ? 1u : C->queue_layers;
            if (t >= C->n_tiles * n_layers) {
                queue_empty = true;
            } else {
                // layer by layer, each in the heaviest-first order of the tiles
                const uint32_t lay = PILOT ? 0u : t / C->n_tiles, ti = t - lay * C->n_tiles;
                const uint32_t tile = lay * C->n_tiles + (C->tile_order ? C->tile_order[ti] : ti);
                uint32_t fx, fy, ft, fl;
                const uint32_t fq = (tile << 6) | lane;
                const bool ok = locate(C, fq, fx, fy, ft, fl);
                const unsigned long long m = __ballot(ok);
                if (ok) ring[(tail + rank_in(m)) & (kRingCap - 1u)] = fq;
                const uint32_t nf = (uint32_t)__popcll(m);
                tail += nf;
                avail += nf;
            }
        }
        // ---- acquire: idle lanes take the pixels at the front of the FIFO
        if (need != 0ull && avail != 0u) {
            const uint32_t rk = rank_in(need);
            const bool take = !has_task && rk < avail;
            if (take) {
                const KArgPtr C = cold_args();
                uint32_t px, py, layer;
                locate(C, ring[(head + rk) & (kRingCap - 1u)], px, py, texel, layer);
                const float Wf = (float)C->locals.shape[0], Hf = (float)C->locals.shape[1];
                base_x = (((float)px + 0.5f) - 0.5f * Wf) * pixel_side;            // fs_main :374
                base_y = (((float)py + 0.5f) - 0.5f * Hf) * pixel_side;
                const uint4 sd = reinterpret_cast<const uint4*>(C->seeds)[texel];     // xoshiro128plus_load :44-47
                // the frame's rng_shuffle; in a batch of frames (stream mode) the layer's own
                const uint32_t li = CTR ? 0u : layer;
                rng.s0 = sd.x ^ C->layer_shuffle[li][0];
                rng.s1 = sd.y ^ C->layer_shuffle[li][1];
                rng.s2 = sd.z ^ C->layer_shuffle[li][2];
                rng.s3 = sd.w ^ C->layer_shuffle[li][3];
                if (!CTR) texel += layer * C->pix_stride;      // where this (frame, pixel)'s colour sum goes
                color = v3(0.0f, 0.0f, 0.0f);                                         // :376
                s_done = 0;
                pix_trips = 0;
                has_task = true;
                blk = 0u;       // stream mode: the frame of the batch this lane is on (lane_frames > 1), else unused
                if (CTR) {      // this lane's block of the pixel's samples
                    const uint32_t first = layer * kCtrBlock;
                    const uint32_t cnt = spp > first ? (spp - first < kCtrBlock ? spp - first : kCtrBlock) : 0u;
                    blk = (layer << 7) | cnt;
                }
                task_done = (spp == 0u);                // nothing to draw: colour 0/0, as the reference
                start_sample = !task_done;
                if (DBG) {
                    const float* const r6 = C->dbg_rays + 6u * (size_t)texel;
                    o = v3(r6[0], r6[1], r6[2]);
                    d = v3(r6[3], r6[4], r6[5]);
                    depth_left = 1u;
                    task_done = false;
                    start_sample = false;
                }
            }
            const uint32_t np = (uint32_t)__popcll(need);
            const uint32_t took = np < avail ? np : avail;
            head += took;
            avail -= took;
        }
        if (!__any(has_task)) {
            if (queue_empty) break;
            continue;                       // nothing acquired (an all-invalid edge tile): pull the next one
        }
        MRT_STAMP(5);

        if (!DBG) {
            // -- (a) lanes starting a sample: jitter and camera point (shader.wgsl:378-381); the lens-disk draws follow below
            V3 cam_p = v3(0.0f, 0.0f, 0.0f);
            bool need_lens = false;
            if (start_sample) need_lens = new_sample_head(ndir, cam_p);
            // -- (b) lanes with a hit: the rest of sphere_hit for the winning sphere (:298-309) and the material's own part
            // (centre, radius) and (material colour, fuzz | ior) of sphere `best` come packed in two 16-byte records built
            // at upload from the reference's SoA arrays (sphere_load_* :254-268, albedo / fuzz loads :205, :232-240): the
            // same bits, one round trip instead of a chain of four.
            V3 normal = v3(0.0f, 0.0f, 0.0f), refl = v3(0.0f, 0.0f, 0.0f);
            float fuzz = 0.0f;
            bool is_lambertian = false, is_metal = false;
            if (hit) {
                const KArgPtr C = cold_args();
                const float4 sh0 = reinterpret_cast<const float4*>(C->shade)[2 * best];
                const float4 sh1 = reinterpret_cast<const float4*>(C->shade)[2 * best + 1];
                const int32_t m_ty = C->i32_data[C->world.spheres.material_ty_base_idx + best];
                const V3 center = v3(sh0.x, sh0.y, sh0.z);
                const float radius = sh0.w;
                const V3 at = o + t_sup * d;                            // ray_normalized_at :103-105
                // (at - center) / radius, :299: one refined reciprocal for the three quotients where no lane's operands
                // call for the scaling steps of `/` (see div_unscaled; a component that is 0 or below 2^-90, a radius
                // below 2^-30: the literal expression for the wave)
                const V3 rel = at - center;
                const float rel_min = __builtin_fminf(__builtin_fminf(__builtin_fabsf(rel.x), __builtin_fabsf(rel.y)), __builtin_fabsf(rel.z));
                if (__builtin_expect(__any(!normal_unscaled_ok(rel_min, radius)), 0)) {
                    normal = rel / radius;
                } else {
                    const Divisor by_r = divisor_of(radius);
                    normal = v3(div_unscaled(rel.x, by_r), div_unscaled(rel.y, by_r), div_unscaled(rel.z, by_r));
                }
                const bool front_face = dot3(normal, d) <= 0.0f;
                if (!front_face) normal = -normal;
                // dyn_material_scatter, shader.wgsl:244-252
                is_lambertian = m_ty == MRT_LAMBERTIAN;
                is_metal = m_ty == MRT_METAL;
                if (is_lambertian || is_metal) {
                    att = att * v3(sh1.x, sh1.y, sh1.z);                    // :353 (moot if the scatter fails: the path ends)
                    if (is_metal) { refl = reflect3(d, normal); fuzz = sh1.w; }     // :230, :232
                } else if (m_ty == MRT_DIELECTRIC) {                        // extension, DESIGN.md §3
                    // The quantities that depend on the sphere alone come from its shading record, evaluated by
                    // the host with the same f32 operations (api.cpp): 1/ior and ((1-ri)/(1+ri))^2 for ri = 1/ior
                    // (front face) and ri = ior (back face).  The attenuation of a Dielectric is (1,1,1).
                    const float ior = sh1.w;
                    const float ri = front_face ? sh1.x : ior;
                    const float r0 = front_face ? sh1.y : sh1.z;
                    float cos_t = dot3(-d, normal);
                    cos_t = (cos_t < 1.0f) ? cos_t : 1.0f;
                    // (1 - cos^2 for cos in [0, 1] is 0 or at least 2^-24: never in sqrtf()'s rescaled range)
                    const float sin_t = sqrt_unscaled(1.0f - cos_t * cos_t);
                    const bool cannot_refract = (ri * sin_t) > 1.0f;
                    const float x1 = 1.0f - cos_t;
                    const float x2 = x1 * x1; const float x4 = x2 * x2; const float x5 = x4 * x1;
                    const float reflectance = r0 + (1.0f - r0) * x5;
                    const float u = rng_f32(rng);                       // always exactly one draw
                    if (cannot_refract || reflectance > u) {
                        ndir = reflect3(d, normal);
                    } else {
                        const V3 perp = v3(ri * (d.x + cos_t * normal.x), ri * (d.y + cos_t * normal.y),
                                           ri * (d.z + cos_t * normal.z));
                        const float k = -sqrt_unscaled(__builtin_fabsf(1.0f - dot3(perp, perp)));      // (0 or >= 2^-25, likewise)
                        ndir = v3(perp.x + k * normal.x, perp.y + k * normal.y, perp.z + k * normal.z);
                    }
                } else {
                    depth_left = 1u;                                        // :249-251: no scatter -> the path ends below
                    ndir = d;
                }
                o = at;
            }
            // -- (c) ONE rejection loop for both kinds of lanes (each lane draws only its own numbers, in its own order):
            // Lambertian / Metal lanes a point of the unit ball (:84-90: three draws per try), lanes starting a sample
            // under a defocusing camera a point of the unit disk (two draws per try; z = 0 leaves the test's value as it is)
            const bool need_ball = is_lambertian || is_metal;
            float bx = 0.0f, by = 0.0f, bz = 0.0f;
            if (need_ball || need_lens) {
                do {
                    bx = rng_pm1(rng); by = rng_pm1(rng);                                 // :77-82 order x, y, z
                    bz = 0.0f;
                    if (need_ball) bz = rng_pm1(rng);
                } while (__builtin_fmaf(bz, bz, __builtin_fmaf(by, by, bx * bx)) > 1.0f);
            }
            // -- (d) what the draws are for
            if (need_lens) new_sample_lens(ndir, cam_p, bx, by);
            if (hit) {
                bool scattered = true;
                if (is_lambertian) {                                    // :203-216
                    // unit_sphere :92-94 = ball / sqrt(dot(ball, ball)).  A component of the ball point is +0 or at
                    // least 2^-24 in magnitude (rng_pm1: a multiple of 2^-24 near 0), so the length is 0 or in [2^-24, 1.74]:
                    // no operand in the range where `/` or sqrtf() rescale; +0 / len comes out +0 and the all-zero point
                    // NaN, as written
                    const Divisor by_len = divisor_of(sqrt_unscaled(__builtin_fmaf(bz, bz, __builtin_fmaf(by, by, bx * bx))));
                    ndir = normal + v3(div_unscaled(bx, by_len), div_unscaled(by, by_len), div_unscaled(bz, by_len));
                    if (dot3(ndir, ndir) == 0.0f) ndir = normal;
                } else if (is_metal) {                                  // :228-242
                    ndir = v3(refl.x + fuzz * bx, refl.y + fuzz * by, refl.z + fuzz * bz);
                    scattered = !(dot3(ndir, normal) <= 0.0f);
                }
                // A failed scatter (:349-351) or the last allowed bounce (:339's loop ends, :357) ends the path with
                // vec3(0): the lane enters the next iteration with nothing left to trace and is accounted there, with the
                // paths that miss (one iteration later than a miss would be; rare at any depth worth rendering).
                depth_left = scattered ? depth_left - 1u : 0u;
            }
            // normalize() of the scattered direction (:354) and of the new sample's camera ray (:381), one code
            // path for the lanes of either kind
            if (has_task && !task_done) {
                // d = ndir / sqrt(dot(ndir, ndir)); as for the normal: the literal expression if any lane has a component
                // that is 0 or below 2^-90, or a squared length outside [2^-60, 2^60)
                const float dd = dot3(ndir, ndir);
                const float nd_min = __builtin_fminf(__builtin_fminf(__builtin_fabsf(ndir.x), __builtin_fabsf(ndir.y)), __builtin_fabsf(ndir.z));
                if (__builtin_expect(__any(!normalize_unscaled_ok(dd, nd_min)), 0)) {
                    d = normalize3(ndir);
                } else {
                    const Divisor by_len = divisor_of(sqrt_unscaled(dd));
                    d = v3(div_unscaled(ndir.x, by_len), div_unscaled(ndir.y, by_len), div_unscaled(ndir.z, by_len));
                }
            }
        }
        MRT_STAMP(0);
        if (!PILOT) started += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(start_sample));
    }

    // samples, world_hit calls, lane slots and member tests are wave totals kept on the scalar side (a ballot and a count per
    // loop trip); the RNG draws are a per-lane counter in the rejection loop and only exist in the COUNT instantiation
    // (mrt_set_draw_counting), without which rng.draws is dead code
    if (!PILOT) {
        const unsigned long long c0 = started, c1 = bounces;      // already wave totals
        unsigned long long c2 = COUNT ? rng.draws : 0u;
        if (COUNT) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c2 += __shfl_xor(c2, off);
        }
        if (lane == 0 && P.counters) {
            atomicAdd(P.counters + 0, c0);
            atomicAdd(P.counters + 1, c1);
            atomicAdd(P.counters + 2, c2);
            atomicAdd(P.counters + 3, 64ull * trips);
            atomicAdd(P.counters + 4, (unsigned long long)mtests);     // wave-uniform
#ifdef MRT_STAMPS
            for (int k = 0; k < 6; k++) atomicAdd(P.counters + 6 + k, (unsigned long long)phase_[k]);
            atomicAdd(P.counters + 5, (unsigned long long)phase_[6]);
            atomicAdd(P.counters + 12, (unsigned long long)rounds_a_);
            atomicAdd(P.counters + 13, (unsigned long long)rounds_b_);
            atomicAdd(P.counters + 14, (unsigned long long)items_a_);
            atomicAdd(P.counters + 15, (unsigned long long)items_b_);
            if (P.wave_log) {
                unsigned long long* wl = P.wave_log + 4ull * (blockIdx.x * kWavesPerGroup + wave);
                wl[0] = wave_t0_; wl[1] = __builtin_amdgcn_s_memrealtime(); wl[2] = trips; wl[3] = c1;
            }
#endif
        }
    }
}

// After render_kernel: per 8x8 tile, one coalesced pass over the parked colour sums --
// colour / spp blended with the previous framebuffer (shader.wgsl:383-385), whole 128-byte lines --
// and the tile's cost (sum of its pixels' bounce-loop trips) for the next frame's queue order.
template <bool PILOT>
__global__ void __launch_bounds__(64) finalize_kernel(const KParams P) {
    const uint32_t lane = threadIdx.x, tile = blockIdx.x;
    const uint32_t W = P.locals.shape[0], H = P.locals.shape[1];
    const uint32_t tile_x = tile % P.tiles_x, band = tile / P.tiles_x;
    const uint32_t px = tile_x * kTileW + (lane & 7u);
    const uint32_t py = (band * P.shard_world + P.shard_rank) * kBandRows + (lane >> 3);
    const size_t texel = (size_t)(band * kBandRows + (lane >> 3)) * W + px;
    uint32_t cost = 0;
    if (px < W && py < H) {
        PixAcc sa = reinterpret_cast<const PixAcc*>(P.pix_acc)[texel];
        // counter-RNG mode: the pixel's blocks of 64 samples were summed separately (possibly by different lanes);
        // their sums are added in block order -- ((S0 + S1) + S2) ... -- which is how the mode defines the colour
        for (uint32_t b = 1; b < (PILOT ? 1u : P.n_blocks); b++) {
            const PixAcc sb = reinterpret_cast<const PixAcc*>(P.pix_acc)[(size_t)b * P.pix_stride + texel];
            sa.r += sb.r; sa.g += sb.g; sa.b += sb.b; sa.cost += sb.cost;
        }
        cost = sa.cost;
        if (!PILOT) {
            const float n = (float)P.locals.samples_per_frame;
            const V3 mean = v3(sa.r / n, sa.g / n, sa.b / n);                       // :383
            const float w = P.locals.framebuffer_weight;
            const float4 prev = reinterpret_cast<const float4*>(P.prev)[texel];        // framebuffer_load :366-369
            float4 res;
            res.x = mixf(mean.x, prev.x, w);                                        // :385
            res.y = mixf(mean.y, prev.y, w);
            res.z = mixf(mean.z, prev.z, w);
            res.w = mixf(1.0f, prev.w, w);
            reinterpret_cast<float4*>(P.out)[texel] = res;
        }
    } else if (!PILOT && px < W) {
        reinterpret_cast<float4*>(P.out)[texel] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // shard padding rows
    }
    // A pixel's samples form one sequential chain, so the frame's critical path is its longest
    // pixel: tiles are ranked by their HEAVIEST pixel, not by their sum.
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o2 = __shfl_xor(cost, off); cost = cost > o2 ? cost : o2; }
    if (lane == 0 && P.tile_cost) P.tile_cost[tile] = cost;
    // the slot's tile queue is empty again for its next render launch (which follows this pass: finalize_done)
    if (lane == 0 && tile == 0) *P.tile_queue = 0u;
}

// Seed texture (Subject::new, lib.rs:389-415) generated on the device: SplitMix64 (mrt_device.h) used as a
// counter-based generator keyed by the GLOBAL pixel index, two outputs per pixel.
__global__ void __launch_bounds__(256) fill_seeds_kernel(uint32_t* seeds, uint64_t seed, uint32_t W, uint32_t H,
                                                         uint32_t shard_rank, uint32_t shard_world,
                                                         uint32_t local_rows) {
    const uint32_t px = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lrow = blockIdx.y;
    if (px >= W || lrow >= local_rows) return;
    const uint32_t py = ((lrow / kBandRows) * shard_world + shard_rank) * kBandRows + (lrow % kBandRows);
    uint4 s = make_uint4(0, 0, 0, 0);
    if (py < H) {
        const uint64_t p = (uint64_t)py * W + px;
        const uint64_t a = splitmix64_at(seed, 2u * p), b = splitmix64_at(seed, 2u * p + 1u);
        s = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
        if ((s.x | s.y | s.z | s.w) == 0u)                             // lib.rs:393 filter
            s = make_uint4(0x9E3779B9u, 0x7F4A7C15u, 0xBF58476Du, 0x1CE4E5B9u);
    }
    reinterpret_cast<uint4*>(seeds)[(size_t)lrow * W + px] = s;
}

}  // namespace

// SMALL scenes (every node id < 1024): member records live in LDS and work items are u16
static bool scene_is_small(const KParams& p) { return p.n_members <= 1024u; }
static uint32_t group_lds_bytes(const KParams& p, bool small) {
    return (small ? p.n_nodes * (uint32_t)sizeof(SphereRec) + lds_index_bytes(p.n_members) : p.box_lds_count * (uint32_t)sizeof(BoxRec)) +
           kWavesPerGroup * lds_wave_bytes(small, p.levels, p.gen_cap, p.mask_chunks);
}

// host: LDS bytes of one workgroup and how many of them one CU holds, for this scene's layout (launch sizing; pinned by
// tests/test_hierarchy_host.py so that a change of the layout cannot drop residency unnoticed)
void render_lds_layout(const KParams& p, uint32_t out[2]) {
    const bool small = scene_is_small(p);
    const uint32_t lds = group_lds_bytes(p, small);
    uint32_t per_cu = (160u * 1024u) / lds;
    if (!small && per_cu > 4u) per_cu = 4u;         // the large-scene kernels are built for 4 waves per SIMD (render_kernel)
    out[0] = lds;
    out[1] = per_cu;
}
// large scenes: how many leading boxes the workgroup keeps in LDS (mrt_internal.h)
uint32_t large_scene_box_lds_count(uint32_t n_top_padded, uint32_t levels, uint32_t mask_chunks, uint32_t cap) {
    // the most that leaves every wave a work stack of kMinStack entries: top + the level below, the top alone, or nothing
    constexpr uint32_t kMinStack = 480;
    const uint32_t both = levels >= 2u ? 5u * n_top_padded : n_top_padded;
    if (both <= cap && large_scene_stack_cap(mask_chunks, both) >= kMinStack) return both;
    if (n_top_padded <= cap && large_scene_stack_cap(mask_chunks, n_top_padded) >= kMinStack) return n_top_padded;
    return 0u;
}
// how many render waves the chip holds at a time for p's scene layout (what a launch width divides)
uint32_t render_resident_waves(const KParams& p) {
    uint32_t lay[2];
    render_lds_layout(p, lay);
    return p.cus * lay[1] * kWavesPerGroup;
}
// the wave's work-stack capacity that makes a large scene's workgroup fit 4 per CU (160 KB / 4 groups, minus the boxes the group
// shares, / 4 waves); a multiple of 4 entries, so that every wave's LDS starts 16-byte aligned
uint32_t large_scene_stack_cap(uint32_t mask_chunks, uint32_t box_lds_count) {
    const uint32_t group = 160u * 1024u / 4u, shared = box_lds_count * (uint32_t)sizeof(BoxRec);
    const uint32_t fixed = lds_off_queues() + (kStackReserve + 2u * kQueueCap) * 4u + mask_chunks * 128u;
    if (shared >= group || (group - shared) / kWavesPerGroup <= fixed) return 0u;
    return (((group - shared) / kWavesPerGroup - fixed) / 4u) & ~3u;
}

// the persistent render waves (pilot: + its cost-only finalize) on `stream`
int launch_render(const KParams& p, bool pilot, uint32_t n_waves, void* stream, uint32_t* which) {
    if (which) *which = 0xFFFFFFFFu;
    if (p.n_tiles == 0 || n_waves == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // (the queue counter is zero: reset at allocation and by every finalize pass of the slot)
    const bool small = scene_is_small(p);
    // persistent grid: as many workgroups as are resident with this launch's LDS footprint
    // (n_waves comes from the register-limited occupancy)
    uint32_t lay[2];
    render_lds_layout(p, lay);
    const uint32_t lds = lay[0];
    {
        const uint32_t cap = p.cus * lay[1] * kWavesPerGroup;
        if (cap < n_waves) n_waves = cap;
    }
    const uint32_t want = n_waves < p.n_tiles ? n_waves : p.n_tiles;
    dim3 grid((want + kWavesPerGroup - 1) / kWavesPerGroup), block(64 * kWavesPerGroup);
    const bool ctr = p.locals.rng_mode == MRT_RNG_COUNTER;
    const bool mfma = p.use_mfma != 0, quad = p.box_quad != 0;
#define MRT_LAUNCH(C_, P_, R_)                                                                                      \
    do {                                                                                                            \
        if (small && mfma) hipLaunchKernelGGL((render_kernel<C_, P_, R_, 0, true>), grid, block, lds, st, p);       \
        else if (small) hipLaunchKernelGGL((render_kernel<C_, P_, R_, 0, false>), grid, block, lds, st, p);         \
        else if (quad && mfma) hipLaunchKernelGGL((render_kernel<C_, P_, R_, 2, true>), grid, block, lds, st, p);   \
        else if (quad) hipLaunchKernelGGL((render_kernel<C_, P_, R_, 2, false>), grid, block, lds, st, p);          \
        else if (mfma) hipLaunchKernelGGL((render_kernel<C_, P_, R_, 1, true>), grid, block, lds, st, p);           \
        else hipLaunchKernelGGL((render_kernel<C_, P_, R_, 1, false>), grid, block, lds, st, p);                    \
    } while (0)
    // which instantiation this is, for mrt_debug_last_launch: bit 0 COUNT, 1 PILOT, 2 CTR, 3 SMALL, 4 MFMA, 5 quadratic box slack
    if (which) *which = ((!pilot && p.count_draws) ? 1u : 0u) | (pilot ? 2u : 0u) | (ctr ? 4u : 0u) | (small ? 8u : 0u) | (mfma ? 16u : 0u) |
                        ((!small && quad) ? 32u : 0u);
    if (pilot) {
        if (ctr) MRT_LAUNCH(false, true, true); else MRT_LAUNCH(false, true, false);
        hipLaunchKernelGGL((finalize_kernel<true>), dim3(p.n_tiles), dim3(64), 0, st, p);
    } else if (ctr) {
        if (p.count_draws) MRT_LAUNCH(true, false, true); else MRT_LAUNCH(false, false, true);
    } else if (p.count_draws) {
        MRT_LAUNCH(true, false, false);
    } else {
        MRT_LAUNCH(false, false, false);
    }
#undef MRT_LAUNCH
    return (int)hipGetLastError();
}

// mrt_debug_world_hit: one world_hit per ray of p.dbg_rays (an 8-wide virtual image, ray = texel index)
int launch_debug_world_hit(const KParams& p, uint32_t n_waves, void* stream) {
    if (p.n_tiles == 0 || n_waves == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(p.tile_queue, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    const bool small = scene_is_small(p);
    uint32_t lay[2];
    render_lds_layout(p, lay);
    const uint32_t lds = lay[0];
    const uint32_t cap = p.cus * lay[1] * kWavesPerGroup;
    if (cap < n_waves) n_waves = cap;
    const uint32_t want = n_waves < p.n_tiles ? n_waves : p.n_tiles;
    dim3 grid((want + kWavesPerGroup - 1) / kWavesPerGroup), block(64 * kWavesPerGroup);
    const bool mfma = p.use_mfma != 0;
    const bool quad = p.box_quad != 0;
    if (small && mfma) hipLaunchKernelGGL((render_kernel<false, false, false, 0, true, true>), grid, block, lds, st, p);
    else if (small) hipLaunchKernelGGL((render_kernel<false, false, false, 0, false, true>), grid, block, lds, st, p);
    else if (quad && mfma) hipLaunchKernelGGL((render_kernel<false, false, false, 2, true, true>), grid, block, lds, st, p);
    else if (quad) hipLaunchKernelGGL((render_kernel<false, false, false, 2, false, true>), grid, block, lds, st, p);
    else if (mfma) hipLaunchKernelGGL((render_kernel<false, false, false, 1, true, true>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((render_kernel<false, false, false, 1, false, true>), grid, block, lds, st, p);
    return (int)hipGetLastError();
}

// colour sums -> framebuffer (+ per-tile costs), one wave per 8x8 tile
int launch_finalize(const KParams& p, void* stream) {
    if (p.n_tiles == 0) return 0;
    hipLaunchKernelGGL((finalize_kernel<false>), dim3(p.n_tiles), dim3(64), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

// how many waves of the render kernel one CU holds (occupancy API)
int render_waves_per_cu(int* out) {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_kernel<true, false, false, 1, false>, 64 * kWavesPerGroup, 0);
    *out = nb * (int)kWavesPerGroup;
    return (int)e;
}

int launch_fill_seeds(uint32_t* seeds, uint64_t seed, uint32_t width, uint32_t height,
                      uint32_t shard_rank, uint32_t shard_world, uint32_t local_bands, void* stream) {
    if (width == 0 || local_bands == 0) return 0;
    const uint32_t local_rows = local_bands * kBandRows;
    dim3 grid((width + 255) / 256, local_rows), block(256);
    hipLaunchKernelGGL(fill_seeds_kernel, grid, block, 0, (hipStream_t)stream, seeds, seed, width, height,
                       shard_rank, shard_world, local_rows);
    return (int)hipGetLastError();
}

}  // namespace mrt
