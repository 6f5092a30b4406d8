/*
 * A plain-C caller of the ABI, written the way the reference's Rust side would bind it:
 * it builds the reference's raw::World -- the 64-byte struct of raytracer/src/lib.rs:676-684, NOT the
 * 80-byte mrt_world -- and the three SoA arrays exactly as lib.rs:722-799 lays them out for the shipped
 * 4-sphere scene (lib.rs:687-720), and drives create -> set_world_raw -> redraw -> read_framebuffer.
 *
 * The 64-byte struct is placed so that it ENDS at the last byte of a page and the next page is PROT_NONE:
 * a library that read sizeof(mrt_world) = 80 bytes from it would fault here.
 *
 * Compiled by tests/test_abi.py / tests/test_gpu_golden_and_api.py with gcc and linked against
 * myraytracer_amd/lib/libmyraytracer_amd.so.
 *     abi_c_caller host                  host-only checks (no GPU needed), exit 0
 *     abi_c_caller render W H SPP DEPTH SEED OUT.bin    renders one frame, writes W*H*4 floats
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include "../include/myraytracer_amd.h"

/* raw::World, lib.rs:651-684: SphereRange (32 B) + LambertianRange (16 B) + MetalRange (16 B) */
typedef struct {
    int32_t center_base_idx, radius_base_idx, material_ty_base_idx, material_idx_base_idx, length, pad0[3];
    int32_t l_albedo_base_idx, l_length, pad1[2];
    int32_t m_albedo_base_idx, m_fuzz_base_idx, m_length, pad2;
} raw_world;

static raw_world* world_at_page_end(void) {
    const long page = sysconf(_SC_PAGESIZE);
    char* p = mmap(NULL, (size_t)(2 * page), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return NULL;
    if (mprotect(p + page, (size_t)page, PROT_NONE) != 0) return NULL;
    return (raw_world*)(p + page - (long)sizeof(raw_world));
}

/* the shipped scene as lib.rs:687-799 packs it */
static void shipped_scene(raw_world* w, float vec4[8][4], float f32[6], int32_t i32[8]) {
    const float centers[4][3] = {{0.0f, -100.5f, -1.0f}, {0.0f, 0.0f, -1.0f}, {-1.0f, 0.0f, -1.0f}, {1.0f, 0.0f, -1.0f}};
    const float radii[4] = {100.0f, 0.5f, 0.5f, 0.5f};
    const float lamb[2][3] = {{0.8f, 0.8f, 0.0f}, {0.7f, 0.3f, 0.3f}};
    const float metal[2][3] = {{0.8f, 0.8f, 0.8f}, {0.8f, 0.6f, 0.2f}};
    const float fuzz[2] = {0.3f, 1.0f};
    memset(w, 0, sizeof *w);
    for (int i = 0; i < 4; i++) { memcpy(vec4[i], centers[i], 12); vec4[i][3] = 1.0f; f32[i] = radii[i]; }
    for (int i = 0; i < 2; i++) { memcpy(vec4[4 + i], lamb[i], 12); vec4[4 + i][3] = 1.0f; }
    for (int i = 0; i < 2; i++) { memcpy(vec4[6 + i], metal[i], 12); vec4[6 + i][3] = 1.0f; }
    w->center_base_idx = 0; w->radius_base_idx = 0; w->material_ty_base_idx = 0; w->material_idx_base_idx = 4; w->length = 4;
    w->l_albedo_base_idx = 4; w->l_length = 2;
    w->m_albedo_base_idx = 6; w->m_fuzz_base_idx = 4; w->m_length = 2;
    f32[4] = fuzz[0]; f32[5] = fuzz[1];
    const int32_t ty[4] = {MRT_LAMBERTIAN, MRT_LAMBERTIAN, MRT_METAL, MRT_METAL}, idx[4] = {0, 1, 0, 1};
    memcpy(i32, ty, sizeof ty);
    memcpy(i32 + 4, idx, sizeof idx);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: abi_c_caller host | render W H SPP DEPTH SEED OUT.bin\n"); return 2; }
    if (sizeof(raw_world) != MRT_WORLD_BYTES_REFERENCE) { fprintf(stderr, "raw_world is %zu bytes\n", sizeof(raw_world)); return 1; }
    raw_world* w = world_at_page_end();
    if (!w) { perror("mmap"); return 1; }
    float vec4[8][4]; float f32[6]; int32_t i32[8];
    memset(vec4, 0, sizeof vec4);
    shipped_scene(w, vec4, f32, i32);

    if (strcmp(argv[1], "host") == 0) {
        if (mrt_abi_version() != MRT_ABI_VERSION) { fprintf(stderr, "abi version %d != header %d\n", mrt_abi_version(), MRT_ABI_VERSION); return 1; }
        /* the library's own packing of its own shipped scene must give the same blocks (lib.rs:722-799) */
        mrt_sphere sp[4];
        if (mrt_scene_default(sp, 4) != 4) return 1;
        mrt_world mw; float v2[16][4]; float f2[16]; int32_t i2[16]; size_t nv, nf, ni;
        if (mrt_pack_world(sp, 4, &mw, &v2[0][0], 16, &nv, f2, 16, &nf, i2, 16, &ni) != MRT_OK) return 1;
        if (nv != 8 || nf != 6 || ni != 8) { fprintf(stderr, "packed sizes %zu %zu %zu\n", nv, nf, ni); return 1; }
        if (memcmp(&mw, w, sizeof *w) != 0) { fprintf(stderr, "raw::World block differs from mrt_pack_world's\n"); return 1; }
        if (memcmp(v2, vec4, 8 * 16) != 0 || memcmp(f2, f32, 6 * 4) != 0 || memcmp(i2, i32, 8 * 4) != 0) { fprintf(stderr, "arrays differ\n"); return 1; }
        /* without a GPU the product must refuse, not fall back */
        mrt_args a; mrt_args_default(&a); a.width = 16; a.height = 8;
        mrt_ctx* ctx = NULL;
        const int st = mrt_create(&a, 1, 0, &ctx);
        printf("host ok; mrt_create -> %s\n", mrt_status_string(st));
        if (st == MRT_OK) mrt_destroy(ctx);
        return 0;
    }
    if (strcmp(argv[1], "render") != 0 || argc != 8) return 2;
    mrt_args a; mrt_args_default(&a);
    a.width = (uint32_t)atoi(argv[2]); a.height = (uint32_t)atoi(argv[3]);
    a.samples_per_frame = (uint32_t)atoi(argv[4]); a.ray_depth = (uint32_t)atoi(argv[5]);
    const uint64_t seed = strtoull(argv[6], NULL, 10);
    mrt_ctx* ctx = NULL;
    int st = mrt_create(&a, seed, 0, &ctx);
    if (st != MRT_OK) { fprintf(stderr, "mrt_create: %s (%s)\n", mrt_status_string(st), mrt_last_error(NULL)); return 1; }
#define TRY(call) do { st = (call); if (st != MRT_OK) { fprintf(stderr, "%s: %s (%s)\n", #call, mrt_status_string(st), mrt_last_error(ctx)); mrt_destroy(ctx); return 1; } } while (0)
    /* a wrong size must be refused, the reference's 64 bytes accepted */
    if (mrt_set_world_raw(ctx, w, 48, &vec4[0][0], 8, f32, 6, i32, 8) != MRT_ERR_INVALID_ARG) { fprintf(stderr, "world_bytes 48 accepted\n"); return 1; }
    TRY(mrt_set_world_raw(ctx, w, sizeof *w, &vec4[0][0], 8, f32, 6, i32, 8));
    TRY(mrt_redraw(ctx));
    TRY(mrt_sync(ctx));
    const size_t n = (size_t)a.width * a.height * 4;
    float* fb = malloc(n * sizeof(float));
    TRY(mrt_read_framebuffer(ctx, fb, n));
    FILE* f = fopen(argv[7], "wb");
    if (!f || fwrite(fb, sizeof(float), n, f) != n || fclose(f) != 0) { fprintf(stderr, "cannot write %s\n", argv[7]); return 1; }
    mrt_destroy(ctx);
    printf("rendered %ux%u, %u spp\n", a.width, a.height, a.samples_per_frame);
    return 0;
}
