// Scene builders (host only).  mrt_scene_default is the reference's hard-coded scene
// (raytracer/src/lib.rs:687-720); the other two are the synthetic inputs SURVEY.md §8(d)
// prescribes for configs C2-C5 and have no reference counterpart.  They produce the
// api::World-style AoS (lib.rs:611-639) that mrt_set_world packs.

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mrt_internal.h"

namespace {

// scene-generation RNG: sequential SplitMix64, 24-bit floats in [0,1)
struct SceneRng {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    float unit() { return (float)(next() >> 40) * 0x1p-24f; }
    float range(float lo, float hi) { return lo + (hi - lo) * unit(); }
};

mrt_sphere make(float x, float y, float z, float r, int32_t ty, float ar, float ag, float ab, float param) {
    mrt_sphere s;
    s.center[0] = x; s.center[1] = y; s.center[2] = z; s.radius = r; s.material_ty = ty;
    s.albedo[0] = ar; s.albedo[1] = ag; s.albedo[2] = ab; s.param = param;
    return s;
}

struct Sink {
    mrt_sphere* out; size_t cap; size_t n = 0;
    void push(const mrt_sphere& s) { if (out && n < cap) out[n] = s; n++; }
};

void lookat(mrt_camera* c, float fx, float fy, float fz, float ax, float ay, float az, float vfov, float defocus, float focus) {
    if (!c) return;
    std::memset(c, 0, sizeof *c);
    c->mode = 1;
    c->lookfrom[0] = fx; c->lookfrom[1] = fy; c->lookfrom[2] = fz;
    c->lookat[0] = ax; c->lookat[1] = ay; c->lookat[2] = az;
    c->vup[0] = 0.0f; c->vup[1] = 1.0f; c->vup[2] = 0.0f;
    c->vfov_deg = vfov; c->defocus_angle_deg = defocus; c->focus_dist = focus;
}

}  // namespace

extern "C" {

// lib.rs:687-720
int mrt_scene_default(mrt_sphere* out, size_t cap) {
    Sink k{out, cap};
    k.push(make(0.0f, -100.5f, -1.0f, 100.0f, MRT_LAMBERTIAN, 0.8f, 0.8f, 0.0f, 0.0f));
    k.push(make(0.0f, 0.0f, -1.0f, 0.5f, MRT_LAMBERTIAN, 0.7f, 0.3f, 0.3f, 0.0f));
    k.push(make(-1.0f, 0.0f, -1.0f, 0.5f, MRT_METAL, 0.8f, 0.8f, 0.8f, 0.3f));
    k.push(make(1.0f, 0.0f, -1.0f, 0.5f, MRT_METAL, 0.8f, 0.6f, 0.2f, 1.0f));
    return (int)k.n;
}

// "Ray Tracing in One Weekend" cover scene in the reference's conventions.
int mrt_scene_cover(uint64_t scene_seed, int dielectric, mrt_sphere* out, size_t cap, mrt_camera* cam_out) {
    Sink k{out, cap};
    SceneRng rng{scene_seed};
    k.push(make(0.0f, -1000.0f, 0.0f, 1000.0f, MRT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0f));
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            const float choose = rng.unit();
            const float cx = (float)a + 0.9f * rng.unit();
            const float cz = (float)b + 0.9f * rng.unit();
            const float dx = cx - 4.0f, dz = cz;
            if (std::sqrt(dx * dx + dz * dz) <= 0.9f) continue;
            if (choose < 0.8f) {
                const float r0 = rng.unit() * rng.unit(), g0 = rng.unit() * rng.unit(), b0 = rng.unit() * rng.unit();
                k.push(make(cx, 0.2f, cz, 0.2f, MRT_LAMBERTIAN, r0, g0, b0, 0.0f));
            } else if (choose < 0.95f) {
                const float r0 = rng.range(0.5f, 1.0f), g0 = rng.range(0.5f, 1.0f), b0 = rng.range(0.5f, 1.0f);
                const float fuzz = rng.range(0.0f, 0.5f);
                k.push(make(cx, 0.2f, cz, 0.2f, MRT_METAL, r0, g0, b0, fuzz));
            } else if (dielectric) {
                k.push(make(cx, 0.2f, cz, 0.2f, MRT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f));
            } else {
                k.push(make(cx, 0.2f, cz, 0.2f, MRT_METAL, 0.9f, 0.9f, 0.9f, 0.0f));
            }
        }
    }
    if (dielectric) k.push(make(0.0f, 1.0f, 0.0f, 1.0f, MRT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f));
    else k.push(make(0.0f, 1.0f, 0.0f, 1.0f, MRT_METAL, 0.9f, 0.9f, 0.9f, 0.0f));
    k.push(make(-4.0f, 1.0f, 0.0f, 1.0f, MRT_LAMBERTIAN, 0.4f, 0.2f, 0.1f, 0.0f));
    k.push(make(4.0f, 1.0f, 0.0f, 1.0f, MRT_METAL, 0.7f, 0.6f, 0.5f, 0.0f));
    lookat(cam_out, 13.0f, 2.0f, 3.0f, 0.0f, 0.0f, 0.0f, 20.0f, dielectric ? 0.6f : 0.0f, 10.0f);
    return (int)k.n;
}

// Stress scene: ground + n_side^2 small spheres on a jittered unit grid, 80/15/5 % L/M/D.
int mrt_scene_stress(uint64_t scene_seed, uint32_t n_side, mrt_sphere* out, size_t cap, mrt_camera* cam_out) {
    if (n_side == 0 || n_side > 1000) return -MRT_ERR_INVALID_ARG;    // 1000^2 + 1 <= kMaxSpheres
    Sink k{out, cap};
    SceneRng rng{scene_seed};
    k.push(make(0.0f, -1000.0f, 0.0f, 1000.0f, MRT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0f));
    const float half = 0.5f * (float)n_side;
    for (uint32_t a = 0; a < n_side; a++) {
        for (uint32_t b = 0; b < n_side; b++) {
            const float r = rng.range(0.1f, 0.2f);
            const float cx = (float)a - half + 0.6f * rng.unit();
            const float cz = (float)b - half + 0.6f * rng.unit();
            const float choose = rng.unit();
            const float r0 = rng.unit(), g0 = rng.unit(), b0 = rng.unit(), p0 = rng.unit();
            if (choose < 0.8f) k.push(make(cx, r, cz, r, MRT_LAMBERTIAN, r0 * r0, g0 * g0, b0 * b0, 0.0f));
            else if (choose < 0.95f) k.push(make(cx, r, cz, r, MRT_METAL, 0.5f + 0.5f * r0, 0.5f + 0.5f * g0, 0.5f + 0.5f * b0, 0.5f * p0));
            else k.push(make(cx, r, cz, r, MRT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f));
        }
    }
    const float s = (float)n_side;
    lookat(cam_out, 0.0f, 0.4f * s, 0.9f * s, 0.0f, 0.0f, 0.0f, 40.0f, 0.0f, 10.0f);
    return (int)k.n;
}


// ---- scenes as data ------------------------------------------------------------------------------------

int mrt_scene_save(const char* path, const mrt_sphere* spheres, size_t n, const mrt_camera* cam) {
    if (!path || (!spheres && n)) return MRT_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, "w");
    if (!f) { mrt::set_global_error("mrt_scene_save: cannot open the file for writing"); return MRT_ERR_IO; }
    std::fprintf(f, "# myraytracer_amd scene, %zu spheres; sphere order = the reference's sphere index order\n", n);
    if (cam && cam->mode == 1)
        std::fprintf(f, "camera lookat %.9g %.9g %.9g  %.9g %.9g %.9g  %.9g %.9g %.9g  %.9g %.9g %.9g\n",
                     cam->lookfrom[0], cam->lookfrom[1], cam->lookfrom[2], cam->lookat[0], cam->lookat[1], cam->lookat[2],
                     cam->vup[0], cam->vup[1], cam->vup[2], cam->vfov_deg, cam->defocus_angle_deg, cam->focus_dist);
    else if (cam)
        std::fprintf(f, "camera pinhole\n");
    for (size_t i = 0; i < n; i++) {
        const mrt_sphere& s = spheres[i];
        std::fprintf(f, "sphere %.9g %.9g %.9g %.9g ", s.center[0], s.center[1], s.center[2], s.radius);
        // the canonical forms drop the fields the material does not read; anything else is kept verbatim
        if (s.material_ty == MRT_LAMBERTIAN && s.param == 0.0f && !std::signbit(s.param))
            std::fprintf(f, "lambertian %.9g %.9g %.9g\n", s.albedo[0], s.albedo[1], s.albedo[2]);
        else if (s.material_ty == MRT_METAL)
            std::fprintf(f, "metal %.9g %.9g %.9g %.9g\n", s.albedo[0], s.albedo[1], s.albedo[2], s.param);
        else if (s.material_ty == MRT_DIELECTRIC && s.albedo[0] == 1.0f && s.albedo[1] == 1.0f && s.albedo[2] == 1.0f)
            std::fprintf(f, "dielectric %.9g\n", s.param);
        else
            std::fprintf(f, "material %d %.9g %.9g %.9g %.9g\n", s.material_ty, s.albedo[0], s.albedo[1], s.albedo[2], s.param);
    }
    const bool bad = std::ferror(f) != 0;
    if (std::fclose(f) != 0 || bad) { mrt::set_global_error("mrt_scene_save: write failed"); return MRT_ERR_IO; }
    return MRT_OK;
}

int mrt_scene_load(const char* path, mrt_sphere* out, size_t cap, mrt_camera* cam_out, int* has_camera) {
    if (!path || (!out && cap)) return -MRT_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, "r");
    if (!f) { mrt::set_global_error("mrt_scene_load: cannot open the file"); return -MRT_ERR_IO; }
    if (cam_out) std::memset(cam_out, 0, sizeof *cam_out);
    if (has_camera) *has_camera = 0;
    Sink k{out, cap};
    std::vector<char> line(4096);
    int lineno = 0, status = 0;
    auto fail = [&](const char* what) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "mrt_scene_load: line %d: %s", lineno, what);
        mrt::set_global_error(buf);
        status = -MRT_ERR_BAD_SCENE;
    };
    while (!status && std::fgets(line.data(), (int)line.size(), f)) {
        lineno++;
        if (!std::strchr(line.data(), '\n') && !std::feof(f)) { fail("line too long"); break; }
        // tokens
        std::vector<char*> tok;
        for (char* p = std::strtok(line.data(), " \t\r\n"); p; p = std::strtok(nullptr, " \t\r\n")) {
            if (*p == '#') break;
            tok.push_back(p);
        }
        if (tok.empty()) continue;
        auto num = [&](size_t i, float* v) -> bool {
            if (i >= tok.size()) return false;
            char* end = nullptr;
            *v = std::strtof(tok[i], &end);
            return end != tok[i] && *end == 0;
        };
        auto nums = [&](size_t first, size_t cnt, float* v) -> bool {
            for (size_t q = 0; q < cnt; q++) if (!num(first + q, v + q)) return false;
            return true;
        };
        const std::string kw = tok[0];
        if (kw == "camera") {
            if (tok.size() == 2 && std::string(tok[1]) == "pinhole") {
                if (cam_out) std::memset(cam_out, 0, sizeof *cam_out);
            } else if (tok.size() == 14 && std::string(tok[1]) == "lookat") {
                float v[12];
                if (!nums(2, 12, v)) { fail("camera lookat needs 12 numbers"); break; }
                if (cam_out) {
                    cam_out->mode = 1;
                    std::memcpy(cam_out->lookfrom, v, 12); std::memcpy(cam_out->lookat, v + 3, 12); std::memcpy(cam_out->vup, v + 6, 12);
                    cam_out->vfov_deg = v[9]; cam_out->defocus_angle_deg = v[10]; cam_out->focus_dist = v[11];
                }
            } else { fail("expected `camera pinhole` or `camera lookat` + 12 numbers"); break; }
            if (has_camera) *has_camera = 1;
        } else if (kw == "sphere") {
            float g[4];
            if (tok.size() < 6 || !nums(1, 4, g)) { fail("sphere needs centre xyz, radius and a material"); break; }
            const std::string mat = tok[5];
            float a[4] = {1.0f, 1.0f, 1.0f, 0.0f};
            int32_t ty = 0;
            if (mat == "lambertian" && tok.size() == 9 && nums(6, 3, a)) ty = MRT_LAMBERTIAN;
            else if (mat == "metal" && tok.size() == 10 && nums(6, 4, a)) ty = MRT_METAL;
            else if (mat == "dielectric" && tok.size() == 7 && num(6, a + 3)) ty = MRT_DIELECTRIC;
            else if (mat == "material" && tok.size() == 11 && nums(7, 4, a)) {
                char* end = nullptr;
                const long t = std::strtol(tok[6], &end, 10);
                if (end == tok[6] || *end != 0 || t < INT32_MIN || t > INT32_MAX) { fail("material type must be an integer"); break; }
                ty = (int32_t)t;
            } else { fail("unknown material or wrong number of fields"); break; }
            k.push(make(g[0], g[1], g[2], g[3], ty, a[0], a[1], a[2], a[3]));
        } else {
            fail("unknown keyword (expected `camera` or `sphere`)");
        }
    }
    if (!status && std::ferror(f)) { mrt::set_global_error("mrt_scene_load: read failed"); status = -MRT_ERR_IO; }
    std::fclose(f);
    if (status) return status;
    if (k.n > (size_t)INT32_MAX) return -MRT_ERR_BAD_SCENE;
    return (int)k.n;
}

}  // extern "C"
