// Headless counterpart of native-runner/src/main.rs: the same five flags
// (main.rs:20-31: --width --height --samples-per-frame --ray-depth --max-framebuffer-weight)
// driving the MI355X backend through the C ABI, plus what a windowless run needs
// (--frames, --seed, --scene, --out, --device; --warmup N renders N untimed frames first and restarts the accumulation;
// --rng stream|counter selects the RNG mode, MRT_RNG_*).  The reference renders forever into a
// window (lib.rs:187-192); this renders --frames frames and writes the image.
// --gpus N (devices 0..N-1) or --devices a,b,c tile-shards the frame over several GPUs from this one
// process: one context per GPU, interleaved 8-row bands, one mrt_gather per frame onto the first device
// (SURVEY.md 8e).  A device may be listed more than once (--devices 0,0 rehearses the path on one GPU).

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/myraytracer_amd.h"

static void usage() {
    std::fprintf(stderr,
        "usage: native_runner [--width N] [--height N] [--samples-per-frame N] [--ray-depth N]\n"
        "                     [--max-framebuffer-weight F] [--frames N] [--warmup N] [--seed N] [--rng stream|counter]\n"
        "                     [--scene default|cover|cover-glass|stress | --scene-file FILE] [--save-scene FILE]\n"
        "                     [--out FILE.pfm|FILE.ppm|FILE.png] [--device N | --gpus N | --devices a,b,...]\n"
                         "                     [--schedule div,mult]\n");
}

int main(int argc, char** argv) {
    // The HOST's job, before its first HIP call: up to eight frames of a pixel-starved shard run side by
    // side, each on a stream of its own, and HIP gives a process 4 hardware queues unless told otherwise.  The library itself
    // never touches the environment.  A value the user exported wins.
    (void)setenv("GPU_MAX_HW_QUEUES", "20", 0);
    mrt_args args;
    mrt_args_default(&args);
    uint32_t frames = 1, warmup = 0, rng_mode = MRT_RNG_PIXEL_STREAM; uint64_t seed = 1; int device = 0;
    uint32_t hint_div = 0, hint_mult = 0;
    std::string scene = "default", scene_file, save_scene, out;
    std::vector<int> devices;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i], v;
        size_t eq = a.find('=');
        if (eq != std::string::npos) { v = a.substr(eq + 1); a = a.substr(0, eq); }
        else if (a == "--help" || a == "-h") { usage(); return 0; }
        else if (i + 1 < argc) v = argv[++i];
        else { usage(); return 2; }
        if (a == "--width") args.width = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
        else if (a == "--height") args.height = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
        else if (a == "--samples-per-frame") args.samples_per_frame = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
        else if (a == "--ray-depth") args.ray_depth = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
        else if (a == "--max-framebuffer-weight") args.max_framebuffer_weight = std::strtof(v.c_str(), nullptr);
        else if (a == "--frames") frames = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
        else if (a == "--warmup") warmup = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
        else if (a == "--rng") {
            if (v == "counter") rng_mode = MRT_RNG_COUNTER;
            else if (v != "stream") { std::fprintf(stderr, "unknown --rng %s\n", v.c_str()); return 2; }
        }
        else if (a == "--seed") seed = std::strtoull(v.c_str(), nullptr, 10);
        else if (a == "--scene") scene = v;
        else if (a == "--scene-file") scene_file = v;
        else if (a == "--save-scene") save_scene = v;
        else if (a == "--schedule") {       // "div,mult": pin the launch schedule on every GPU (mrt_set_schedule_hint)
            if (std::sscanf(v.c_str(), "%u,%u", &hint_div, &hint_mult) != 2) { std::fprintf(stderr, "--schedule wants div,mult\n"); return 2; }
        }
        else if (a == "--out") out = v;
        else if (a == "--device") device = std::atoi(v.c_str());
        else if (a == "--gpus") { devices.clear(); for (int d = 0; d < std::atoi(v.c_str()); d++) devices.push_back(d); }
        else if (a == "--devices") {
            devices.clear();
            for (size_t p0 = 0; p0 <= v.size();) {
                const size_t p1 = v.find(',', p0) == std::string::npos ? v.size() : v.find(',', p0);
                if (p1 > p0) devices.push_back(std::atoi(v.substr(p0, p1 - p0).c_str()));
                p0 = p1 + 1;
            }
        }
        else { std::fprintf(stderr, "unknown flag %s\n", a.c_str()); usage(); return 2; }
    }
    mrt_args_resolve_size(&args);

    std::vector<mrt_sphere> spheres(70000);
    mrt_camera cam; std::memset(&cam, 0, sizeof cam);
    int n;
    if (!scene_file.empty()) {
        // scenes as data: count first, then read
        int has_cam = 0;
        n = mrt_scene_load(scene_file.c_str(), nullptr, 0, &cam, &has_cam);
        if (n >= 0) { spheres.resize((size_t)n + 1); n = mrt_scene_load(scene_file.c_str(), spheres.data(), spheres.size(), &cam, &has_cam); }
        if (n < 0) { std::fprintf(stderr, "%s: %s (%s)\n", scene_file.c_str(), mrt_status_string(-n), mrt_last_error(nullptr)); return 1; }
    }
    else if (scene == "default") n = mrt_scene_default(spheres.data(), spheres.size());
    else if (scene == "cover") n = mrt_scene_cover(1, 0, spheres.data(), spheres.size(), &cam);
    else if (scene == "cover-glass") n = mrt_scene_cover(1, 1, spheres.data(), spheres.size(), &cam);
    else if (scene == "stress") n = mrt_scene_stress(1, 100, spheres.data(), spheres.size(), &cam);
    else { std::fprintf(stderr, "unknown scene %s\n", scene.c_str()); return 2; }
    if (n < 0) { std::fprintf(stderr, "scene generation failed\n"); return 1; }
    if (!save_scene.empty()) {
        const int ss = mrt_scene_save(save_scene.c_str(), spheres.data(), (size_t)n, &cam);
        if (ss != MRT_OK) { std::fprintf(stderr, "%s: %s (%s)\n", save_scene.c_str(), mrt_status_string(ss), mrt_last_error(nullptr)); return 1; }
    }

    // one context per GPU; with a single device this is exactly the reference's one State
    if (devices.empty()) devices.push_back(device);
    const uint32_t n_gpus = (uint32_t)devices.size();
    std::vector<mrt_ctx*> ctxs(n_gpus, nullptr);
    auto destroy_all = [&]() { for (mrt_ctx* c : ctxs) mrt_destroy(c); };
#define TRY(ctx, call) do { int s_ = (call); if (s_ != MRT_OK) { std::fprintf(stderr, "%s: %s (%s)\n", #call, mrt_status_string(s_), mrt_last_error(ctx)); destroy_all(); return 1; } } while (0)
    for (uint32_t i = 0; i < n_gpus; i++) {
        int st = mrt_create(&args, seed, devices[i], &ctxs[i]);
        if (st != MRT_OK) { std::fprintf(stderr, "mrt_create(device %d): %s (%s)\n", devices[i], mrt_status_string(st), mrt_last_error(nullptr)); destroy_all(); return 1; }
        if (n_gpus > 1) TRY(ctxs[i], mrt_set_shard(ctxs[i], i, n_gpus));
        TRY(ctxs[i], mrt_set_world(ctxs[i], spheres.data(), (size_t)n));
        TRY(ctxs[i], mrt_set_camera(ctxs[i], &cam));
        if (rng_mode != MRT_RNG_PIXEL_STREAM) TRY(ctxs[i], mrt_set_rng_mode(ctxs[i], rng_mode));
        if (hint_div) TRY(ctxs[i], mrt_set_schedule_hint(ctxs[i], hint_div, hint_mult));
    }
    if (warmup) {           // untimed: the tile-cost estimate, buffers and peer mappings exist afterwards
        for (mrt_ctx* c : ctxs) TRY(c, mrt_render(c, warmup));
        if (n_gpus > 1) TRY(ctxs[0], mrt_gather(ctxs.data(), n_gpus, 0));
        for (mrt_ctx* c : ctxs) TRY(c, mrt_sync(c));
        for (mrt_ctx* c : ctxs) TRY(c, mrt_reset(c));
        // every GPU renders an equal share of the same frame: one launch schedule for all of them -- the first context's, if its
        // controller has settled (the setting is measured on the host's clock: GPUs may otherwise decide differently)
        uint32_t sch[6] = {0, 0, 0, 0, 0, 0};
        TRY(ctxs[0], mrt_get_schedule(ctxs[0], sch));
        if (n_gpus > 1 && !hint_div && sch[0] != 0 && sch[2] != 0)
            for (mrt_ctx* c : ctxs) TRY(c, mrt_set_schedule_hint(c, sch[0], sch[1]));
    }
    for (mrt_ctx* c : ctxs) TRY(c, mrt_sync(c));
    auto t0 = std::chrono::steady_clock::now();
    // only the final accumulated image is wanted, so every GPU renders all its frames (mrt_render may share a launch
    // among several frames when its shard is too small to fill the GPU) and the shards are gathered once
    for (mrt_ctx* c : ctxs) TRY(c, mrt_render(c, frames));                   // asynchronous: all GPUs render at once
    if (n_gpus > 1) TRY(ctxs[0], mrt_gather(ctxs.data(), n_gpus, 0));         // every shard's bands -> the first GPU
    for (mrt_ctx* c : ctxs) TRY(c, mrt_sync(c));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double samples = (double)args.width * args.height * args.samples_per_frame * frames;
    std::printf("%ux%u, %u spp x %u frames, depth %u, %d spheres, %u GPU(s): %.3f s, %.1f Msamples/s\n", args.width, args.height,
                args.samples_per_frame, frames, args.ray_depth, n, n_gpus, sec, samples / sec * 1e-6);
    if (!out.empty()) {
        std::vector<float> fb((size_t)args.width * args.height * 4);
        if (n_gpus > 1) TRY(ctxs[0], mrt_read_gathered(ctxs[0], fb.data(), fb.size()));
        else TRY(ctxs[0], mrt_read_framebuffer(ctxs[0], fb.data(), fb.size()));
        const std::string ext = out.size() > 4 ? out.substr(out.size() - 4) : "";
        int s2 = ext == ".ppm" ? mrt_write_ppm(out.c_str(), fb.data(), args.width, args.height)
               : ext == ".png" ? mrt_write_png(out.c_str(), fb.data(), args.width, args.height)
                               : mrt_write_pfm(out.c_str(), fb.data(), args.width, args.height);
        if (s2 != MRT_OK) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); destroy_all(); return 1; }
        std::printf("wrote %s\n", out.c_str());
    }
    destroy_all();
    return 0;
}
