// Image writers (host only).  The reference never leaves the GPU: it presents the
// accumulated RGBA32F texture to a window surface (raytracer/src/lib.rs:270-297,
// sample_framebuffer.wgsl:38-41, with a Y flip at :24).  These writers are the headless
// stand-in for that present pass.  The surface the reference draws to is the adapter's default
// format (surface.get_default_config, lib.rs:349-351; the fragment target takes that format,
// lib.rs:1133), an sRGB format on every wgpu backend: the hardware applies the sRGB OETF to the
// linear value fs_main returns (sample_framebuffer.wgsl:38-41) and rounds to 8 bits.  mrt_write_ppm
// does the same conversion on the host.

#include <cmath>
#include <cstdio>
#include <vector>

#include "mrt_internal.h"

// linear [0,1] -> 8-bit sRGB as a UNORM sRGB render target stores it: clamp, the piecewise OETF of
// IEC 61966-2-1, round to nearest
static unsigned char srgb8(float v) {
    if (!(v > 0.0f)) return 0;                       // negatives and NaN clamp to 0
    if (v >= 1.0f) return 255;
    const double l = v;
    const double e = l <= 0.0031308 ? 12.92 * l : 1.055 * std::pow(l, 1.0 / 2.4) - 0.055;
    return (unsigned char)(e * 255.0 + 0.5);
}

extern "C" {

uint8_t mrt_srgb8(float linear) { return srgb8(linear); }

int mrt_write_pfm(const char* path, const float* rgba, uint32_t width, uint32_t height) {
    if (!path || !rgba || !width || !height) return MRT_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, "wb");
    if (!f) return MRT_ERR_IO;
    std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height);   // negative scale = little endian; rows bottom-up
    std::vector<float> row(3 * (size_t)width);
    for (uint32_t y = 0; y < height; y++) {
        const float* src = rgba + (size_t)y * width * 4;
        for (uint32_t x = 0; x < width; x++) { row[3 * x] = src[4 * x]; row[3 * x + 1] = src[4 * x + 1]; row[3 * x + 2] = src[4 * x + 2]; }
        if (std::fwrite(row.data(), sizeof(float), row.size(), f) != row.size()) { std::fclose(f); return MRT_ERR_IO; }
    }
    return std::fclose(f) == 0 ? MRT_OK : MRT_ERR_IO;
}

int mrt_write_ppm(const char* path, const float* rgba, uint32_t width, uint32_t height) {
    if (!path || !rgba || !width || !height) return MRT_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, "wb");
    if (!f) return MRT_ERR_IO;
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    std::vector<unsigned char> row(3 * (size_t)width);
    for (uint32_t y = 0; y < height; y++) {
        const float* src = rgba + (size_t)(height - 1 - y) * width * 4;   // flip: fb row 0 is the bottom
        for (uint32_t x = 0; x < width; x++) {
            for (int ch = 0; ch < 3; ch++) row[3 * x + ch] = srgb8(src[4 * x + ch]);
        }
        if (std::fwrite(row.data(), 1, row.size(), f) != row.size()) { std::fclose(f); return MRT_ERR_IO; }
    }
    return std::fclose(f) == 0 ? MRT_OK : MRT_ERR_IO;
}

}  // extern "C"
