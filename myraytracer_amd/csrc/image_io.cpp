// Image writers (host only).  The reference never leaves the GPU: it presents the
// accumulated RGBA32F texture to a window surface (raytracer/src/lib.rs:270-297,
// sample_framebuffer.wgsl:38-41, with a Y flip at :24).  These writers are the headless
// stand-in for that present pass.  The surface the reference draws to is the adapter's default
// format (surface.get_default_config, lib.rs:349-351; the fragment target takes that format,
// lib.rs:1133), an sRGB format on every wgpu backend: the hardware applies the sRGB OETF to the
// linear value fs_main returns (sample_framebuffer.wgsl:38-41) and rounds to 8 bits.  mrt_write_ppm
// does the same conversion on the host.

#include <cmath>
#include <cstdint>
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

// PNG, 8-bit RGB, the same sRGB encoding as the PPM, rows top-down.  Self-contained: the image data go into "stored"
// (uncompressed) deflate blocks inside a zlib stream -- a renderer's noisy output barely compresses anyway -- so no zlib is
// linked.  CRC-32 (ISO 3309) per chunk, Adler-32 over the raw scanlines.
static uint32_t crc32_update(uint32_t crc, const unsigned char* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}
static void put_be32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16)); v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
static bool write_chunk(FILE* f, const char type[4], const std::vector<unsigned char>& data) {
    std::vector<unsigned char> head;
    put_be32(head, (uint32_t)data.size());
    head.insert(head.end(), type, type + 4);
    uint32_t crc = crc32_update(0xFFFFFFFFu, head.data() + 4, 4);
    crc = crc32_update(crc, data.data(), data.size()) ^ 0xFFFFFFFFu;
    std::vector<unsigned char> tail;
    put_be32(tail, crc);
    return std::fwrite(head.data(), 1, head.size(), f) == head.size() &&
           (data.empty() || std::fwrite(data.data(), 1, data.size(), f) == data.size()) &&
           std::fwrite(tail.data(), 1, 4, f) == 4;
}

int mrt_write_png(const char* path, const float* rgba, uint32_t width, uint32_t height) {
    if (!path || !rgba || !width || !height) return MRT_ERR_INVALID_ARG;
    if ((uint64_t)width * height > (1ull << 28)) return MRT_ERR_INVALID_ARG;
    // raw scanlines: filter byte 0 + RGB, top row first (fb row 0 is the bottom)
    const size_t stride = 1 + 3 * (size_t)width;
    std::vector<unsigned char> raw(stride * height);
    for (uint32_t y = 0; y < height; y++) {
        const float* src = rgba + (size_t)(height - 1 - y) * width * 4;
        unsigned char* dst = raw.data() + y * stride;
        dst[0] = 0;
        for (uint32_t x = 0; x < width; x++)
            for (int ch = 0; ch < 3; ch++) dst[1 + 3 * x + ch] = srgb8(src[4 * x + ch]);
    }
    std::vector<unsigned char> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);                    // zlib header: deflate, 32 K window, no preset dictionary
    uint32_t a = 1, b = 0;                                    // Adler-32
    for (size_t off = 0; off < raw.size(); off += 65535) {
        const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + n == raw.size() ? 1 : 0);           // BFINAL, BTYPE = 00 (stored)
        z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + n));
        for (size_t i = 0; i < n; i++) { a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
    }
    put_be32(z, (b << 16) | a);
    FILE* f = std::fopen(path, "wb");
    if (!f) return MRT_ERR_IO;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit, RGB, deflate, adaptive, no interlace
    const std::vector<unsigned char> srgb_chunk = {0};       // sRGB chunk: rendering intent "perceptual"
    bool ok = std::fwrite(sig, 1, 8, f) == 8 && write_chunk(f, "IHDR", ihdr) && write_chunk(f, "sRGB", srgb_chunk) &&
              write_chunk(f, "IDAT", z) && write_chunk(f, "IEND", {});
    ok = (std::fclose(f) == 0) && ok;
    return ok ? MRT_OK : MRT_ERR_IO;
}

}  // extern "C"
