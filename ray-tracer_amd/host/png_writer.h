// png_writer.h — minimal 8-bit RGB PNG writer (stored deflate blocks, no compression) for
// RayTracer::savePNG, which the reference implements with stbi_write_png
// (RTBase/Renderer.h:866-869).
#pragma once

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "RayTracerHost.h"

namespace pngw {
inline uint32_t crc32(const unsigned char* d, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; } init = true; }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ d[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
inline void be32(std::vector<unsigned char>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((unsigned char)(x >> s)); }
inline void chunk(FILE* f, const char* type, const std::vector<unsigned char>& data)
{
    std::vector<unsigned char> b; be32(b, (uint32_t)data.size());
    std::vector<unsigned char> body(type, type + 4); body.insert(body.end(), data.begin(), data.end());
    b.insert(b.end(), body.begin(), body.end()); be32(b, crc32(body.data(), body.size()));
    fwrite(b.data(), 1, b.size(), f);
}
inline void write_rgb(const std::string& filename, unsigned w, unsigned h, const unsigned char* rgb)
{
    FILE* f = fopen(filename.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + filename);
    const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    fwrite(sig, 1, 8, f);
    std::vector<unsigned char> ihdr; be32(ihdr, w); be32(ihdr, h); ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
    chunk(f, "IHDR", ihdr);
    std::vector<unsigned char> raw; raw.reserve((size_t)h * (w * 3 + 1));
    for (unsigned y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3); }
    std::vector<unsigned char> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size() || pos == 0;) {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
        for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521; b = (b + a) % 65521; }
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
        if (n == 0) break;
    }
    be32(z, (b << 16) | a);
    chunk(f, "IDAT", z);
    chunk(f, "IEND", {});
    fclose(f);
}
} // namespace pngw

inline void RayTracer::savePNG(const std::string& filename)
{
    pngw::write_rgb(filename, canvas->getWidth(), canvas->getHeight(), canvas->getBackBuffer());
}
