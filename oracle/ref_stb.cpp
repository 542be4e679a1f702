// ref_stb.cpp — the image decoder the reference's Texture::load calls (RTBase/stb_image.h, vendored
// source in the reference tree, compiles on Linux as it is), behind one C entry point that
// condenses a decoded file into numbers: width, height, channels, FNV-1a hash of the pixels.
//
// TEST INFRASTRUCTURE ONLY. Built by oracle/Makefile into oracle/_ref/libref_stb.so from the header
// where it lies under /root/reference (never copied into this repository); tests compare it with
// what the host's own PNG / Radiance HDR decoders (ray-tracer_amd/host/GemScene.h) produce.
#include <cstdint>
#include <cstring>

#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"

static uint64_t fnv(const void* p, size_t n)
{
    uint64_t h = 1469598103934665603ull; const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

// as Texture::load does (RTBase/Imaging.h:39-78): ".hdr" files through stbi_loadf, others through
// stbi_load, both with the file's own channel count
extern "C" int ref_stb_digest(const char* path, int hdr, uint64_t out[4])
{
    int w = 0, h = 0, c = 0;
    if (hdr) {
        float* d = stbi_loadf(path, &w, &h, &c, 0);
        if (!d) return -1;
        out[3] = fnv(d, (size_t)w * h * c * sizeof(float)); stbi_image_free(d);
    } else {
        unsigned char* d = stbi_load(path, &w, &h, &c, 0);
        if (!d) return -1;
        out[3] = fnv(d, (size_t)w * h * c); stbi_image_free(d);
    }
    out[0] = (uint64_t)w; out[1] = (uint64_t)h; out[2] = (uint64_t)c;
    return 0;
}
