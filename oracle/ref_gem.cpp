// ref_gem.cpp — a digest of what the REAL reference loader (RTBase/GEMLoader.h, which compiles
// on Linux as it is) reads from a scene directory: scene properties, instances, meshes.
//
// TEST INFRASTRUCTURE ONLY. Built by oracle/Makefile into oracle/_ref/libref_gem.so from the
// header where it lies under /root/reference (never copied into this repository); tests compare
// it with the digest the host's own reader (ray-tracer_amd/host/GemScene.h) computes for the
// same files.
#include <cstdint>
#include <cstring>

#include "GEMLoader.h"

namespace {
struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void bytes(const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } }
    void str(const std::string& s) { uint32_t n = (uint32_t)s.size(); bytes(&n, 4); bytes(s.data(), s.size()); }
};
}

// out[0] scene properties, out[1] their hash (name, value in iteration order), out[2] instances,
// out[3] hash of (mesh file name, 16 world floats, material properties), out[4] vertices and
// out[5] indices over all meshes of all instances, out[6] hash of the 44-byte vertex records and
// the indices, out[7] hash of the mesh-level properties
extern "C" int ref_gem_digest(const char* dir, uint64_t out[8])
{
    GEMLoader::GEMScene scene;
    scene.load(std::string(dir) + "/scene.json");
    Fnv props, inst, geo, mprops;
    for (auto& p : scene.sceneProperties) { props.str(p.name); props.str(p.value); }
    uint64_t nv = 0, ni = 0;
    for (auto& i : scene.instances) {
        inst.str(i.meshFilename); inst.bytes(i.w.m, 64);
        for (auto& p : i.material.properties) { inst.str(p.name); inst.str(p.value); }
        GEMLoader::GEMModelLoader loader;
        std::vector<GEMLoader::GEMMesh> meshes;
        loader.load(std::string(dir) + "/" + i.meshFilename, meshes);
        for (auto& m : meshes) {
            for (auto& p : m.material.properties) { mprops.str(p.name); mprops.str(p.value); }
            nv += m.verticesStatic.size(); ni += m.indices.size();
            if (!m.verticesStatic.empty()) geo.bytes(m.verticesStatic.data(), m.verticesStatic.size() * sizeof(GEMLoader::GEMStaticVertex));
            if (!m.indices.empty()) geo.bytes(m.indices.data(), m.indices.size() * 4);
        }
    }
    out[0] = scene.sceneProperties.size(); out[1] = props.h; out[2] = scene.instances.size(); out[3] = inst.h;
    out[4] = nv; out[5] = ni; out[6] = geo.h; out[7] = mprops.h;
    return 0;
}
