// device_scene.h — the scene as the HIP kernels see it (plain pointers into HBM).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtx.h"

namespace rtx {

// One triangle in leaf order: the nine floats Triangle::rayIntersect reads (RTBase/Geometry.h:95-126),
// 36 bytes and nothing else — the texture path moves a fixed number of bytes per lane and clock, so
// padding and the id (needed once per ray, kept in DeviceScene::tri_ids) would cost traversal time.
struct TriIsect { float e1[3], e2[3], v2[3]; };
static_assert(sizeof(TriIsect) == 36, "packed triangle record");

// Shading attributes per ORIGINAL triangle index, 96 bytes (Triangle members read by
// Scene::calculateShadingData, RTBase/Scene.h:183-213, and by the lights).
struct TriShade {
    float n0[3], n1[3], n2[3];   // vertex normals
    float uv0[2], uv1[2], uv2[2];
    float gn[3];                 // Triangle::n, Geometry.h:85 (unflipped)
    float gsign;                 // sign applied by Triangle::gNormal, Geometry.h:158-161
    uint32_t material;
    float area;                  // Geometry.h:87
    uint32_t shade_class;        // what a hit on the triangle asks of the shade kernels (kClass*): lets them group their items before the material is read
    float pad[2];
};
// classes of shade-kernel items, ordered by how much work they are: a miss, a hit on an emitter, a hit on a purely
// specular surface (no next-event estimation, RTBase/Renderer.h:420-423), then the surfaces that sample a light by BSDF
constexpr uint32_t kClassMiss = 0, kClassLight = 1, kClassSpecular = 2, kClassDiffuse = 3, kClassOrenNayar = 4, kClassConductor = 5, kClassPlastic = 6, kClassCount = 7;
static_assert(sizeof(TriShade) == 96, "TriShade layout");

// Material record after the constructors of RTBase/Materials.h have run.
struct MaterialDev {
    int32_t type;
    int32_t tex_offset;          // first texel (in RGB triples) inside DeviceScene::texels
    int32_t tex_w, tex_h;
    float emission[3];
    int32_t is_light;            // BSDF::isLight, Materials.h:123
    float eta[3];
    float alpha;                 // 1.62142*sqrt(roughness), Materials.h:248,582; Oren-Nayar: A, Materials.h:500
    float k[3];
    float eta_d;                 // extIOR/intIOR, Materials.h:368,584
    float inv_eta;               // Materials.h:369
    float phong_e;               // Materials.h:585
    int32_t two_sided;           // BSDF::isTwoSided
    float on_b;                  // Oren-Nayar B, Materials.h:501
};

// One emitter of Scene::lights (RTBase/Scene.h:83): kind 0 = area triangle, 1 = constant
// background, 2 = environment map.
struct LightDev {
    int32_t kind;
    uint32_t tri;                // original triangle index (area)
    float v0[3], v1[3], v2[3];   // area light vertices (Triangle::sample, Geometry.h:137-156)
    float gnormal[3];            // Triangle::gNormal()
    float area;
    float emission[3];
};

struct CameraDev {
    float proj[16], inv_proj[16], camera[16], camera_to_view[16];
    float origin[3], view_dir[3];
    float afilm, width, height;
    int32_t w, h;
};

struct DeviceScene {
    const float4* nodes4;        // Node4 array as float4[8] per node
    const float4* nodes2;        // Node2Dev array as float4[2] per node
    const TriIsect* tris;        // leaf order
    const uint32_t* tri_ids;     // leaf order -> original triangle index
    const TriShade* shade;       // original order
    const MaterialDev* materials;
    const float* texels;         // all textures, RGB triples
    const LightDev* lights;
    uint32_t n_lights;
    uint32_t n_tris;
    float root_min[3], root_max[3];
    // background / environment (RTBase/Lights.h:86-135, 307-413)
    int32_t bg_kind;             // 1 constant, 2 environment map
    float bg_emission[3];
    int32_t env_w, env_h;
    const float* env_texels;     // RGB triples
    const float* env_cdf_rows;   // [h]   TabulatedDistribution::cdfRows
    const float* env_cdf_cols;   // [h*w] TabulatedDistribution::cdfCols
    const uint32_t* env_guide_rows;   // [h + 1]       first row whose CDF value is >= b / h (0xFFFFFFFF: no guide)
    const uint32_t* env_guide_cols;   // [h * (w + 1)] the same per row of cdfCols
    float env_inv_w, env_inv_h;
    float scene_centre[3], scene_radius;   // SceneBounds, SceneLoader.h:387-388
    uint32_t fine_per_wave;      // the last fine_per_wave items per wave of the grid are handed out item by item instead of in chunks (traverse_dyn.h); 0: none
    uint32_t spread_below;       // launches with fewer items per wave of the grid than this are spread thin (traverse_dyn.h); 0: never
};

} // namespace rtx
