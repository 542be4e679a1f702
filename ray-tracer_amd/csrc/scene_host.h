// scene_host.h — host-side preparation of the device scene (everything Scene::init,
// Scene::build, the material constructors and TabulatedDistribution::init compute once).
#pragma once

#include <functional>
#include <string>
#include <vector>

#include "bvh_host.h"
#include "device_scene.h"

namespace rtx {

struct HostScene {
    Bvh2 bvh2;
    Bvh4 bvh4;
    std::vector<TriIsect> tris;        // leaf order
    std::vector<TriShade> shade;       // original order
    std::vector<MaterialDev> materials;
    std::vector<float> texels;         // all textures back to back, RGB
    std::vector<LightDev> lights;      // Scene::lights order: background first (if it has power), then emissive triangles by index
    int bg_kind = 1;
    float bg_emission[3] = {0, 0, 0};
    int env_w = 0, env_h = 0;
    std::vector<float> env_texels, env_cdf_rows, env_cdf_cols;
    std::vector<uint32_t> env_guide_rows, env_guide_cols;   // [h + 1], [h * (w + 1)]: see cdf_guide() in scene_host.cpp
    float env_avg_lum = 0;
    float scene_centre[3] = {0, 0, 0}, scene_radius = 0;
    float build_ms = 0;
};

// Returns an empty string on success, otherwise the reason the description was rejected.
// `build` replaces build_bvh2() when given (the device build); it returns false on failure.
using Bvh2BuildFn = std::function<bool(const std::vector<TriPrep>&, const float*, const float*, Bvh2&)>;
std::string prepare_scene(const rtx_scene_desc& d, HostScene& out, const Bvh2BuildFn& build = nullptr);

// Camera::init + Camera::updateView (RTBase/Scene.h:22-41) after loadScene's matrix set-up
// (RTBase/SceneLoader.h:342-360).
// loadInstance's vertex transform, RTBase/SceneLoader.h:298-316
void transform_instance(const float world[16], size_t n, float* positions, float* normals);

void camera_look_at(rtx_camera& out, const float from[3], const float to[3], const float up[3],
                    float fov_deg, int width, int height, int flip_x);

} // namespace rtx
