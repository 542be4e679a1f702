// scene_host.cpp — see scene_host.h.
#include "scene_host.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rtx {

namespace {

constexpr double kPi = 3.14159265358979323846;   // M_PI: a double wherever the reference writes it

struct F3 { float x, y, z; };
inline F3 f3(const float* p) { return F3{p[0], p[1], p[2]}; }
inline F3 sub(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 add(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline F3 scale(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
inline float dot3(F3 a, F3 b) { return ((a.x * b.x) + (a.y * b.y) + (a.z * b.z)); }                    // Core.h:178
inline F3 cross3(F3 a, F3 b) { return F3{(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)}; } // Core.h:182
inline float len3(F3 a) { return sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); }                      // Core.h:169
inline F3 norm3(F3 a) { float l = 1.0f / sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); return F3{a.x * l, a.y * l, a.z * l}; } // Core.h:173
inline float lumf(const float* c) { return ((0.2126f * c[0]) + (0.7152f * c[1]) + (0.0722f * c[2])); } // Core.h:101

// Matrix::invert, RTBase/Core.h:338-450: adjugate by cofactors, every cofactor expanded
// along the first remaining column with the reference's term order, then scaled by 1/det.
void invert4(const float m[16], float out[16])
{
    float adj[16];
    for (int idx = 0; idx < 16; idx++) {
        const int i = idx / 4, j = idx % 4;     // adj[i][j] = cofactor of m[j][i]
        int rows[3], cols[3], nr = 0, nc = 0;
        for (int r = 0; r < 4; r++) { if (r != j) rows[nr++] = r; if (r != i) cols[nc++] = r; }
        const float outer = ((i + j) & 1) ? -1.0f : 1.0f;
        float acc = 0.0f; bool first = true;
        for (int k = 0; k < 3; k++) {
            const int ra = rows[k == 0 ? 1 : 0], rb = rows[k == 2 ? 1 : 2];
            const float lead = m[rows[k] * 4 + cols[0]];
            const float sgn = outer * ((k & 1) ? -1.0f : 1.0f);
            float t0 = ((sgn * lead) * m[ra * 4 + cols[1]]) * m[rb * 4 + cols[2]];
            float t1 = ((-sgn * lead) * m[ra * 4 + cols[2]]) * m[rb * 4 + cols[1]];
            if (first) { acc = t0; first = false; } else acc = acc + t0;
            acc = acc + t1;
        }
        adj[idx] = acc;
    }
    float det = m[0] * adj[0] + m[1] * adj[4] + m[2] * adj[8] + m[3] * adj[12];
    if (det == 0) { std::memset(out, 0, 64); out[0] = out[5] = out[10] = out[15] = 1.0f; return; }
    det = 1.0f / det;
    for (int k = 0; k < 16; k++) out[k] = adj[k] * det;
}

inline F3 mul_point(const float m[16], F3 v)      // Matrix::mulPoint, Core.h:314
{ return F3{(v.x * m[0] + v.y * m[1] + v.z * m[2]) + m[3], (v.x * m[4] + v.y * m[5] + v.z * m[6]) + m[7], (v.x * m[8] + v.y * m[9] + v.z * m[10]) + m[11]}; }
inline F3 mul_vec(const float m[16], F3 v)        // Matrix::mulVec, Core.h:307
{ return F3{(v.x * m[0] + v.y * m[1] + v.z * m[2]), (v.x * m[4] + v.y * m[5] + v.z * m[6]), (v.x * m[8] + v.y * m[9] + v.z * m[10])}; }

} // namespace

void transform_instance(const float world[16], size_t n, float* positions, float* normals)
{
    float inv[16], vt[16];
    invert4(world, inv);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) vt[r * 4 + c] = inv[c * 4 + r];     // Matrix::transpose, Core.h:253-259
    const float* m = world;
    for (size_t i = 0; i < n; i++) {
        float* p = positions + 3 * i; float* q = normals + 3 * i;
        const float x = p[0], y = p[1], z = p[2];
        p[0] = (x * m[0] + y * m[1] + z * m[2]) + m[3];                                          // Matrix::mulPoint, Core.h:314-321
        p[1] = (x * m[4] + y * m[5] + z * m[6]) + m[7];
        p[2] = (x * m[8] + y * m[9] + z * m[10]) + m[11];
        const float a = q[0], b = q[1], c = q[2];
        const float nx = (a * vt[0] + b * vt[1] + c * vt[2]), ny = (a * vt[4] + b * vt[5] + c * vt[6]), nz = (a * vt[8] + b * vt[9] + c * vt[10]);   // mulVec, Core.h:307-313
        const float l = 1.0f / sqrtf((nx * nx) + (ny * ny) + (nz * nz));                         // Vec3::normalize, Core.h:173-177
        q[0] = nx * l; q[1] = ny * l; q[2] = nz * l;
    }
}

void camera_look_at(rtx_camera& c, const float from_[3], const float to_[3], const float up_[3],
                    float fov, int width, int height, int flip_x)
{
    std::memset(&c, 0, sizeof c);
    // Matrix::perspective(0.001, 10000, aspect, fov), Core.h:472-483 / SceneLoader.h:345
    const float n = 0.001f, f = 10000.0f, aspect = (float)width / (float)height;
    const float t = 1.0f / (tanf(fov * 0.5f * 3.141592654f / 180.0f));
    c.proj[0] = t / aspect; c.proj[5] = t; c.proj[10] = -f / (f - n); c.proj[11] = -(f * n) / (f - n); c.proj[14] = -1.0f;
    if (flip_x == 1) c.proj[0] = -c.proj[0];                                                     // SceneLoader.h:354-358
    // Matrix::lookAt, Core.h:451-471, then inverted (SceneLoader.h:352-353)
    const F3 from = f3(from_), to = f3(to_), up = f3(up_);
    const F3 dir = norm3(sub(from, to)), left = norm3(cross3(up, dir)), nup = cross3(dir, left);
    float V[16] = {left.x, left.y, left.z, -dot3(from, left), nup.x, nup.y, nup.z, -dot3(from, nup),
                   dir.x, dir.y, dir.z, -dot3(from, dir), 0, 0, 0, 1};
    invert4(V, c.camera);
    // Camera::init, Scene.h:22-32
    invert4(c.proj, c.inv_proj);
    c.width = width; c.height = height;
    const float Wlens = (2.0f / c.proj[5]);
    const float asp = c.proj[0] / c.proj[5];
    const float Hlens = Wlens * asp;
    c.afilm = Wlens * Hlens;
    // Camera::updateView, Scene.h:33-41
    invert4(c.camera, c.camera_to_view);
    F3 o = mul_point(c.camera, F3{0, 0, 0});
    c.origin[0] = o.x; c.origin[1] = o.y; c.origin[2] = o.z;
    F3 p = mul_point(c.inv_proj, F3{0, 0, 1});                                                    // mulPointAndPerspectiveDivide, Core.h:322
    float w = (c.inv_proj[12] * 0.0f) + (c.inv_proj[13] * 0.0f) + (c.inv_proj[14] * 1.0f) + c.inv_proj[15];
    w = 1.0f / w;
    F3 vd = norm3(mul_vec(c.camera, scale(p, w)));
    c.view_dir[0] = vd.x; c.view_dir[1] = vd.y; c.view_dir[2] = vd.z;
}

// Guide table of a CDF with n entries: guide[b] = first index whose value is >= b / n (clamped to n - 1), for
// b = 0..n. TabulatedDistribution::binarySearch (Lights.h:210-230) returns the first index whose value is >= the
// argument, which does not depend on the probing order when the table is non-decreasing; the device then only has
// to search between two guide entries. A table that is not non-decreasing and finite (an all-black row divides by
// zero) gets guide[0] = 0xFFFFFFFF and is searched the reference's way.
static void cdf_guide(const float* cdf, uint32_t n, uint32_t* guide)
{
    bool ok = n > 0 && std::isfinite(cdf[0]);
    for (uint32_t i = 1; ok && i < n; i++) ok = std::isfinite(cdf[i]) && cdf[i] >= cdf[i - 1];
    if (!ok) { for (uint32_t b = 0; b <= n; b++) guide[b] = 0xFFFFFFFFu; return; }
    uint32_t i = 0;
    for (uint32_t b = 0; b <= n; b++) {
        const float v = (float)b / (float)n;
        while (i < n - 1 && cdf[i] < v) i++;
        guide[b] = i;
    }
}

std::string prepare_scene(const rtx_scene_desc& d, HostScene& hs, const Bvh2BuildFn& build)
{
    if (d.n_triangles >= kMaxTriangles) return "too many triangles (limit 2^27 - 1)";
    if (d.n_triangles && (!d.positions || !d.normals || !d.uvs || !d.material_index)) return "triangle arrays missing";
    if (d.n_materials == 0 || !d.materials) return "no materials";
    if (d.n_textures == 0 || !d.textures) return "no textures";
    for (uint32_t i = 0; i < d.n_textures; i++)
        if (d.textures[i].width < 1 || d.textures[i].height < 1 || !d.textures[i].rgb) return "bad texture";
    for (uint32_t i = 0; i < d.n_materials; i++) {
        if (d.materials[i].texture < 0 || (uint32_t)d.materials[i].texture >= d.n_textures) return "material texture index out of range";
        const int base = d.materials[i].type & ~RTX_BSDF_LAYERED;
        if (base < RTX_BSDF_DIFFUSE || base > RTX_BSDF_ORENNAYAR) return "unknown BSDF type";
    }
    for (uint64_t i = 0; i < d.n_triangles; i++) if (d.material_index[i] >= d.n_materials) return "triangle material index out of range";
    if (d.env_rgb && (d.env_width < 2 || d.env_height < 2)) return "environment map must be at least 2x2";

    auto t0 = std::chrono::steady_clock::now();

    // textures, back to back
    std::vector<int> texOffset(d.n_textures);
    hs.texels.clear();
    for (uint32_t i = 0; i < d.n_textures; i++) {
        texOffset[i] = (int)(hs.texels.size() / 3);
        size_t n = (size_t)d.textures[i].width * d.textures[i].height * 3;
        hs.texels.insert(hs.texels.end(), d.textures[i].rgb, d.textures[i].rgb + n);
    }
    // materials: what the constructors of Materials.h compute
    hs.materials.assign(d.n_materials, MaterialDev{});
    for (uint32_t i = 0; i < d.n_materials; i++) {
        const rtx_material& m = d.materials[i]; MaterialDev& o = hs.materials[i];
        o.type = m.type & ~RTX_BSDF_LAYERED; o.tex_offset = texOffset[m.texture]; o.tex_w = d.textures[m.texture].width; o.tex_h = d.textures[m.texture].height;
        for (int k = 0; k < 3; k++) { o.emission[k] = m.emission[k]; o.eta[k] = m.eta[k]; o.k[k] = m.k[k]; }
        o.is_light = lumf(m.emission) > 0 ? 1 : 0;                                                 // Materials.h:123-126
        o.alpha = 0; o.eta_d = 1; o.inv_eta = 1; o.phong_e = 0; o.on_b = 0;
        // isTwoSided: false for Glass (Materials.h:425) and Dielectric (:475), true otherwise and for any LayeredBSDF (:716)
        o.two_sided = ((m.type & RTX_BSDF_LAYERED) || (o.type != RTX_BSDF_GLASS && o.type != RTX_BSDF_DIELECTRIC)) ? 1 : 0;
        if (o.type == RTX_BSDF_ORENNAYAR) {                                                        // Materials.h:500-501
            const float sigma = m.roughness;
            o.alpha = 1.0f - (sigma * sigma / (2.0f * ((sigma * sigma) + 0.33f)));
            o.on_b = (float)((0.45 * sigma * sigma) / (sigma * sigma + 0.09f));
        }
        if (o.type == RTX_BSDF_CONDUCTOR) o.alpha = 1.62142f * sqrtf(m.roughness);                // Materials.h:248
        if (o.type == RTX_BSDF_GLASS) { o.eta_d = m.ext_ior / m.int_ior; o.inv_eta = 1.0f / o.eta_d; } // Materials.h:368-369
        if (o.type == RTX_BSDF_PLASTIC) {                                                          // Materials.h:582-585
            o.alpha = 1.62142f * sqrtf(m.roughness);
            o.eta_d = m.ext_ior / m.int_ior;
            float a = o.alpha > 0.001f ? o.alpha : 0.001f;   // std::max(alpha, 0.001f)
            o.phong_e = (2.0f / (a * a)) - 2.0f;
        }
    }

    // triangles: Triangle::init, Geometry.h:74-92
    std::vector<TriPrep> prep;
    float smin[3], smax[3];
    prepare_triangles(d.positions, d.n_triangles, prep, smin, smax);
    hs.shade.assign(d.n_triangles, TriShade{});
    parallel_ranges(d.n_triangles, [&](uint64_t rb, uint64_t re) { for (uint64_t i = rb; i < re; i++) {
        TriShade& s = hs.shade[i];
        std::memcpy(s.n0, d.normals + i * 9, 36);
        std::memcpy(s.uv0, d.uvs + i * 6, 24);
        F3 e1 = f3(prep[i].e1), e2 = f3(prep[i].e2);
        F3 c = cross3(e1, e2);
        F3 n = norm3(c);
        s.gn[0] = n.x; s.gn[1] = n.y; s.gn[2] = n.z;
        s.gsign = dot3(f3(s.n0), n) > 0 ? 1.0f : -1.0f;                                            // Geometry.h:158-161
        s.area = len3(c) * 0.5f;
        s.material = d.material_index[i];
        const MaterialDev& m = hs.materials[s.material];
        s.shade_class = m.is_light ? kClassLight : (m.type == RTX_BSDF_MIRROR || m.type == RTX_BSDF_GLASS) ? kClassSpecular : m.type == RTX_BSDF_ORENNAYAR ? kClassOrenNayar
                        : m.type == RTX_BSDF_CONDUCTOR ? kClassConductor : m.type == RTX_BSDF_PLASTIC ? kClassPlastic : kClassDiffuse;
    } });

    auto t1 = std::chrono::steady_clock::now();
    if (build) { if (!build(prep, smin, smax, hs.bvh2)) return "device BVH build failed"; }
    else build_bvh2(prep, smin, smax, hs.bvh2);
    auto t2 = std::chrono::steady_clock::now();
    flatten_bvh4(hs.bvh2, hs.bvh4);
    // the traversal kernels address nodes by a 32-bit byte offset (128 bytes per node)
    if (hs.bvh4.nodes.size() >= (1u << 25)) return "too many BVH nodes (limit 2^25 - 1 four-wide nodes)";
    auto t3 = std::chrono::steady_clock::now();
#ifdef RTX_TUNING
    if (getenv("RTX_DEBUG_COUNTERS"))
        fprintf(stderr, "[rtx] scene preparation: triangles %.1f ms, binary BVH (%s) %.1f ms, 4-wide layout %.1f ms\n",
                std::chrono::duration<float, std::milli>(t1 - t0).count(), build ? "device" : "host",
                std::chrono::duration<float, std::milli>(t2 - t1).count(), std::chrono::duration<float, std::milli>(t3 - t2).count());
#else
    (void)t0; (void)t1; (void)t2; (void)t3;
#endif
    hs.tris.resize(d.n_triangles);
    parallel_ranges(d.n_triangles, [&](uint64_t rb, uint64_t re) { for (uint64_t p = rb; p < re; p++) {
        const uint32_t id = hs.bvh4.order[p]; const TriPrep& t = prep[id]; TriIsect& o = hs.tris[p];
        for (int a = 0; a < 3; a++) { o.e1[a] = t.e1[a]; o.e2[a] = t.e2[a]; o.v2[a] = t.v2[a]; }
    } });

    // scene bounding sphere, SceneLoader.h:387-388
    if (d.n_triangles) {
        F3 mx = f3(smax), mn = f3(smin);
        F3 centre = scale(add(mx, mn), 0.5f);
        hs.scene_radius = len3(sub(mx, centre));
        hs.scene_centre[0] = centre.x; hs.scene_centre[1] = centre.y; hs.scene_centre[2] = centre.z;
    }

    // background: BackgroundColour (Lights.h:86-135) or EnvironmentMap + TabulatedDistribution::init (Lights.h:161-208)
    float bgPower = 0;
    if (d.env_rgb) {
        hs.bg_kind = 2; hs.env_w = d.env_width; hs.env_h = d.env_height;
        const unsigned W = d.env_width, H = d.env_height;
        hs.env_texels.assign(d.env_rgb, d.env_rgb + (size_t)W * H * 3);
        hs.env_cdf_rows.assign(H, 0.f); hs.env_cdf_cols.assign((size_t)W * H, 0.f);
        float totalLum = 0.0f;
        for (unsigned y = 0; y < H; y++) {
            float sinTheta = sinf((float)(kPi * ((float)y + 0.05f) / (float)(H - 1)));
            float rowSum = 0.0f;
            for (unsigned x = 0; x < W; x++) {
                size_t idx = (size_t)y * W + x;
                float l = lumf(&hs.env_texels[idx * 3]) * sinTheta;
                hs.env_cdf_cols[idx] = l;
                rowSum += l;
            }
            hs.env_cdf_rows[y] = rowSum;
            totalLum += rowSum;
        }
        const float invW = 1.0f / (float)W, invH = 1.0f / (float)H;
        hs.env_avg_lum = totalLum * invH * invW;
        float accumRow = 0.0f;
        for (unsigned y = 0; y < H; y++) {
            float accumCol = 0.0f;
            for (unsigned x = 0; x < W; x++) {
                size_t idx = (size_t)y * W + x;
                accumCol += hs.env_cdf_cols[idx] / hs.env_cdf_rows[y];
                hs.env_cdf_cols[idx] = accumCol;
            }
            accumRow += hs.env_cdf_rows[y] / totalLum;
            hs.env_cdf_rows[y] = accumRow;
        }
        bgPower = (float)(hs.env_avg_lum * 4.0f * kPi);                                            // Lights.h:378-380
        // guide tables for the device's CDF look-ups (same result as the full binary search, fewer dependent loads)
        hs.env_guide_rows.assign((size_t)H + 1, 0u); hs.env_guide_cols.assign((size_t)H * (W + 1), 0u);
        cdf_guide(hs.env_cdf_rows.data(), H, hs.env_guide_rows.data());
        parallel_ranges(H, [&](size_t y0, size_t y1) {
            for (size_t y = y0; y < y1; y++) cdf_guide(&hs.env_cdf_cols[y * W], W, &hs.env_guide_cols[y * (W + 1)]);
        });
    } else {
        hs.bg_kind = 1; hs.env_w = hs.env_h = 0;
        for (int k = 0; k < 3; k++) hs.bg_emission[k] = d.background[k];
        bgPower = (float)(lumf(d.background) * 4.0f * kPi);                                        // Lights.h:117-120
    }

    // light list: Scene::init (Scene.h:165-168) puts the background first when it has power,
    // Scene::build (Scene.h:101-110) appends one area light per emissive triangle
    hs.lights.clear();
    if (bgPower > 0) { LightDev l{}; l.kind = hs.bg_kind; for (int k = 0; k < 3; k++) l.emission[k] = hs.bg_emission[k]; hs.lights.push_back(l); }
    for (uint64_t i = 0; i < d.n_triangles; i++) {
        const MaterialDev& m = hs.materials[d.material_index[i]];
        if (!m.is_light) continue;
        LightDev l{}; l.kind = 0; l.tri = (uint32_t)i;
        std::memcpy(l.v0, d.positions + i * 9, 36);
        const TriShade& s = hs.shade[i];
        for (int k = 0; k < 3; k++) { l.gnormal[k] = s.gn[k] * s.gsign; l.emission[k] = m.emission[k]; }
        l.area = s.area;
        hs.lights.push_back(l);
    }
    hs.build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return "";
}

} // namespace rtx
