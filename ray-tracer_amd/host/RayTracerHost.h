// RayTracerHost.h — C++ host side above the C ABI (include/rtx.h).
//
// Keeps the names and call semantics of the reference's host objects for the path the
// MI355X core replaces, so that a caller written against the reference (RTBase/Main.cpp)
// reads the same:
//
//   SETTINGS, ALGORITHM, DRAW_MODE, TONEMAP, IMAGE_FILTER   RTBase/Settings.h, RTBase/Imaging.h:16-21,225-232
//   Vec3, Colour, Vertex, Matrix (lookAt / perspective / invert)  RTBase/Core.h
//   Texture, BSDF + Diffuse/Mirror/Conductor/Glass/Plastic   RTBase/Imaging.h:23-137, RTBase/Materials.h (description only)
//   Triangle, Camera, Scene (init / build / traverse / visible)   RTBase/Geometry.h:56-92, RTBase/Scene.h
//   Film (film, SPP, tonemap, save)                          RTBase/Imaging.h:234-418
//   RayTracer (init / clear / render / renderMT / getSPP / draw / saveHDR / savePNG)   RTBase/Renderer.h:25-870
//   GamesEngineeringBase::Window / Timer                     headless stand-ins for the Win32 window the reference draws into
//
// Everything per-pixel (BVH traversal, shading, sampling, film accumulation) happens behind
// rtx_render_passes(); these classes only describe the scene and move results.
#pragma once

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtx.h"

enum ALGORITHM { AL_PATH_TRACE, AL_LIGHT_TRACE, AL_INSTANT_RADIOSITY };
enum DRAW_MODE { DM_ALGORITHM, DM_NORMALS, DM_ALBEDO, DM_DIRECT };
enum IMAGE_FILTER { FT_BOX, FT_GAUSSIAN, FT_MITCHELL_NETRAVALI };
enum TONEMAP { TM_NONE, TM_LINEAR, TM_LINEAR_EXPOSURE, TM_REINHARD_GLOBAL, TM_FILMIC };

struct SETTINGS {
    ALGORITHM algorithm = AL_PATH_TRACE;
    DRAW_MODE drawMode = DM_ALGORITHM;
    TONEMAP toneMap = TM_NONE;
    IMAGE_FILTER filter = FT_BOX;
    bool useMultithreading = false, useMis = false, adaptiveSampling = false, canHitLight = false, debug = false, denoise = false;
    unsigned int numThreads = 3, maxBounces = 5, initSPP = 10, totalSPP = 8192, vplRaysPerTile = 1;
    unsigned int seed = 1;   // counter-RNG seed (the reference seeds one MT19937 per thread instead)
};

class Colour {
public:
    float r = 0, g = 0, b = 0;
    Colour() {}
    Colour(float r_, float g_, float b_) : r(r_), g(g_), b(b_) {}
    Colour operator/(float v) const { return Colour(r / v, g / v, b / v); }
    float Lum() const { return ((0.2126f * r) + (0.7152f * g) + (0.0722f * b)); }
};

class Vec3 {
public:
    float x = 0, y = 0, z = 0, w = 1.0f;
    Vec3() {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    Vec3 operator+(const Vec3& v) const { return Vec3(x + v.x, y + v.y, z + v.z); }
    Vec3 operator-(const Vec3& v) const { return Vec3(x - v.x, y - v.y, z - v.z); }
    Vec3 operator*(float s) const { return Vec3(x * s, y * s, z * s); }
    float length() const { return sqrtf((x * x) + (y * y) + (z * z)); }
    Vec3 normalize() const { float l = 1.0f / sqrtf((x * x) + (y * y) + (z * z)); return Vec3(x * l, y * l, z * l); }   // Core.h:173-177
    float dot(const Vec3& v) const { return ((x * v.x) + (y * v.y) + (z * v.z)); }
    Vec3 cross(const Vec3& v) const { return Vec3((y * v.z) - (z * v.y), (z * v.x) - (x * v.z), (x * v.y) - (y * v.x)); }
};
inline Vec3 Cross(const Vec3& a, const Vec3& b) { return a.cross(b); }

struct Vertex { Vec3 p, normal; float u = 0, v = 0; };

class Ray {
public:
    Vec3 o, dir;
    Ray() {}
    Ray(Vec3 o_, Vec3 d_) : o(o_), dir(d_) {}
};

struct IntersectionData { unsigned int ID; float t, alpha, beta, gamma; };

inline void rtxCheck(int rc, const char* what)
{
    if (rc != RTX_OK) throw std::runtime_error(std::string(what) + ": " + rtx_last_error());
}

class Texture {
public:
    std::vector<Colour> texels; int width = 1, height = 1;
    void loadDefault() { width = height = 1; texels.assign(1, Colour(1.0f, 1.0f, 1.0f)); }   // Imaging.h:31-38
    void fromPixels(int w, int h, const float* rgb) { width = w; height = h; texels.resize((size_t)w * h); for (size_t i = 0; i < texels.size(); i++) texels[i] = Colour(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]); }
};

// Material description objects with the reference's constructor signatures. Evaluation
// happens on the device; the host object only carries the parameters.
class BSDF {
public:
    Colour emission; Texture* albedo = nullptr;
    virtual ~BSDF() {}
    virtual rtx_material describe() const = 0;
    bool isLight() const { return emission.Lum() > 0; }
    void addLight(Colour e) { emission = e; }
protected:
    rtx_material base(int type) const
    {
        rtx_material m; std::memset(&m, 0, sizeof m);
        m.type = type; m.emission[0] = emission.r; m.emission[1] = emission.g; m.emission[2] = emission.b;
        m.roughness = 1.0f; m.int_ior = 1.33f; m.ext_ior = 1.0f;
        return m;
    }
};
class DiffuseBSDF : public BSDF { public: DiffuseBSDF(Texture* t) { albedo = t; } rtx_material describe() const override { return base(RTX_BSDF_DIFFUSE); } };
class MirrorBSDF : public BSDF { public: MirrorBSDF(Texture* t) { albedo = t; } rtx_material describe() const override { return base(RTX_BSDF_MIRROR); } };
class ConductorBSDF : public BSDF {
public:
    Colour eta, k; float roughness;
    ConductorBSDF(Texture* t, Colour eta_, Colour k_, float roughness_) : eta(eta_), k(k_), roughness(roughness_) { albedo = t; }
    rtx_material describe() const override { rtx_material m = base(RTX_BSDF_CONDUCTOR); m.eta[0] = eta.r; m.eta[1] = eta.g; m.eta[2] = eta.b; m.k[0] = k.r; m.k[1] = k.g; m.k[2] = k.b; m.roughness = roughness; return m; }
};
class GlassBSDF : public BSDF {
public:
    float intIOR, extIOR;
    GlassBSDF(Texture* t, float i, float e) : intIOR(i), extIOR(e) { albedo = t; }
    rtx_material describe() const override { rtx_material m = base(RTX_BSDF_GLASS); m.int_ior = intIOR; m.ext_ior = extIOR; return m; }
};
class PlasticBSDF : public BSDF {
public:
    float intIOR, extIOR, roughness;
    PlasticBSDF(Texture* t, float i, float e, float r) : intIOR(i), extIOR(e), roughness(r) { albedo = t; }
    rtx_material describe() const override { rtx_material m = base(RTX_BSDF_PLASTIC); m.int_ior = intIOR; m.ext_ior = extIOR; m.roughness = roughness; return m; }
};
class DielectricBSDF : public BSDF {      // Materials.h:436-483 (the reference's body is a cosine lobe; roughness and the IORs are stored, not used)
public:
    float intIOR, extIOR, roughness;
    DielectricBSDF(Texture* t, float i, float e, float r) : intIOR(i), extIOR(e), roughness(r) { albedo = t; }
    rtx_material describe() const override { rtx_material m = base(RTX_BSDF_DIELECTRIC); m.int_ior = intIOR; m.ext_ior = extIOR; m.roughness = roughness; return m; }
};
class OrenNayarBSDF : public BSDF {       // Materials.h:485-562
public:
    float sigma;
    OrenNayarBSDF(Texture* t, float sigma_) : sigma(sigma_) { albedo = t; }
    rtx_material describe() const override { rtx_material m = base(RTX_BSDF_ORENNAYAR); m.roughness = sigma; return m; }
};
class LayeredBSDF : public BSDF {         // Materials.h:680-724: forwards to its base BSDF, two-sided
public:
    BSDF* baseBSDF; Colour sigmaa; float thickness, intIOR, extIOR;
    LayeredBSDF(BSDF* b, Colour sa, float th, float i, float e) : baseBSDF(b), sigmaa(sa), thickness(th), intIOR(i), extIOR(e) { albedo = b->albedo; }
    rtx_material describe() const override
    {
        rtx_material m = baseBSDF->describe();
        m.type |= RTX_BSDF_LAYERED;
        m.emission[0] = emission.r; m.emission[1] = emission.g; m.emission[2] = emission.b;   // isLight()/emit() are the wrapper's own
        return m;
    }
};

class Triangle {
public:
    Vertex vertices[3]; unsigned int materialIndex = 0; float area = 0;
    void init(Vertex v0, Vertex v1, Vertex v2, unsigned int mat)     // the area test of SceneLoader.h:330 needs Geometry.h:82-87
    {
        vertices[0] = v0; vertices[1] = v1; vertices[2] = v2; materialIndex = mat;
        Vec3 e1 = v0.p - v2.p, e2 = v1.p - v2.p;
        Vec3 c((e1.y * e2.z) - (e1.z * e2.y), (e1.z * e2.x) - (e1.x * e2.z), (e1.x * e2.y) - (e1.y * e2.x));
        area = c.length() * 0.5f;
    }
};

class Camera {
public:
    rtx_camera state{}; float width = 0, height = 0; Vec3 origin, viewDirection; float Afilm = 0;
    float fov = 45.0f; int flip = 0; rtx_ctx* bound = nullptr;      // what lookAt was last called with; context to refresh on a move
    // loadScene's camera block (SceneLoader.h:342-360): perspective + lookAt, as one call
    void lookAt(Vec3 from, Vec3 to, Vec3 up, float fov, int w, int h, int flipX = 0)
    {
        const float f[3] = {from.x, from.y, from.z}, t[3] = {to.x, to.y, to.z}, u[3] = {up.x, up.y, up.z};
        rtxCheck(rtx_camera_look_at(&state, f, t, u, fov, w, h, flipX), "Camera::lookAt");
        this->fov = fov; flip = flipX;
        width = (float)w; height = (float)h; Afilm = state.afilm;
        origin = Vec3(state.origin[0], state.origin[1], state.origin[2]);
        viewDirection = Vec3(state.view_dir[0], state.view_dir[1], state.view_dir[2]);
        if (bound) rtxCheck(rtx_set_camera(bound, &state), "Camera::lookAt (device)");
    }
};

// A constant background colour or a lat-long environment map (Lights.h:86, :307)
struct Background { Colour colour; Texture* env = nullptr; };

class Scene {
public:
    std::vector<Triangle> triangles; std::vector<BSDF*> materials; Background background; Camera camera;
    rtx_ctx* ctx = nullptr;

    explicit Scene(int device = 0) { rtxCheck(rtx_create(&ctx, device), "rtx_create"); }
    ~Scene() { rtx_destroy(ctx); }
    Scene(const Scene&) = delete; Scene& operator=(const Scene&) = delete;

    void init(const std::vector<Triangle>& meshTriangles, const std::vector<BSDF*>& meshMaterials, Background bg)   // Scene.h:151-169
    { triangles = meshTriangles; materials = meshMaterials; background = bg; }

    // Scene::build, Scene.h:94-110: BVH + light list — done by the core at upload
    void build()
    {
        const size_t n = triangles.size();
        std::vector<float> pos(n * 9), nrm(n * 9), uv(n * 6); std::vector<uint32_t> mi(n);
        for (size_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) {
            const Vertex& v = triangles[i].vertices[k];
            pos[i * 9 + k * 3] = v.p.x; pos[i * 9 + k * 3 + 1] = v.p.y; pos[i * 9 + k * 3 + 2] = v.p.z;
            nrm[i * 9 + k * 3] = v.normal.x; nrm[i * 9 + k * 3 + 1] = v.normal.y; nrm[i * 9 + k * 3 + 2] = v.normal.z;
            uv[i * 6 + k * 2] = v.u; uv[i * 6 + k * 2 + 1] = v.v; mi[i] = triangles[i].materialIndex;
        }
        std::vector<Texture*> texList; std::vector<rtx_material> mats;
        for (BSDF* b : materials) {
            rtx_material m = b->describe();
            size_t t = 0; while (t < texList.size() && texList[t] != b->albedo) t++;
            if (t == texList.size()) texList.push_back(b->albedo);
            m.texture = (int32_t)t; mats.push_back(m);
        }
        std::vector<std::vector<float>> texData(texList.size()); std::vector<rtx_texture> tex(texList.size());
        for (size_t t = 0; t < texList.size(); t++) {
            const Texture& T = *texList[t]; texData[t].resize(T.texels.size() * 3);
            for (size_t i = 0; i < T.texels.size(); i++) { texData[t][3 * i] = T.texels[i].r; texData[t][3 * i + 1] = T.texels[i].g; texData[t][3 * i + 2] = T.texels[i].b; }
            tex[t].width = T.width; tex[t].height = T.height; tex[t].rgb = texData[t].data();
        }
        std::vector<float> env;
        rtx_scene_desc d; std::memset(&d, 0, sizeof d);
        d.n_triangles = n; d.positions = pos.data(); d.normals = nrm.data(); d.uvs = uv.data(); d.material_index = mi.data();
        d.n_materials = (uint32_t)mats.size(); d.materials = mats.data(); d.n_textures = (uint32_t)tex.size(); d.textures = tex.data();
        d.background[0] = background.colour.r; d.background[1] = background.colour.g; d.background[2] = background.colour.b;
        if (background.env) {
            const Texture& E = *background.env; env.resize(E.texels.size() * 3);
            for (size_t i = 0; i < E.texels.size(); i++) { env[3 * i] = E.texels[i].r; env[3 * i + 1] = E.texels[i].g; env[3 * i + 2] = E.texels[i].b; }
            d.env_width = E.width; d.env_height = E.height; d.env_rgb = env.data();
        }
        rtxCheck(rtx_upload_scene(ctx, &d), "Scene::build");
        rtxCheck(rtx_set_camera(ctx, &camera.state), "Scene::build (camera)");
        camera.bound = ctx;
    }
    IntersectionData traverse(const Ray& ray)                               // Scene.h:137-141
    {
        rtx_ray r = {{ray.o.x, ray.o.y, ray.o.z}, {ray.dir.x, ray.dir.y, ray.dir.z}}; rtx_hit h;
        rtxCheck(rtx_trace_closest(ctx, &r, 1, &h), "Scene::traverse");
        return IntersectionData{h.id, h.t, h.alpha, h.beta, h.gamma};
    }
    bool visible(const Vec3& p1, const Vec3& p2)                            // Scene.h:170-178
    {
        const float s[6] = {p1.x, p1.y, p1.z, p2.x, p2.y, p2.z}; uint8_t v = 0;
        rtxCheck(rtx_trace_visible(ctx, s, 1, &v), "Scene::visible");
        return v != 0;
    }
};

// The fly-through camera of the reference's viewer, RTBase/SceneLoader.h:10-191. A move
// recomputes the view (updateCamera -> Camera::updateView, :115-120) and hands it to the core;
// the caller then calls RayTracer::clear() as RTBase/Main.cpp:158-162 does.
class RTCamera {
public:
    Vec3 from, to, up; Camera* camera = nullptr; float movespeed = 1.0f, rotspeed = 5.0f;
    void forward() { Vec3 dir = (to - from).normalize(); from = from + dir * movespeed; to = from + dir; }
    void back() { Vec3 dir = (to - from).normalize(); from = from - dir * movespeed; to = from + dir; }
    void right() { Vec3 dir = (to - from).normalize(); Vec3 r = Cross(dir, up); from = from + r * movespeed; to = from + dir; }
    void left() { Vec3 dir = (to - from).normalize(); Vec3 r = Cross(dir, up); from = from - r * movespeed; to = from + dir; }
    void flyUp() { Vec3 dir = up * movespeed; from = from + dir; to = to + dir; }
    void flyDown() { Vec3 dir = up * movespeed; from = from - dir; to = to - dir; }
    static Vec3 rotate(const Vec3& v, const Vec3& k, float angle)       // Rodrigues, as written at :79,:96,:112
    {
        float rad = (float)(angle * (M_PI / 180.0f));
        float cosTheta = cosf(rad), sinTheta = sinf(rad);
        return (v * cosTheta) + (k.cross(v) * sinTheta) + (k * (k.dot(v) * (1 - cosTheta)));
    }
    void yaw(float angle) { Vec3 dir = (to - from).normalize(); to = from + rotate(dir, up.normalize(), angle); }
    void pitch(float angle) { Vec3 dir = (to - from).normalize(); to = from + rotate(dir, dir.cross(up).normalize(), angle); }
    void roll(float angle) { Vec3 dir = (to - from).normalize(); up = rotate(up, dir, angle).normalize(); }
    void updateCamera() { camera->lookAt(from, to, up, camera->fov, (int)camera->width, (int)camera->height, camera->flip); }
    template <class WindowT> bool update(WindowT& canvas)                // key bindings of :122-190
    {
        bool change = false;
        if (canvas.keyPressed('W')) { forward(); change = true; }
        if (canvas.keyPressed('S')) { back(); change = true; }
        if (canvas.keyPressed('A')) { left(); change = true; }
        if (canvas.keyPressed('D')) { right(); change = true; }
        if (canvas.keyPressed('E')) { flyUp(); change = true; }
        if (canvas.keyPressed('Q')) { flyDown(); change = true; }
        if (canvas.keyPressed(0x25)) { if (canvas.keyPressed(0x10)) roll(-rotspeed); else yaw(rotspeed); change = true; }     // VK_LEFT, VK_SHIFT
        if (canvas.keyPressed(0x27)) { if (canvas.keyPressed(0x10)) roll(rotspeed); else yaw(-rotspeed); change = true; }     // VK_RIGHT
        if (canvas.keyPressed(0x26)) { pitch(rotspeed); change = true; }                                                      // VK_UP
        if (canvas.keyPressed(0x28)) { pitch(-rotspeed); change = true; }                                                     // VK_DOWN
        if (change) updateCamera();
        return change;
    }
};

namespace GamesEngineeringBase {
// Headless stand-in for the Win32/D3D11 window of the reference: a back buffer in memory.
class Window {
    std::vector<unsigned char> back; unsigned int w = 0, h = 0;
public:
    void create(unsigned int width, unsigned int height, const std::string& = "", float = 1.0f) { w = width; h = height; back.assign((size_t)w * h * 3, 0); }
    unsigned int getWidth() const { return w; }
    unsigned int getHeight() const { return h; }
    void draw(int pixelIndex, unsigned char r, unsigned char g, unsigned char b) { back[(size_t)pixelIndex * 3] = r; back[(size_t)pixelIndex * 3 + 1] = g; back[(size_t)pixelIndex * 3 + 2] = b; }
    unsigned char* getBackBuffer() { return back.data(); }
    void clear() {}
    void present() {}
    void checkInput() {}
    bool keyPressed(int) const { return false; }
    bool isQuitRequested() const { return false; }
};
class Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
public:
    void reset() { t0 = std::chrono::steady_clock::now(); }
    float dt() { auto t1 = std::chrono::steady_clock::now(); float s = std::chrono::duration<float>(t1 - t0).count(); t0 = t1; return s; }
};
} // namespace GamesEngineeringBase

class Film {                                                                // Imaging.h:234-418
public:
    Colour* film = nullptr; unsigned int width = 0, height = 0; int SPP = 0;
    ~Film() { delete[] film; }
    void init(int w, int h) { width = w; height = h; delete[] film; film = new Colour[(size_t)w * h]; SPP = 0; }
    // tonemap(x, y, ..., spp, toneMap), Imaging.h:343-373 (Filmic constants :306-308)
    void tonemap(int x, int y, unsigned char& r, unsigned char& g, unsigned char& b, int spp, TONEMAP tm = TM_LINEAR) const
    {
        Colour p = film[(size_t)y * width + x] / (float)spp;
        tonemap(p.r, p.g, p.b, r, g, b, tm);
    }
    // tonemap(r, g, b, ...), Imaging.h:236-316: an already normalised colour
    void tonemap(float pr, float pg, float pb, unsigned char& r, unsigned char& g, unsigned char& b, TONEMAP tm = TM_LINEAR) const
    {
        float c[3] = {std::fmax(pr, 0.0f), std::fmax(pg, 0.0f), std::fmax(pb, 0.0f)};
        const float inv2p2 = 1.0f / 2.2f;
        for (float& v : c) {
            switch (tm) {
            case TM_NONE: v *= 255; break;
            case TM_LINEAR: v = powf(std::fmax(v, 0.0f), inv2p2) * 255; break;
            case TM_LINEAR_EXPOSURE: v = powf(std::fmax(v, 0.0f), inv2p2) * std::pow(2.0f, 1.0f * inv2p2) * 255; break;
            case TM_REINHARD_GLOBAL: v = powf(std::fmax(v / (1.0f + v), 0.0f), inv2p2) * 255; break;
            case TM_FILMIC: {
                const float A = 0.15f, B = 0.5f, C = 0.1f, D = 0.2f, E = 0.02f, F = 0.3f, W = 11.2f;
                auto cx = [&](float x_) { return std::fabs((x_ * (A * x_ + C * B) + D * E) / (x_ * (A * x_ + B) + D * F) - E / F); };
                v = cx(v) * (1.0f / cx(W)) * 255.0f; break;
            }
            }
            v = std::fmin(v, 255.f);
        }
        r = (unsigned char)c[0]; g = (unsigned char)c[1]; b = (unsigned char)c[2];
    }
    // Film::save, Imaging.h:408-417: film / SPP as a Radiance .hdr (flat RGBE scanlines)
    void save(const std::string& filename) const
    {
        FILE* f = fopen(filename.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write " + filename);
        fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %u +X %u\n", height, width);
        for (size_t i = 0; i < (size_t)width * height; i++) {
            Colour p = film[i] / (float)SPP; unsigned char e4[4] = {0, 0, 0, 0};
            float m = std::fmax(p.r, std::fmax(p.g, p.b));
            if (m > 1e-32f) { int e; float s = frexpf(m, &e) * 256.0f / m; e4[0] = (unsigned char)(p.r * s); e4[1] = (unsigned char)(p.g * s); e4[2] = (unsigned char)(p.b * s); e4[3] = (unsigned char)(e + 128); }
            fwrite(e4, 1, 4, f);
        }
        fclose(f);
    }
};

// The denoiser's input / output planes, RTBase/Denoiser.h:14-36
struct AOV {
    std::vector<float> albedo, normal, color, output; unsigned int width = 0, height = 0;
    AOV() = default;
    AOV(unsigned int w, unsigned int h) : width(w), height(h)
    { albedo.resize((size_t)w * h * 3, 0); normal.resize((size_t)w * h * 3, 0); color.resize((size_t)w * h * 3, 0); output.resize((size_t)w * h * 3, 0); }
};

class RayTracer {                                                           // Renderer.h:25-870
public:
    Scene* scene = nullptr; GamesEngineeringBase::Window* canvas = nullptr; Film* film = nullptr; SETTINGS settings;
    std::vector<unsigned int> tileSamples; unsigned int totalTiles = 0, totalXTiles = 0; const unsigned int tileSize = 16;
    // extension (the reference renders on one device): which share of a multi-GPU render this process takes; set before init().
    // mode 1: the 16x16 tiles t with t % world == rank; mode 2: every pixel, the global samples s with s % world == rank
    int shardMode = 0, shardRank = 0, shardWorld = 1;

    ~RayTracer() { delete film; }
    void init(Scene* scene_, GamesEngineeringBase::Window* canvas_, SETTINGS s)   // Renderer.h:69-87
    {
        scene = scene_; canvas = canvas_; settings = s;
        delete film; film = new Film(); film->init((int)scene->camera.width, (int)scene->camera.height);
        settings.adaptiveSampling = settings.adaptiveSampling && settings.algorithm == AL_PATH_TRACE;
        if (!(settings.adaptiveSampling && settings.useMultithreading)) settings.initSPP = settings.totalSPP;
        totalXTiles = (film->width + tileSize - 1) / tileSize;
        totalTiles = totalXTiles * ((film->height + tileSize - 1) / tileSize);
        tileSamples.assign(totalTiles, 1);
        rtx_settings r; std::memset(&r, 0, sizeof r);
        r.algorithm = settings.algorithm; r.draw_mode = settings.drawMode; r.filter = settings.filter;
        r.use_mis = settings.useMis; r.can_hit_light = settings.canHitLight; r.adaptive_sampling = settings.adaptiveSampling;
        r.max_bounces = settings.maxBounces; r.init_spp = settings.initSPP; r.total_spp = settings.totalSPP;
        r.vpl_rays_per_tile = settings.vplRaysPerTile; r.num_threads = settings.numThreads; r.seed = settings.seed;
        r.shard_mode = shardWorld > 1 ? shardMode : 0; r.shard_rank = shardWorld > 1 ? shardRank : 0; r.shard_world = shardWorld > 1 ? shardWorld : 1;
        rtxCheck(rtx_set_settings(scene->ctx, &r), "RayTracer::init");
        rtxCheck(rtx_clear(scene->ctx), "RayTracer::init (clear)");
    }
    void clear() { rtxCheck(rtx_clear(scene->ctx), "RayTracer::clear"); film->SPP = 0; }      // Renderer.h:117
    // one sample per pixel per call, blocking — Renderer.h:577 / :753 (the device path has one driver for both)
    void renderMT() { rtxCheck(rtx_render_passes(scene->ctx, 1), "RayTracer::renderMT"); rtxCheck(rtx_get_spp(scene->ctx, &film->SPP), "getSPP"); }
    void render() { renderMT(); }
    // extension: several passes per call let the core keep three of them in flight (~30 % more samples per second)
    void renderMT(int passes) { rtxCheck(rtx_render_passes(scene->ctx, passes), "RayTracer::renderMT"); rtxCheck(rtx_get_spp(scene->ctx, &film->SPP), "getSPP"); }
    // extension: all totalSPP passes of a render that is split over the GPUs of `coll`'s communicator (this process's share of
    // them; one meeting at initSPP for adaptive sampling, the film sum at the end); every rank ends with the complete film
    void renderSharded(const rtx_collective* coll) { rtxCheck(rtx_render_sharded(scene->ctx, coll), "RayTracer::renderSharded"); rtxCheck(rtx_get_spp(scene->ctx, &film->SPP), "getSPP"); }
    int getSPP() const { return film->SPP; }                                                   // Renderer.h:856
    // pull the accumulated film and the tile targets to the host (film->film, tileSamples)
    void fetch() { rtxCheck(rtx_film_read(scene->ctx, &film->film[0].r, &film->SPP, tileSamples.data()), "RayTracer::fetch"); }
    void draw()                                                                                // Renderer.h:822-837
    {
        fetch();
        unsigned char r, g, b;
        for (unsigned int y = 0; y < film->height; y++) for (unsigned int x = 0; x < film->width; x++) {
            unsigned int idx = (y / tileSize) * totalXTiles + x / tileSize;
            int spp = settings.adaptiveSampling && (int)settings.initSPP < film->SPP ? (int)std::min<unsigned>(tileSamples[idx], (unsigned)film->SPP) : film->SPP;
            film->tonemap((int)x, (int)y, r, g, b, spp, settings.toneMap);
            canvas->draw((int)(y * film->width + x), r, g, b);
        }
    }
    // Renderer.h:786-820: colour, albedo and normal planes for a denoiser (output is left for it to fill)
    void createAOV(AOV& aov)
    {
        aov = AOV(film->width, film->height);
        rtxCheck(rtx_create_aov(scene->ctx, aov.color.data(), aov.albedo.data(), aov.normal.data()), "RayTracer::createAOV");
    }
    void draw(const AOV& aov)                                                                  // Renderer.h:839-854
    {
        unsigned char r, g, b;
        for (size_t i = 0; i < (size_t)film->width * film->height; i++) {
            film->tonemap(aov.output[i * 3], aov.output[i * 3 + 1], aov.output[i * 3 + 2], r, g, b, settings.toneMap);
            canvas->draw((int)i, r, g, b);
        }
    }
    void saveHDR(const std::string& filename) { fetch(); film->save(filename); }              // Renderer.h:861
    void savePNG(const std::string& filename);                                                 // Renderer.h:866 (defined in png_writer.h)
};
static_assert(sizeof(Colour) == 12, "film rows are read as packed RGB floats");
