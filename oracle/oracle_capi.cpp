// oracle_capi.cpp — C entry points of the CPU oracle (ctypes-loadable).
//
// TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and the cpu_baseline
// leg of bench.py; never by the product. See oracle.hpp for the parity status.
#include "oracle_render.hpp"

#include <chrono>

using namespace orc;

namespace {
template <class F> void parallel_for(size_t n, unsigned threads, F f)
{
    if (threads <= 1 || n < 1024) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::vector<std::thread> th; std::atomic<size_t> next{0}; const size_t chunk = 4096;
    for (unsigned t = 0; t < threads; t++) th.emplace_back([&] {
        for (;;) { size_t b = next.fetch_add(chunk); if (b >= n) break; size_t e = b + chunk < n ? b + chunk : n; for (size_t i = b; i < e; i++) f(i); }
    });
    for (auto& t : th) t.join();
}
} // namespace

extern "C" {

struct orc_scene { Scene s; };
struct orc_renderer { Renderer r; };

orc_scene* orc_scene_create(const rtx_scene_desc* d) { auto* o = new orc_scene; o->s.load(*d); return o; }
void orc_scene_destroy(orc_scene* o) { delete o; }
double orc_scene_build_seconds(const rtx_scene_desc* d)
{
    auto t0 = std::chrono::steady_clock::now(); Scene s; s.load(*d);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
void orc_camera_look_at(rtx_camera* out, const float* from, const float* to, const float* up, float fov, int w, int h, int flip)
{
    Camera c; c.lookAtSetup(v3(from[0], from[1], from[2]), v3(to[0], to[1], to[2]), v3(up[0], up[1], up[2]), fov, w, h, flip); c.store(*out);
}
void orc_set_camera(orc_scene* o, const rtx_camera* c) { o->s.camera.load(*c); }
unsigned orc_num_lights(orc_scene* o) { return (unsigned)o->s.lights.size(); }

// BVH as the reference keeps it: bounds [n][6] = min xyz, max xyz; topo [n][4] = start,end,child_l,child_r
void orc_get_bvh(orc_scene* o, uint32_t* n_nodes, uint32_t* depth, float* bounds, uint32_t* topo, uint32_t* indices)
{
    const BVH& b = o->s.bvh;
    if (n_nodes) *n_nodes = (uint32_t)b.nodes.size();
    if (depth) *depth = b.depth;
    for (size_t i = 0; i < b.nodes.size(); i++) {
        const auto& n = b.nodes[i];
        if (bounds) { float* p = bounds + i * 6; p[0] = n.bounds.min.x; p[1] = n.bounds.min.y; p[2] = n.bounds.min.z; p[3] = n.bounds.max.x; p[4] = n.bounds.max.y; p[5] = n.bounds.max.z; }
        if (topo) { uint32_t* q = topo + i * 4; q[0] = n.start; q[1] = n.end; q[2] = n.child_l; q[3] = n.child_r; }
    }
    if (indices) std::memcpy(indices, b.indices.data(), b.indices.size() * sizeof(uint32_t));
}

// AABB::rayAABB(const Ray&), Geometry.h:213-228, on one box / one ray (for the known-answer cases)
int orc_ray_aabb(const float* bmin, const float* bmax, const float* o, const float* d)
{
    Box b; b.min = v3(bmin[0], bmin[1], bmin[2]); b.max = v3(bmax[0], bmax[1], bmax[2]);
    Ray r(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]));
    return b.rayAABB(r) ? 1 : 0;
}

void orc_generate_rays(orc_scene* o, const float* xy, size_t n, rtx_ray* rays)
{
    for (size_t i = 0; i < n; i++) {
        Ray r = o->s.camera.generateRay(xy[2 * i], xy[2 * i + 1]);
        rays[i].o[0] = r.o.x; rays[i].o[1] = r.o.y; rays[i].o[2] = r.o.z; rays[i].d[0] = r.dir.x; rays[i].d[1] = r.dir.y; rays[i].d[2] = r.dir.z;
    }
}

void orc_trace_closest(orc_scene* o, const rtx_ray* rays, size_t n, rtx_hit* hits, unsigned threads)
{
    const Scene& s = o->s;
    parallel_for(n, threads, [&](size_t i) {
        Ray r(v3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), v3(rays[i].d[0], rays[i].d[1], rays[i].d[2]));
        Hit h = s.traverse(r);
        hits[i].id = h.ID; hits[i].t = h.t; hits[i].alpha = h.alpha; hits[i].beta = h.beta; hits[i].gamma = h.gamma;
    });
}

void orc_trace_visible(orc_scene* o, const float* p, size_t n, uint8_t* vis, unsigned threads)
{
    const Scene& s = o->s;
    parallel_for(n, threads, [&](size_t i) { vis[i] = s.visible(v3(p[6 * i], p[6 * i + 1], p[6 * i + 2]), v3(p[6 * i + 3], p[6 * i + 4], p[6 * i + 5])) ? 1 : 0; });
}

// Scene::calculateShadingData, Scene.h:183-213, for n (ray, hit) pairs: out[i] = x(3) wo(3) sNormal(3) gNormal(3) tu tv u(3) v(3) w(3) material
void orc_shading_data(orc_scene* o, const rtx_ray* rays, const rtx_hit* hits, size_t n, float* out /*[n][24]*/)
{
    for (size_t i = 0; i < n; i++) {
        Ray r(v3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), v3(rays[i].d[0], rays[i].d[1], rays[i].d[2]));
        Hit h; h.ID = hits[i].id; h.t = hits[i].t; h.alpha = hits[i].alpha; h.beta = hits[i].beta; h.gamma = hits[i].gamma;
        Shading sd = o->s.calculateShadingData(h, r);
        float* q = out + i * 24; std::memset(q, 0, 24 * sizeof(float));
        if (h.t < FLT_MAX) {
            const V3 v[7] = {sd.x, sd.wo, sd.sNormal, sd.gNormal, sd.frame.u, sd.frame.v, sd.frame.w};
            int k = 0; for (int j = 0; j < 4; j++) { q[k++] = v[j].x; q[k++] = v[j].y; q[k++] = v[j].z; }
            q[k++] = sd.tu; q[k++] = sd.tv;
            for (int j = 4; j < 7; j++) { q[k++] = v[j].x; q[k++] = v[j].y; q[k++] = v[j].z; }
            q[k++] = (float)(sd.bsdf - o->s.materials.data());
        }
    }
}

// BSDF::sample / evaluate / PDF (Materials.h:139-678) on a table of inputs.
// in[i]  = normal(3) wo(3) tu tv wi_eval(3) u0 u1 u2     (14 floats; u* feed Sampler::next() in call order)
// out[i] = wi(3) colour(3) pdf  evaluate(3) PDF           (11 floats)
struct ListSampler : Sampler { const float* u; int i = 0; explicit ListSampler(const float* p) : u(p) {} float next() override { return u[i++ % 3]; } };
void orc_bsdf_table(orc_scene* o, unsigned material, const float* in, size_t n, float* out)
{
    const BSDF& b = o->s.materials[material];
    for (size_t i = 0; i < n; i++) {
        const float* p = in + i * 14; float* q = out + i * 11;
        Shading sd; sd.sNormal = v3(p[0], p[1], p[2]); sd.gNormal = sd.sNormal; sd.wo = v3(p[3], p[4], p[5]); sd.tu = p[6]; sd.tv = p[7];
        sd.frame.fromVector(sd.sNormal); sd.bsdf = &b;
        ListSampler ls(p + 11); RGB c; float pdf = 0; V3 wi = b.sample(sd, &ls, c, pdf);
        q[0] = wi.x; q[1] = wi.y; q[2] = wi.z; q[3] = c.r; q[4] = c.g; q[5] = c.b; q[6] = pdf;
        V3 we = v3(p[8], p[9], p[10]); RGB e = b.evaluate(sd, we); q[7] = e.r; q[8] = e.g; q[9] = e.b; q[10] = b.PDF(sd, we);
    }
}

// TabulatedDistribution (Lights.h:138-305) of the scene's environment map:
// rows → cdfRows[h]; cols → cdfCols[h*w]; for n (u0,u1) pairs: out = wi(3) pdf Le(3) ; pdfdir[i] = PDF(dirs[i]), Le of evaluate(dirs[i])
int orc_env_tables(orc_scene* o, float* rows, float* cols, float* avg_lum)
{
    const Light& bg = o->s.background; if (bg.kind != Light::ENVMAP) return -1;
    const TabDist& t = bg.tab;
    if (rows) std::memcpy(rows, t.cdfRows.data(), t.height * sizeof(float));
    if (cols) for (unsigned y = 0; y < t.height; y++) std::memcpy(cols + (size_t)y * t.width, t.cdfCols[y].data(), t.width * sizeof(float));
    if (avg_lum) *avg_lum = t.avgLum;
    return 0;
}
int orc_env_sample(orc_scene* o, const float* u01, size_t n, float* out /*[n][7]*/)
{
    const Light& bg = o->s.background; if (bg.kind != Light::ENVMAP) return -1;
    for (size_t i = 0; i < n; i++) {
        struct S2 : Sampler { const float* u; int i = 0; float next() override { return u[i++ & 1]; } } s; s.u = u01 + 2 * i;
        RGB e; float pdf; V3 wi = bg.sample(&s, e, pdf);
        float* q = out + i * 7; q[0] = wi.x; q[1] = wi.y; q[2] = wi.z; q[3] = pdf; q[4] = e.r; q[5] = e.g; q[6] = e.b;
    }
    return 0;
}
int orc_env_eval(orc_scene* o, const float* dirs, size_t n, float* out /*[n][4] = Le(3) pdf*/)
{
    const Light& bg = o->s.background;
    for (size_t i = 0; i < n; i++) {
        V3 d = v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]); RGB e = bg.evaluate(d);
        out[4 * i] = e.r; out[4 * i + 1] = e.g; out[4 * i + 2] = e.b; out[4 * i + 3] = bg.PDF(d);
    }
    return 0;
}

orc_renderer* orc_renderer_create(orc_scene* o, const rtx_settings* st, unsigned threads, int counter_rng)
{ auto* r = new orc_renderer; r->r.init(&o->s, *st, threads, counter_rng != 0); return r; }
void orc_renderer_destroy(orc_renderer* r) { delete r; }
void orc_renderer_clear(orc_renderer* r) { r->r.clear(); }
void orc_render_passes(orc_renderer* r, int n) { for (int i = 0; i < n; i++) r->r.renderPass(); }
unsigned orc_num_vpls(orc_renderer* r) { return (unsigned)r->r.vpls.size(); }
void orc_get_vpls(orc_renderer* r, float* out /*[n][9] = x n Le*/)
{
    for (size_t i = 0; i < r->r.vpls.size(); i++) { const VPL& v = r->r.vpls[i]; float* q = out + 9 * i; q[0] = v.x.x; q[1] = v.x.y; q[2] = v.x.z; q[3] = v.n.x; q[4] = v.n.y; q[5] = v.n.z; q[6] = v.Le.r; q[7] = v.Le.g; q[8] = v.Le.b; }
}
void orc_film_read(orc_renderer* r, float* rgbs, int* spp, uint32_t* tiles)
{
    const Film& f = r->r.film;
    if (rgbs) std::memcpy(rgbs, f.film.data(), f.film.size() * sizeof(RGB));
    if (spp) *spp = f.SPP;
    if (tiles) std::memcpy(tiles, r->r.tileSamples.data(), r->r.tileSamples.size() * sizeof(uint32_t));
}
// Film write access so calculateTileSamples can be checked on a fixed film
void orc_film_write(orc_renderer* r, const float* rgbs, int spp) { Film& f = r->r.film; std::memcpy((void*)f.film.data(), rgbs, f.film.size() * sizeof(RGB)); f.SPP = spp; }
void orc_calculate_tile_samples(orc_renderer* r) { r->r.calculateTileSamples(); }
// RayTracer::tileSamples (Renderer.h:41) set from outside: targets gathered from the ranks of a sharded render
void orc_tile_samples_write(orc_renderer* r, const uint32_t* tiles) { for (size_t i = 0; i < r->r.tileSamples.size(); i++) r->r.tileSamples[i] = tiles[i]; }
// RayTracer::createAOV, Renderer.h:786-820
void orc_create_aov(orc_renderer* r, float* color, float* albedo, float* normal) { r->r.createAOV(color, albedo, normal); }
// Film::splat (Imaging.h:318-340) of n samples (x, y, r, g, b) in order
void orc_film_splat(orc_renderer* r, const float* xyrgb, size_t n) { for (size_t i = 0; i < n; i++) r->r.film.splat(xyrgb[5 * i], xyrgb[5 * i + 1], rgb(xyrgb[5 * i + 2], xyrgb[5 * i + 3], xyrgb[5 * i + 4])); }

// ---- the oracle's math / sampling layer, entry by entry as oracle/ref_core.cpp exposes the
// real RTBase/Core.h + RTBase/Sampling.h (same signatures, so tests can diff them) ----------
static M4 m4(const float* m) { M4 a; std::memcpy(a.m, m, 64); return a; }
static void put3(float* o, V3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
void orc_matrix_invert(const float* m, float* out) { M4 r = invert(m4(m)); std::memcpy(out, r.m, 64); }
void orc_look_at(const float* f, const float* t, const float* u, float* out) { M4 r = lookAt(v3(f[0], f[1], f[2]), v3(t[0], t[1], t[2]), v3(u[0], u[1], u[2])); std::memcpy(out, r.m, 64); }
void orc_perspective(float n, float f, float aspect, float fov, float* out) { M4 r = perspective(n, f, aspect, fov); std::memcpy(out, r.m, 64); }
void orc_mul_point(const float* m, const float* v, float* out) { put3(out, mulPoint(m4(m), v3(v[0], v[1], v[2]))); }
void orc_mul_vec(const float* m, const float* v, float* out) { put3(out, mulVec(m4(m), v3(v[0], v[1], v[2]))); }
void orc_mul_point_divide(const float* m, const float* v, float* out) { put3(out, mulPointDivide(m4(m), v3(v[0], v[1], v[2]))); }
float orc_lum(const float* c) { return lum(rgb(c[0], c[1], c[2])); }
void orc_frame(const float* n, const float* vec, float* out)
{
    Frame f; f.fromVector(v3(n[0], n[1], n[2])); V3 a = v3(vec[0], vec[1], vec[2]);
    put3(out, f.u); put3(out + 3, f.v); put3(out + 6, f.w); put3(out + 9, f.toLocal(a)); put3(out + 12, f.toWorld(a));
}
void orc_warp(int kind, float r1, float r2, float* out)
{
    V3 d; float pdf;
    if (kind == 0) { d = cosineSampleHemisphere(r1, r2); pdf = cosineHemispherePDF(d); }
    else if (kind == 1) { d = uniformSampleSphere(r1, r2); pdf = uniformSpherePDF(); }
    else {   // uniformSampleHemisphere / uniformHemispherePDF, Sampling.h:35-46
        float cosTheta = r1; float sinTheta = sqrtf(1.0f - cosTheta * cosTheta); float phi = (float)(2.0f * kPi * r2);
        d = v3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta); pdf = (float)(fabsf(d.z) / (2.0f * kPi));
    }
    put3(out, d); out[3] = pdf;
}
void orc_mt_stream(unsigned seed, int n, float* out) { MTSampler s(seed); for (int i = 0; i < n; i++) out[i] = s.next(); }
// f(sampler->next(), sampler->next()) as the oracle spells it out (second parameter drawn first)
void orc_warp_from_sampler(int kind, unsigned seed, float* out)
{
    MTSampler ms(seed); Sampler* s = &ms;
    ORC_DRAW_R2_THEN_R1(s, r1, r2);
    V3 d;
    if (kind == 0) d = cosineSampleHemisphere(r1, r2);
    else if (kind == 1) d = uniformSampleSphere(r1, r2);
    else { float cosTheta = r1; float sinTheta = sqrtf(1.0f - cosTheta * cosTheta); float phi = (float)(2.0f * kPi * r2); d = v3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta); }
    put3(out, d);
}
// SphericalCoordinates, Core.h:557-572, as the Oren-Nayar restatement uses them
void orc_spherical(float theta, float phi, const float* wi, float* out)
{
    put3(out, v3(cosf(phi) * sinf(theta), sinf(phi) * sinf(theta), cosf(theta)));
    out[3] = BSDF::sphericalTheta(v3(wi[0], wi[1], wi[2])); out[4] = BSDF::sphericalPhi(v3(wi[0], wi[1], wi[2]));
}

// the counter sampler itself, for checking the device generator
float orc_rng(uint32_t seed, uint32_t stream, uint32_t item, uint32_t sample, uint32_t dim) { return rtx_rng_float(rtx_rng_key(seed, stream, item, sample), dim); }

} // extern "C"
