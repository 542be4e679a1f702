// shade.h — device-side shading: sampler, shading data, textures, BSDFs, lights,
// environment map. Operation order, the double-precision M_PI sub-expressions and the
// order of Sampler::next() calls follow the reference files cited at each function.
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>

#include "../../include/rtx_rng.h"
#include "device_scene.h"

namespace rtx {

constexpr double kPiD = 3.14159265358979323846;   // M_PI (double in the reference's expressions)
constexpr float kEps = 0.001f;                     // EPSILON, RTBase/Geometry.h:54

struct V3 { float x, y, z; };
struct C3 { float r, g, b; };

#define RTX_DI __device__ __forceinline__
RTX_DI V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
RTX_DI V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
RTX_DI V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RTX_DI V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RTX_DI V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RTX_DI V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
RTX_DI float dot(V3 a, V3 b) { return ((a.x * b.x) + (a.y * b.y) + (a.z * b.z)); }                     // Core.h:178
RTX_DI V3 cross(V3 a, V3 b) { return mk((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)); } // Core.h:182
RTX_DI float lengthSq(V3 a) { return ((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); }
RTX_DI V3 normalize(V3 a) { float l = 1.0f / sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); return mk(a.x * l, a.y * l, a.z * l); } // Core.h:173
RTX_DI C3 mkc(float r, float g, float b) { C3 c; c.r = r; c.g = g; c.b = b; return c; }
RTX_DI C3 ldc(const float* p) { return mkc(p[0], p[1], p[2]); }
RTX_DI C3 operator+(C3 a, C3 b) { return mkc(a.r + b.r, a.g + b.g, a.b + b.b); }
RTX_DI C3 operator-(C3 a, C3 b) { return mkc(a.r - b.r, a.g - b.g, a.b - b.b); }
RTX_DI C3 operator*(C3 a, C3 b) { return mkc(a.r * b.r, a.g * b.g, a.b * b.b); }
RTX_DI C3 operator/(C3 a, C3 b) { return mkc(a.r / b.r, a.g / b.g, a.b / b.b); }
RTX_DI C3 operator*(C3 a, float v) { return mkc(a.r * v, a.g * v, a.b * v); }
RTX_DI C3 operator/(C3 a, float v) { return mkc(a.r / v, a.g / v, a.b / v); }
RTX_DI float lum(C3 c) { return ((0.2126f * c.r) + (0.7152f * c.g) + (0.0722f * c.b)); }              // Core.h:101
// Windows min/max macros used by RTBase/Renderer.h (second operand wins on NaN)
RTX_DI float wmin(float a, float b) { return (a < b) ? a : b; }
RTX_DI float wmax(float a, float b) { return (a > b) ? a : b; }

// Sampler::next() of the counter generator (include/rtx_rng.h)
struct Rng {
    uint32_t key, dim;
    RTX_DI float next() { return rtx_rng_float(key, dim++); }
};

struct Frame {                                   // Frame, RTBase/Core.h:519-555
    V3 u, v, w;
    RTX_DI void fromVector(V3 n)
    {
        w = normalize(n);
        if (fabsf(w.x) > fabsf(w.y)) { float l = 1.0f / sqrtf(w.x * w.x + w.z * w.z); u = mk(w.z * l, 0.0f, -w.x * l); }
        else { float l = 1.0f / sqrtf(w.y * w.y + w.z * w.z); u = mk(0, w.z * l, -w.y * l); }
        v = cross(w, u);
    }
    RTX_DI V3 toLocal(V3 a) const { return mk(dot(a, u), dot(a, v), dot(a, w)); }
    RTX_DI V3 toWorld(V3 a) const { return ((u * a.x) + (v * a.y) + (w * a.z)); }
};

struct Shading {                                 // ShadingData, RTBase/Materials.h:11-31
    V3 x, wo, sNormal, gNormal; float tu, tv; Frame frame; const MaterialDev* mat; float t;
    C3 alb;                                      // albedo->sample(tu, tv), fetched once per shading point
};

// SamplingDistributions, RTBase/Sampling.h:48-72
RTX_DI V3 cosineSampleHemisphere(float r1, float r2)
{
    float cosTheta = sqrtf(r1);
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    float phi = (float)(2.0f * kPiD * r2);
    return mk(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
}
RTX_DI float cosineHemispherePDF(V3 wi) { return (float)(fabsf(wi.z) / kPiD); }
RTX_DI V3 uniformSampleSphere(float r1, float r2)
{
    float cosTheta = 1.0f - 2.0f * r1;
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    float phi = (float)(2.0f * kPiD * r2);
    return mk(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
}
RTX_DI float uniformSpherePDF() { return (float)(1.0f / (4.0f * kPiD)); }
// f(sampler->next(), sampler->next()) call sites: the first draw feeds the SECOND parameter
// (right-to-left argument evaluation of the compilers the reference is built with)
#define RTX_DRAW_R2_THEN_R1(rng, r1, r2) float r2 = (rng).next(); float r1 = (rng).next()

// Texture::sample, RTBase/Imaging.h:79-101
RTX_DI C3 tex_sample(const float* texels, int width, int height, float tu, float tv)
{
    float u = fmaxf(0.0f, fabsf(tu)) * width;
    float v = fmaxf(0.0f, fabsf(tv)) * height;
    int x = (int)floorf(u);
    int y = (int)floorf(v);
    float frac_u = u - x;
    float frac_v = v - y;
    float w0 = (1.0f - frac_u) * (1.0f - frac_v);
    float w1 = frac_u * (1.0f - frac_v);
    float w2 = (1.0f - frac_u) * frac_v;
    float w3 = frac_u * frac_v;
    x = x % width;
    y = y % height;
    int x1 = (x + 1) % width, y1 = (y + 1) % height;
    C3 s0 = ldc(texels + 3 * (y * width + x));
    C3 s1 = ldc(texels + 3 * (y * width + x1));
    C3 s2 = ldc(texels + 3 * (y1 * width + x));
    C3 s3 = ldc(texels + 3 * (y1 * width + x1));
    return (s0 * w0) + (s1 * w1) + (s2 * w2) + (s3 * w3);
}
RTX_DI C3 albedo(const DeviceScene&, const Shading& sd) { return sd.alb; }

// ShadingHelper, RTBase/Materials.h:44-112
RTX_DI float fresnelDielectric(float cosTheta, float eta)
{
    float sin2ThetaT = eta * eta * (1.0f - cosTheta * cosTheta);
    if (sin2ThetaT >= 1.0f) return 1.0f;
    float cosThetaT = sqrtf(1.0f - sin2ThetaT);
    float fParl = (cosTheta - eta * cosThetaT) / (cosTheta + eta * cosThetaT);
    float fPerp = (eta * cosTheta - cosThetaT) / (eta * cosTheta + cosThetaT);
    return (fParl * fParl + fPerp * fPerp) * 0.5f;
}
RTX_DI C3 fresnelConductor(float cosTheta, C3 ior, C3 k)
{
    cosTheta = (cosTheta < 0.0f) ? 0.0f : (cosTheta > 1.0f) ? 1.0f : cosTheta;
    C3 eta2k2 = ior * ior + k * k;
    C3 one = mkc(1.0f, 1.0f, 1.0f);
    C3 cos2Theta = one * cosTheta * cosTheta;
    C3 sin2Theta = one * (1.0f - cosTheta * cosTheta);
    C3 fParl = (eta2k2 * cos2Theta - ior * 2.0f * cosTheta + sin2Theta) / (eta2k2 * cos2Theta + ior * 2.0f * cosTheta + sin2Theta);
    C3 fPerp = (eta2k2 - ior * 2.0f * cosTheta + cos2Theta) / (eta2k2 + ior * 2.0f * cosTheta + cos2Theta);
    return (fParl * fParl + fPerp * fPerp) * 0.5f;
}
RTX_DI float lambdaGGX(V3 wi, float alpha)
{
    float cosTheta = fabsf(wi.z);
    float cos2Theta = cosTheta * cosTheta;
    float tan2Theta = (1.0f - cos2Theta) / cos2Theta;
    return (sqrtf(1.0f + alpha * alpha * tan2Theta) - 1.0f) * 0.5f;
}
RTX_DI float Gggx(V3 wi, V3 wo, float alpha)
{
    float g1Wi = 1.0f / (1.0f + lambdaGGX(wi, alpha));
    float g1Wo = 1.0f / (1.0f + lambdaGGX(wo, alpha));
    return g1Wi * g1Wo;
}
RTX_DI float Dggx(V3 h, float alpha)
{
    float cos2Theta = h.z * h.z;
    float alpha2 = alpha * alpha;
    float denom = cos2Theta * (alpha2 - 1.0f) + 1.0f;
    return (float)(alpha2 / (kPiD * denom * denom));
}

RTX_DI bool isPureSpecular(const MaterialDev& m) { return m.type == RTX_BSDF_MIRROR || m.type == RTX_BSDF_GLASS; } // Materials.h:219,420
RTX_DI bool isTwoSided(const MaterialDev& m) { return m.two_sided != 0; }                                           // Materials.h:169,425,475,553,716

// SphericalCoordinates, RTBase/Core.h:564-572
RTX_DI float sphericalTheta(V3 w) { return acosf(w.z); }
RTX_DI float sphericalPhi(V3 w) { float p = atan2f(w.y, w.x); return (p < 0.0f) ? (float)(p + (2.0f * kPiD)) : p; }
// the angular factor of OrenNayarBSDF, Materials.h:517,536, as written there
RTX_DI float orenNayarTerm(const MaterialDev& m, float thetaI, float phiI, float thetaO, float phiO)
{
    float c = cosf(phiI - phiO);
    float cpos = (c < 0.0f) ? 0.0f : c;                    // std::max(c, 0.0f)
    float hi = (thetaI < thetaO) ? thetaO : thetaI;        // std::max(thetaI, thetaO)
    float lo = (thetaO < thetaI) ? thetaO : thetaI;        // std::min(thetaI, thetaO)
    return (m.alpha + m.on_b * cpos * sinf(hi * tanf(lo)));
}

// BSDF::sample, RTBase/Materials.h:148,193,251,372,451,504,588. Sampler: anything with next() (the counter generator in
// the renderer, a list of given numbers in the function-level parity tables)
template <class Sampler>
RTX_DI V3 bsdf_sample(const DeviceScene& sc, const Shading& sd, Sampler& rng, C3& col, float& pdf)
{
    const MaterialDev& m = *sd.mat;
    if (m.type == RTX_BSDF_DIFFUSE || m.type == RTX_BSDF_DIELECTRIC) {
        RTX_DRAW_R2_THEN_R1(rng, r1, r2);
        V3 wi = cosineSampleHemisphere(r1, r2);
        pdf = (float)(wi.z / kPiD);
        col = albedo(sc, sd) / (float)kPiD;
        return sd.frame.toWorld(wi);
    }
    V3 lwo = sd.frame.toLocal(sd.wo);
    if (m.type == RTX_BSDF_ORENNAYAR) {
        float thetaO = sphericalTheta(lwo), phiO = sphericalPhi(lwo);
        float r1 = sqrtf(rng.next());
        float thetaI = acosf(r1);
        float phiI = (float)(kPiD * 2 * rng.next());
        float sinTheta = sinf(thetaI);
        V3 wi = mk(sinTheta * cosf(phiI), sinTheta * sinf(phiI), r1);
        float demon = orenNayarTerm(m, thetaI, phiI, thetaO, phiO);
        pdf = (float)(fabsf(wi.z) / kPiD);
        col = (albedo(sc, sd) / (float)kPiD) * demon;
        return sd.frame.toWorld(wi);
    }
    if (m.type == RTX_BSDF_MIRROR) {
        V3 wi = mk(-lwo.x, -lwo.y, lwo.z);
        C3 F = fresnelConductor(fabsf(lwo.z), mkc(1.5f, 0.9f, 0.6f), mkc(7.0f, 6.0f, 4.7f));   // Aluminium, Materials.h:37,42
        col = albedo(sc, sd) * F / fabsf(wi.z);
        pdf = 1.0f;
        return sd.frame.toWorld(wi);
    }
    if (m.type == RTX_BSDF_CONDUCTOR) {
        col = albedo(sc, sd);
        V3 wi;
        const float alpha = m.alpha;
        if (alpha < kEps) {
            wi = mk(-lwo.x, -lwo.y, lwo.z);
            C3 F = fresnelConductor(fabsf(lwo.z), ldc(m.eta), ldc(m.k));
            col = col * F / fabsf(wi.z);
            pdf = 1.0f;
        } else {
            float r1 = rng.next();
            float r2 = rng.next();
            float cosTheta = sqrtf((1 - r1) / (r1 * (alpha * alpha - 1.0f) + 1.0f));
            float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
            float phi = (float)(2 * kPiD * r2);
            V3 wm = mk(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
            wi = normalize(-lwo + wm * dot(lwo, wm) * 2.0f);
            float ggx = Gggx(wi, lwo, alpha);
            float D = Dggx(wm, alpha);
            C3 F = fresnelConductor(fabsf(dot(wi, wm)), ldc(m.eta), ldc(m.k));
            col = col * F * ggx * D / (4 * fabsf(lwo.z) * fabsf(wi.z));
            pdf = D * fabsf(cosTheta / (4 * dot(lwo, wm)));
        }
        return sd.frame.toWorld(wi);
    }
    if (m.type == RTX_BSDF_GLASS) {
        col = albedo(sc, sd);
        float cosThetaI = fabsf(lwo.z);
        int transmitDir = lwo.z > 0.0f ? -1 : 1;
        float et = transmitDir < 0 ? m.eta_d : m.inv_eta;
        float F = fresnelDielectric(cosThetaI, et);
        V3 wi;
        if (F > rng.next()) { wi = mk(-lwo.x, -lwo.y, lwo.z); pdf = F; }
        else {
            float sin2ThetaT = et * et * (1.0f - cosThetaI * cosThetaI);
            float cosThetaT = sqrtf(1.0f - sin2ThetaT);
            wi = mk(-et * lwo.x, -et * lwo.y, transmitDir * cosThetaT);
            pdf = 1.0f - F;
            col = col * et * et;
        }
        col = col * pdf / fabsf(wi.z);
        return sd.frame.toWorld(wi);
    }
    // PlasticBSDF
    C3 c = albedo(sc, sd);
    V3 wi, wr = mk(-lwo.x, -lwo.y, lwo.z);
    const float e = m.phong_e;
    float F = fresnelDielectric(fabsf(lwo.z), m.eta_d);
    if (rng.next() < F) {
        float cosTheta = powf(rng.next(), 1.0f / (e + 1.0f));
        float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        float phi = (float)(2.0f * kPiD * rng.next());
        V3 wl = mk(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
        Frame f; f.fromVector(wr);
        wi = f.toWorld(wl);
    } else {
        RTX_DRAW_R2_THEN_R1(rng, r1, r2);
        wi = cosineSampleHemisphere(r1, r2);
    }
    float cosAlpha = fmaxf(0.0f, dot(wr, wi));
    float diffPdf = (float)((1.0f - F) * fabsf(wi.z) / kPiD);
    float specPdf = (float)(F * (e + 1.0f) * powf(cosAlpha, e) / (2.0f * kPiD));
    float diff = (float)((1.0f - F) / kPiD);
    float spec = (float)(F * (e + 2.0f) * powf(cosAlpha, e) / (2.0f * kPiD));
    pdf = diffPdf + specPdf;
    col = c * (diff + spec);
    return sd.frame.toWorld(wi);
}

// BSDF::evaluate, RTBase/Materials.h:155,209,300,409,460,525,628
RTX_DI C3 bsdf_evaluate(const DeviceScene& sc, const Shading& sd, V3 wi)
{
    const MaterialDev& m = *sd.mat;
    if (m.type == RTX_BSDF_DIFFUSE || m.type == RTX_BSDF_DIELECTRIC) return albedo(sc, sd) / (float)kPiD;
    if (m.type == RTX_BSDF_GLASS) return mkc(0, 0, 0);
    if (m.type == RTX_BSDF_ORENNAYAR) {
        V3 lo = sd.frame.toLocal(sd.wo), li = sd.frame.toLocal(wi);
        float demon = orenNayarTerm(m, sphericalTheta(li), sphericalPhi(li), sphericalTheta(lo), sphericalPhi(lo));
        return (albedo(sc, sd) / (float)kPiD) * demon;
    }
    if (m.type == RTX_BSDF_MIRROR || (m.type == RTX_BSDF_CONDUCTOR && m.alpha < kEps)) return albedo(sc, sd) / fabsf(sd.frame.toLocal(wi).z);
    V3 lwo = sd.frame.toLocal(sd.wo), lwi = sd.frame.toLocal(wi);
    if (m.type == RTX_BSDF_CONDUCTOR) {
        V3 wm = normalize(lwo + lwi);
        C3 F = fresnelConductor(fabsf(dot(lwi, wm)), ldc(m.eta), ldc(m.k));
        float D = Dggx(wm, m.alpha);
        float ggx = Gggx(lwi, lwo, m.alpha);
        return albedo(sc, sd) * F * D * ggx / (4.0f * fabsf(lwo.z) * fabsf(lwi.z));
    }
    C3 c = albedo(sc, sd);
    const float e = m.phong_e;
    float F = fresnelDielectric(fabsf(lwi.z), m.eta_d);
    V3 wr = mk(-lwo.x, -lwo.y, lwo.z);
    float cosAlpha = fmaxf(0.0f, dot(wr, lwi));
    float diff = (float)((1.0f - F) / kPiD);
    float spec = (float)(F * (e + 2.0f) * powf(cosAlpha, e) / (2.0f * kPiD));
    return c * (diff + spec);
}

// BSDF::PDF, RTBase/Materials.h:159,214,320,415,465,541,646
RTX_DI float bsdf_pdf(const Shading& sd, V3 wi)
{
    const MaterialDev& m = *sd.mat;
    if (m.type == RTX_BSDF_DIFFUSE) return (float)(sd.frame.toLocal(wi).z / kPiD);
    if (m.type == RTX_BSDF_DIELECTRIC || m.type == RTX_BSDF_ORENNAYAR) return cosineHemispherePDF(sd.frame.toLocal(wi));   // Materials.h:465-470, 541-546
    if (m.type == RTX_BSDF_MIRROR || m.type == RTX_BSDF_GLASS) return 0.0f;
    V3 lwo = sd.frame.toLocal(sd.wo), lwi = sd.frame.toLocal(wi);
    if (m.type == RTX_BSDF_CONDUCTOR) {
        if (m.alpha < kEps) return 0.0f;
        V3 wm = normalize(lwo + lwi);
        float D = Dggx(wm, m.alpha);
        return D * fabsf(wm.z / (4 * dot(lwo, wm)));
    }
    const float e = m.phong_e;
    V3 wr = mk(-lwo.x, -lwo.y, lwo.z);
    float cosTheta = fabsf(lwi.z);
    float cosAlpha = fmaxf(0.0f, dot(wr, lwi));
    float F = fresnelDielectric(cosTheta, m.eta_d);
    float diffPdf = (float)((1.0f - F) * cosTheta / kPiD);
    float specPdf = (float)(F * (e + 1.0f) * powf(cosAlpha, e) / (2.0f * kPiD));
    return specPdf + diffPdf;
}

// Scene::calculateShadingData, RTBase/Scene.h:183-213 (hit case)
RTX_DI void shading_data(const DeviceScene& sc, const rtx_hit& h, V3 o, V3 d, Shading& sd)
{
    const TriShade& t = sc.shade[h.id];
    sd.x = o + (d * h.t);
    V3 gn = ld3(t.gn);
    sd.gNormal = gn * t.gsign;                                                     // Triangle::gNormal, Geometry.h:158
    V3 n = ld3(t.n0) * h.alpha + ld3(t.n1) * h.beta + ld3(t.n2) * h.gamma;         // Geometry.h:128-134
    sd.sNormal = normalize(n);
    sd.tu = t.uv0[0] * h.alpha + t.uv1[0] * h.beta + t.uv2[0] * h.gamma;
    sd.tv = t.uv0[1] * h.alpha + t.uv1[1] * h.beta + t.uv2[1] * h.gamma;
    sd.mat = &sc.materials[t.material];
    {   // the material fields every path reads (BSDF type, texture, light flag, two-sidedness), fetched together: left
        // alone the compiler loads each where it is first used, one dependent round trip after the other
        const MaterialDev& m = *sd.mat;
        asm volatile("" :: "v"(m.type), "v"(m.tex_offset), "v"(m.tex_w), "v"(m.tex_h), "v"(m.is_light), "v"(m.two_sided));
    }
    sd.wo = -d;
    if (isTwoSided(*sd.mat)) {
        if (dot(sd.wo, sd.sNormal) < 0) sd.sNormal = -sd.sNormal;
        if (dot(sd.wo, sd.gNormal) < 0) sd.gNormal = -sd.gNormal;
    }
    sd.frame.fromVector(sd.sNormal);
    sd.t = h.t;
    sd.alb = tex_sample(sc.texels + 3 * (size_t)sd.mat->tex_offset, sd.mat->tex_w, sd.mat->tex_h, sd.tu, sd.tv);
}

// ---- environment / background: RTBase/Lights.h:86-135, 138-305, 307-413 -----------------
RTX_DI void dir_to_uv(V3 wi, float& u, float& v)                                 // Lights.h:346-349
{
    u = atan2f(wi.z, wi.x);
    u = (u < 0.0f) ? (float)(u + (2.0f * kPiD)) : u;
    u = (float)(u / (2.0f * kPiD));
    v = (float)(acosf(wi.y) / kPiD);
}
RTX_DI C3 background_evaluate(const DeviceScene& sc, V3 wi)
{
    if (sc.bg_kind == 1) return ldc(sc.bg_emission);                              // Lights.h:101
    float u, v; dir_to_uv(wi, u, v);
    return tex_sample(sc.env_texels, sc.env_w, sc.env_h, u, v);                   // Lights.h:344-352
}
RTX_DI float tab_pdf(const DeviceScene& sc, unsigned row, unsigned col)           // TabulatedDistribution::getPdf, Lights.h:232-241
{
    const float* rows = sc.env_cdf_rows; const float* cols = sc.env_cdf_cols + (size_t)row * sc.env_w;
    float mPDF = (row == 0) ? rows[row] : (rows[row] - rows[row - 1]);
    float cPDF = (col == 0) ? cols[col] : (cols[col] - cols[col - 1]);
    float pdf = (mPDF * cPDF) * (float)((unsigned)sc.env_w * (unsigned)sc.env_h);
    return pdf < kEps ? kEps : pdf;
}
RTX_DI float background_pdf(const DeviceScene& sc, V3 wi)
{
    if (sc.bg_kind == 1) return uniformSpherePDF();                               // Lights.h:105-108
    float u, v; dir_to_uv(wi, u, v);                                              // Lights.h:354-366
    unsigned row = (unsigned)(v * (float)(unsigned)(sc.env_h - 1));               // Lights.h:243-249
    unsigned col = (unsigned)(u * (float)(unsigned)(sc.env_w - 1));
    return tab_pdf(sc, row, col);
}
RTX_DI int cdf_search(const float* cdf, int n, float value)                       // TabulatedDistribution::binarySearch, Lights.h:210-230
{
    if (value <= cdf[0]) return 0;
    if (value >= cdf[n - 1]) return n - 1;
    int left = 0, right = n - 1;
    while (left < right) { int mid = left + (right - left) / 2; if (cdf[mid] < value) left = mid + 1; else right = mid; }
    return left;
}
// The same index through a guide table (guide[b] = first index whose value is >= b / n): the answer for `value`
// lies between the guide entries two buckets below and three above the bucket value * n falls in (the slack covers
// the rounding of that product and of b / n), so the search starts on a handful of entries instead of n.
RTX_DI int cdf_search_guided(const float* cdf, int n, const uint32_t* guide, float value)
{
    if (guide[0] == 0xFFFFFFFFu) return cdf_search(cdf, n, value);
    if (value <= cdf[0]) return 0;
    if (value >= cdf[n - 1]) return n - 1;
    int b = (int)(value * (float)n);
    b = b < 0 ? 0 : (b > n - 1 ? n - 1 : b);
    int left = (int)guide[b - 2 < 0 ? 0 : b - 2], right = (int)guide[b + 3 > n ? n : b + 3];
    while (left < right) { int mid = left + (right - left) / 2; if (cdf[mid] < value) left = mid + 1; else right = mid; }
    return left;
}
template <class Sampler>
RTX_DI V3 env_sample(const DeviceScene& sc, Sampler& rng, C3& emitted, float& pdf)    // Lights.h:252-277, 330-336
{
    unsigned row = cdf_search_guided(sc.env_cdf_rows, sc.env_h, sc.env_guide_rows, rng.next());
    unsigned col = cdf_search_guided(sc.env_cdf_cols + (size_t)row * sc.env_w, sc.env_w, sc.env_guide_cols + (size_t)row * (sc.env_w + 1), rng.next());
    pdf = tab_pdf(sc, row, col);
    float u = (float)col * sc.env_inv_w;
    float v = (float)row * sc.env_inv_h;
    float phi = (float)(u * 2.0f * kPiD);
    float theta = acosf(1.0f - 2.0f * v);
    float sinTheta = sinf(theta);
    V3 wi = mk(sinTheta * cosf(phi), sinTheta * sinf(phi), cosf(theta));
    emitted = background_evaluate(sc, wi);
    return wi;
}

// Scene::sampleLight, RTBase/Scene.h:143-148
RTX_DI const LightDev& sample_light(const DeviceScene& sc, Rng& rng, float& pmf)
{
    const float n = (float)sc.n_lights;
    pmf = 1 / n;
    unsigned i = (unsigned)fminf(rng.next() * n, n - 1.0f);
    return sc.lights[i];
}
// Triangle::sample, RTBase/Geometry.h:137-156
RTX_DI V3 light_tri_sample(const LightDev& l, Rng& rng, float& pdf)
{
    float r1 = rng.next();
    float r2 = rng.next();
    float rootr1 = sqrtf(r1);
    float alpha = 1 - rootr1;
    float beta = r2 * rootr1;
    float gamma = 1 - (alpha + beta);
    V3 p = ld3(l.v0) * alpha + ld3(l.v1) * beta + ld3(l.v2) * gamma;
    pdf = 1 / l.area;
    return p;
}
// Light::sample: AreaLight Lights.h:36-40, BackgroundColour :94-100, EnvironmentMap :338
RTX_DI V3 light_sample(const DeviceScene& sc, const LightDev& l, Rng& rng, C3& emitted, float& pdf)
{
    if (l.kind == 0) { emitted = ldc(l.emission); return light_tri_sample(l, rng, pdf); }
    if (l.kind == 1) { RTX_DRAW_R2_THEN_R1(rng, r1, r2); V3 wi = uniformSampleSphere(r1, r2); pdf = uniformSpherePDF(); emitted = ldc(l.emission); return wi; }
    return env_sample(sc, rng, emitted, pdf);
}
// Light::evaluate: Lights.h:41-48, 101, 344
RTX_DI C3 light_evaluate(const DeviceScene& sc, const LightDev& l, V3 wi)
{
    if (l.kind == 0) { if (dot(wi, ld3(l.gnormal)) < 0) return ldc(l.emission); return mkc(0, 0, 0); }
    return background_evaluate(sc, wi);
}
// Light::samplePositionFromLight: Lights.h:70-73, 121-128, 395-404
RTX_DI V3 light_sample_position(const DeviceScene& sc, const LightDev& l, Rng& rng, float& pdf)
{
    if (l.kind == 0) return light_tri_sample(l, rng, pdf);
    RTX_DRAW_R2_THEN_R1(rng, r1, r2);
    V3 p = uniformSampleSphere(r1, r2);
    p = p * sc.scene_radius;
    p = p + ld3(sc.scene_centre);
    if (l.kind == 1) pdf = (float)(4 * kPiD * sc.scene_radius * sc.scene_radius);
    else pdf = (float)(1.0f / (4 * kPiD * (sc.scene_radius * sc.scene_radius)));
    return p;
}
// Light::sampleDirectionFromLight: Lights.h:75-83, 129-134, 406-412
RTX_DI V3 light_sample_direction(const LightDev& l, Rng& rng, float& pdf)
{
    RTX_DRAW_R2_THEN_R1(rng, r1, r2);
    if (l.kind == 1) { V3 wi = uniformSampleSphere(r1, r2); pdf = uniformSpherePDF(); return wi; }
    V3 wi = cosineSampleHemisphere(r1, r2);
    pdf = cosineHemispherePDF(wi);
    if (l.kind == 2) return wi;
    Frame f; f.fromVector(ld3(l.gnormal));
    return f.toWorld(wi);
}

} // namespace rtx
