// oracle_render.hpp — CPU restatement of the reference's shading, lights, film and
// integrators (second half of the oracle; see oracle.hpp for the status header).
//
// TEST INFRASTRUCTURE ONLY; parity unpinned (the reference ships no fixtures for this
// path and cannot be built in this image). Each function cites the reference lines it
// restates; operation order, double-precision M_PI sub-expressions, the Windows min/max
// macro semantics and the order of Sampler::next() calls are kept.
#pragma once

#include "oracle.hpp"

namespace orc {

// ---------------------------------------------------------------------------------------
// Texture — RTBase/Imaging.h:23-137 (load path stays on the host; texels arrive as floats)
// ---------------------------------------------------------------------------------------
struct Texture {
    std::vector<RGB> texels; int width = 1, height = 1;
    RGB sample(float tu, float tv) const                                             // Imaging.h:79-101
    {
        float u = std::max(0.0f, fabsf(tu)) * width;
        float v = std::max(0.0f, fabsf(tv)) * height;
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
        RGB s0 = texels[y * width + x];
        RGB s1 = texels[y * width + ((x + 1) % width)];
        RGB s2 = texels[((y + 1) % height) * width + x];
        RGB s3 = texels[((y + 1) % height) * width + ((x + 1) % width)];
        return (s0 * w0) + (s1 * w1) + (s2 * w2) + (s3 * w3);
    }
};

// ---------------------------------------------------------------------------------------
// Materials — RTBase/Materials.h
// ---------------------------------------------------------------------------------------
struct BSDF;
struct Shading {                                                  // ShadingData, Materials.h:11-31
    V3 x, wo, sNormal, gNormal; float tu = 0, tv = 0; Frame frame; const BSDF* bsdf = nullptr; float t = 0;
};

static inline float fresnelDielectric(float cosTheta, float eta)                     // Materials.h:47-65
{
    float sin2ThetaT = eta * eta * (1.0f - cosTheta * cosTheta);
    if (sin2ThetaT >= 1.0f) return 1.0f;
    float cosThetaT = sqrtf(1.0f - sin2ThetaT);
    float fParl = (cosTheta - eta * cosThetaT) / (cosTheta + eta * cosThetaT);
    float fPerp = (eta * cosTheta - cosThetaT) / (eta * cosTheta + cosThetaT);
    return (fParl * fParl + fPerp * fPerp) * 0.5f;
}
static inline RGB fresnelConductor(float cosTheta, RGB ior, RGB k)                   // Materials.h:67-84
{
    cosTheta = (cosTheta < 0.0f) ? 0.0f : (cosTheta > 1.0f) ? 1.0f : cosTheta;       // clamp, Core.h:17
    RGB eta2k2 = ior * ior + k * k;
    RGB cos2Theta = rgb(1.0f, 1.0f, 1.0f) * cosTheta * cosTheta;
    RGB sin2Theta = rgb(1.0f, 1.0f, 1.0f) * (1.0f - cosTheta * cosTheta);
    RGB fParl = (eta2k2 * cos2Theta - ior * 2 * cosTheta + sin2Theta) / (eta2k2 * cos2Theta + ior * 2 * cosTheta + sin2Theta);
    RGB fPerp = (eta2k2 - ior * 2 * cosTheta + cos2Theta) / (eta2k2 + ior * 2 * cosTheta + cos2Theta);
    return (fParl * fParl + fPerp * fPerp) * 0.5f;
}
static inline float lambdaGGX(V3 wi, float alpha)                                    // Materials.h:86-94
{
    float cosTheta = fabsf(wi.z);
    float cos2Theta = cosTheta * cosTheta;
    float tan2Theta = (1.0f - cos2Theta) / cos2Theta;
    return (sqrtf(1.0f + alpha * alpha * tan2Theta) - 1.0f) * 0.5f;
}
static inline float Gggx(V3 wi, V3 wo, float alpha)                                  // Materials.h:96-102
{
    float g1Wi = 1.0f / (1.0f + lambdaGGX(wi, alpha));
    float g1Wo = 1.0f / (1.0f + lambdaGGX(wo, alpha));
    return g1Wi * g1Wo;
}
static inline float Dggx(V3 h, float alpha)                                          // Materials.h:104-111
{
    float cos2Theta = h.z * h.z;
    float alpha2 = alpha * alpha;
    float denom = cos2Theta * (alpha2 - 1.0f) + 1.0f;
    return (float)(alpha2 / (kPi * denom * denom));
}

struct BSDF {                                                     // BSDF + subclasses, Materials.h:114-678, as a tagged record
    int type = RTX_BSDF_DIFFUSE; const Texture* albedo = nullptr; RGB emission;
    RGB eta, k; float alpha = 0;        // conductor (eta,k) / plastic (alpha)
    float intIOR = 1, extIOR = 1, etaD = 1, invEta = 1, e = 0;  // glass / plastic
    float onA = 0, onB = 0;             // Oren-Nayar, Materials.h:491
    bool layered = false;               // wrapped in a LayeredBSDF, Materials.h:680-724 (pass-through, two-sided)

    static const RGB alEta() { return rgb(1.5f, 0.9f, 0.6f); }   // Aluminium_Eta, Materials.h:37
    static const RGB alK() { return rgb(7.0f, 6.0f, 4.7f); }     // Aluminium_K, Materials.h:42

    void init(const rtx_material& m, const Texture* tex)
    {
        type = m.type & ~RTX_BSDF_LAYERED; layered = (m.type & RTX_BSDF_LAYERED) != 0; albedo = tex; emission = rgb(m.emission[0], m.emission[1], m.emission[2]);
        eta = rgb(m.eta[0], m.eta[1], m.eta[2]); k = rgb(m.k[0], m.k[1], m.k[2]);
        intIOR = m.int_ior; extIOR = m.ext_ior;
        if (type == RTX_BSDF_ORENNAYAR) {                                            // Materials.h:495-502
            const float sigma = m.roughness;
            onA = 1.0f - (sigma * sigma / (2.0f * ((sigma * sigma) + 0.33f)));
            onB = (0.45 * sigma * sigma) / (sigma * sigma + 0.09f);
        }
        if (type == RTX_BSDF_CONDUCTOR) alpha = 1.62142f * sqrtf(m.roughness);      // Materials.h:248
        if (type == RTX_BSDF_GLASS) { etaD = extIOR / intIOR; invEta = 1.0f / etaD; } // Materials.h:368-369
        if (type == RTX_BSDF_PLASTIC) {                                              // Materials.h:582-585
            alpha = 1.62142f * sqrtf(m.roughness);
            etaD = extIOR / intIOR;
            float a = std::max(alpha, 0.001f);
            e = (2.0f / (a * a)) - 2.0f;
        }
    }
    bool isLight() const { return lum(emission) > 0 ? true : false; }               // Materials.h:123
    RGB emit() const { return emission; }                                           // Materials.h:131
    bool isPureSpecular() const { return type == RTX_BSDF_MIRROR || type == RTX_BSDF_GLASS; } // :165,219,335,420,664
    bool isTwoSided() const { return layered || (type != RTX_BSDF_GLASS && type != RTX_BSDF_DIELECTRIC); } // :169,223,340,425,475,553,669,716
    static float sphericalTheta(V3 w) { return acosf(w.z); }                        // Core.h:564-567
    static float sphericalPhi(V3 w) { float p = atan2f(w.y, w.x); return (p < 0.0f) ? p + (2.0f * kPi) : p; }   // Core.h:568-572
    float orenNayarTerm(float thetaI, float phiI, float thetaO, float phiO) const   // Materials.h:517,536
    {
        return (onA + onB * std::max(cosf(phiI - phiO), 0.0f) * sinf(std::max(thetaI, thetaO) * tanf(std::min(thetaI, thetaO))));
    }

    V3 sample(const Shading& sd, Sampler* s, RGB& col, float& pdf) const
    {
        switch (type) {
        case RTX_BSDF_DIELECTRIC:                                                   // Materials.h:451-459 (same body)
        case RTX_BSDF_DIFFUSE: {                                                    // Materials.h:148-154
            ORC_DRAW_R2_THEN_R1(s, r1, r2);
            V3 wi = cosineSampleHemisphere(r1, r2);
            pdf = (float)(wi.z / kPi);
            col = albedo->sample(sd.tu, sd.tv) / (float)kPi;
            return sd.frame.toWorld(wi);
        }
        case RTX_BSDF_ORENNAYAR: {                                                  // Materials.h:504-523
            V3 localWo = sd.frame.toLocal(sd.wo);
            float thetaO = sphericalTheta(localWo);
            float phiO = sphericalPhi(localWo);
            float r1 = sqrtf(s->next());
            float thetaI = acosf(r1);
            float phiI = kPi * 2 * s->next();
            float sinTheta = sinf(thetaI);
            V3 wi = v3(sinTheta * cosf(phiI), sinTheta * sinf(phiI), r1);
            float demon = orenNayarTerm(thetaI, phiI, thetaO, phiO);
            pdf = (float)(fabsf(wi.z) / kPi);
            col = (albedo->sample(sd.tu, sd.tv) / (float)kPi) * demon;
            return sd.frame.toWorld(wi);
        }
        case RTX_BSDF_MIRROR: {                                                     // Materials.h:193-208
            V3 lwo = sd.frame.toLocal(sd.wo);
            V3 wi = v3(-lwo.x, -lwo.y, lwo.z);
            RGB F = fresnelConductor(fabsf(lwo.z), alEta(), alK());
            col = albedo->sample(sd.tu, sd.tv) * F / fabsf(wi.z);
            pdf = 1.0f;
            return sd.frame.toWorld(wi);
        }
        case RTX_BSDF_CONDUCTOR: {                                                  // Materials.h:251-298
            V3 lwo = sd.frame.toLocal(sd.wo);
            col = albedo->sample(sd.tu, sd.tv);
            V3 wi;
            if (alpha < kEps) {
                wi = v3(-lwo.x, -lwo.y, lwo.z);
                RGB F = fresnelConductor(fabsf(lwo.z), eta, k);
                col = col * F / fabsf(wi.z);
                pdf = 1.0f;
            } else {
                float r1 = s->next();
                float r2 = s->next();
                float cosTheta = sqrtf((1 - r1) / (r1 * (alpha * alpha - 1.0f) + 1.0f));
                float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
                float phi = (float)(2 * kPi * r2);
                V3 wm = v3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
                wi = normalize(-lwo + wm * dot(lwo, wm) * 2.0f);
                float ggx = Gggx(wi, lwo, alpha);
                float D = Dggx(wm, alpha);
                RGB F = fresnelConductor(fabsf(dot(wi, wm)), eta, k);
                col = col * F * ggx * D / (4 * fabsf(lwo.z) * fabsf(wi.z));
                pdf = D * fabsf(cosTheta / (4 * dot(lwo, wm)));
            }
            return sd.frame.toWorld(wi);
        }
        case RTX_BSDF_GLASS: {                                                      // Materials.h:372-407
            V3 lwo = sd.frame.toLocal(sd.wo);
            col = albedo->sample(sd.tu, sd.tv);
            float cosThetaI = fabsf(lwo.z);
            int transmitDir = lwo.z > 0.0f ? -1 : 1;
            float et = transmitDir < 0 ? etaD : invEta;
            float F = fresnelDielectric(cosThetaI, et);
            V3 wi;
            if (F > s->next()) { wi = v3(-lwo.x, -lwo.y, lwo.z); pdf = F; }
            else {
                float sin2ThetaT = et * et * (1.0f - cosThetaI * cosThetaI);
                float cosThetaT = sqrtf(1.0f - sin2ThetaT);
                wi = v3(-et * lwo.x, -et * lwo.y, transmitDir * cosThetaT);
                pdf = 1.0f - F;
                col = col * et * et;
            }
            col = col * pdf / fabsf(wi.z);
            return sd.frame.toWorld(wi);
        }
        default: {                                                                  // PlasticBSDF, Materials.h:588-626
            V3 lwo = sd.frame.toLocal(sd.wo);
            RGB c = albedo->sample(sd.tu, sd.tv);
            V3 wi, wr = v3(-lwo.x, -lwo.y, lwo.z);
            float F = fresnelDielectric(std::abs(lwo.z), etaD);
            if (s->next() < F) {
                float cosTheta = powf(s->next(), 1.0f / (e + 1.0f));
                float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
                float phi = (float)(2.0f * kPi * s->next());
                V3 wl = v3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
                Frame f; f.fromVector(wr);
                wi = f.toWorld(wl);
            } else {
                ORC_DRAW_R2_THEN_R1(s, r1, r2);
                wi = cosineSampleHemisphere(r1, r2);
            }
            float cosAlpha = std::max(0.0f, dot(wr, wi));
            float diffPdf = (float)((1.0f - F) * fabsf(wi.z) / kPi);
            float specPdf = (float)(F * (e + 1.0f) * powf(cosAlpha, e) / (2.0f * kPi));
            float diff = (float)((1.0f - F) / kPi);
            float spec = (float)(F * (e + 2.0f) * powf(cosAlpha, e) / (2.0f * kPi));
            pdf = diffPdf + specPdf;
            col = c * (diff + spec);
            return sd.frame.toWorld(wi);
        }
        }
    }
    RGB evaluate(const Shading& sd, V3 wi) const
    {
        switch (type) {
        case RTX_BSDF_DIELECTRIC:                                                                      // Materials.h:460-464
        case RTX_BSDF_DIFFUSE: return albedo->sample(sd.tu, sd.tv) / (float)kPi;                      // Materials.h:155-158
        case RTX_BSDF_ORENNAYAR: {                                                                     // Materials.h:525-539
            V3 localWo = sd.frame.toLocal(sd.wo), localWi = sd.frame.toLocal(wi);
            float thetaO = sphericalTheta(localWo), phiO = sphericalPhi(localWo);
            float thetaI = sphericalTheta(localWi), phiI = sphericalPhi(localWi);
            float demon = orenNayarTerm(thetaI, phiI, thetaO, phiO);
            return (albedo->sample(sd.tu, sd.tv) / (float)kPi) * demon;
        }
        case RTX_BSDF_MIRROR: return albedo->sample(sd.tu, sd.tv) / fabsf(sd.frame.toLocal(wi).z);     // Materials.h:209-213
        case RTX_BSDF_CONDUCTOR: {                                                                     // Materials.h:300-318
            if (alpha < kEps) return albedo->sample(sd.tu, sd.tv) / fabsf(sd.frame.toLocal(wi).z);
            V3 lwo = sd.frame.toLocal(sd.wo), lwi = sd.frame.toLocal(wi);
            V3 wm = normalize(lwo + lwi);
            RGB F = fresnelConductor(fabsf(dot(lwi, wm)), eta, k);
            float D = Dggx(wm, alpha);
            float ggx = Gggx(lwi, lwo, alpha);
            return albedo->sample(sd.tu, sd.tv) * F * D * ggx / (4.0f * fabsf(lwo.z) * fabsf(lwi.z));
        }
        case RTX_BSDF_GLASS: return rgb(0, 0, 0);                                                      // Materials.h:409-413
        default: {                                                                                     // Materials.h:628-644
            V3 lwo = sd.frame.toLocal(sd.wo), lwi = sd.frame.toLocal(wi);
            RGB c = albedo->sample(sd.tu, sd.tv);
            float F = fresnelDielectric(std::abs(lwi.z), etaD);
            V3 wr = v3(-lwo.x, -lwo.y, lwo.z);
            float cosAlpha = std::max(0.0f, dot(wr, lwi));
            float diff = (float)((1.0f - F) / kPi);
            float spec = (float)(F * (e + 2.0f) * powf(cosAlpha, e) / (2.0f * kPi));
            return c * (diff + spec);
        }
        }
    }
    float PDF(const Shading& sd, V3 wi) const
    {
        switch (type) {
        case RTX_BSDF_DIFFUSE: return (float)(sd.frame.toLocal(wi).z / kPi);                          // Materials.h:159-164
        case RTX_BSDF_DIELECTRIC:                                                                      // Materials.h:465-470
        case RTX_BSDF_ORENNAYAR: return cosineHemispherePDF(sd.frame.toLocal(wi));                    // Materials.h:541-546
        case RTX_BSDF_MIRROR: return 0.0f;                                                             // Materials.h:214-218
        case RTX_BSDF_CONDUCTOR: {                                                                     // Materials.h:320-333
            if (alpha < kEps) return 0.0f;
            V3 lwo = sd.frame.toLocal(sd.wo), lwi = sd.frame.toLocal(wi);
            V3 wm = normalize(lwo + lwi);
            float D = Dggx(wm, alpha);
            return D * fabsf(wm.z / (4 * dot(lwo, wm)));
        }
        case RTX_BSDF_GLASS: return 0.0f;                                                              // Materials.h:415-418
        default: {                                                                                     // Materials.h:646-662
            V3 lwo = sd.frame.toLocal(sd.wo), lwi = sd.frame.toLocal(wi);
            V3 wr = v3(-lwo.x, -lwo.y, lwo.z);
            float cosTheta = fabsf(lwi.z);
            float cosAlpha = std::max(0.0f, dot(wr, lwi));
            float F = fresnelDielectric(cosTheta, etaD);
            float diffPdf = (float)((1.0f - F) * cosTheta / kPi);
            float specPdf = (float)(F * (e + 1.0f) * powf(cosAlpha, e) / (2.0f * kPi));
            return specPdf + diffPdf;
        }
        }
    }
};

// ---------------------------------------------------------------------------------------
// Lights — RTBase/Lights.h
// ---------------------------------------------------------------------------------------
struct TabDist {                                                  // TabulatedDistribution, Lights.h:138-305
    unsigned width = 0, height = 0; float invWidth = 0, invHeight = 0, totalLum = 0, avgLum = 0;
    std::vector<float> lumv, cdfRows; std::vector<std::vector<float>> cdfCols;
    void init(const Texture& txt)                                                    // Lights.h:161-208
    {
        width = txt.width; height = txt.height;
        invWidth = 1.0f / (float)(width); invHeight = 1.0f / (float)(height);
        lumv.assign(width * height, 0.f); cdfRows.assign(height, 0.f); cdfCols.assign(height, std::vector<float>(width));
        totalLum = 0.0f;
        for (unsigned y = 0; y < height; y++) {
            float sinTheta = sinf((float)(kPi * ((float)y + 0.05f) / (float)(height - 1)));
            float rowSum = 0.0f;
            for (unsigned x = 0; x < width; x++) {
                unsigned index = y * width + x;
                lumv[index] = lum(txt.texels[index]) * sinTheta;
                cdfCols[y][x] = lumv[index];
                rowSum += lumv[index];
            }
            cdfRows[y] = rowSum;
            totalLum += rowSum;
        }
        avgLum = totalLum * invHeight * invWidth;
        float accumCol = 0.0f, accumRow = 0.0f;
        for (unsigned y = 0; y < height; y++) {
            accumCol = 0.0f;
            for (unsigned x = 0; x < width; x++) { accumCol += cdfCols[y][x] / cdfRows[y]; cdfCols[y][x] = accumCol; }
            accumRow += cdfRows[y] / totalLum;
            cdfRows[y] = accumRow;
        }
    }
    static int binarySearch(const std::vector<float>& cdf, float value)              // Lights.h:210-230
    {
        const int n = (int)cdf.size();
        if (value <= cdf[0]) return 0;
        if (value >= cdf[n - 1]) return n - 1;
        int left = 0, right = n - 1;
        while (left < right) { int mid = left + (right - left) / 2; if (cdf[mid] < value) left = mid + 1; else right = mid; }
        return left;
    }
    float getPdf(unsigned row, unsigned col) const                                   // Lights.h:232-241
    {
        float mPDF = (row == 0) ? cdfRows[row] : (cdfRows[row] - cdfRows[row - 1]);
        float cPDF = (col == 0) ? cdfCols[row][col] : (cdfCols[row][col] - cdfCols[row][col - 1]);
        float pdf = (mPDF * cPDF) * (width * height);
        return pdf < kEps ? kEps : pdf;
    }
    float getPdf(float u, float v) const                                             // Lights.h:243-249
    {
        unsigned row = (unsigned)(v * (height - 1));
        unsigned col = (unsigned)(u * (width - 1));
        return getPdf(row, col);
    }
    V3 sample(Sampler* s, float& pdf) const                                          // Lights.h:252-277
    {
        unsigned row = binarySearch(cdfRows, s->next());
        unsigned col = binarySearch(cdfCols[row], s->next());
        pdf = getPdf(row, col);
        float u = (float)col * invWidth;
        float v = (float)row * invHeight;
        float phi = (float)(u * 2.0f * kPi);
        float theta = acosf(1.0f - 2.0f * v);
        float sinTheta = sinf(theta);
        return v3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosf(theta));
    }
};

struct Light {                                                    // Light + AreaLight + BackgroundColour + EnvironmentMap, Lights.h:17-413
    enum Kind { AREA, CONSTANT, ENVMAP } kind = AREA;
    const Tri* triangle = nullptr; RGB emission;                  // AREA (Lights.h:33-34) / CONSTANT (Lights.h:89)
    const Texture* env = nullptr; TabDist tab;                    // ENVMAP (Lights.h:310-311)
    V3 sceneCentre; float sceneRadius = 0;                        // use<SceneBounds>(), Lights.h:10-15

    bool isArea() const { return kind == AREA; }
    RGB evaluate(V3 wi) const
    {
        if (kind == AREA) { if (dot(wi, triangle->gNormal()) < 0) return emission; return rgb(0, 0, 0); }   // Lights.h:41-48
        if (kind == CONSTANT) return emission;                                                            // Lights.h:101
        float u = atan2f(wi.z, wi.x);                                                                     // Lights.h:344-352
        u = (u < 0.0f) ? (float)(u + (2.0f * kPi)) : u;
        u = (float)(u / (2.0f * kPi));
        float v = (float)(acosf(wi.y) / kPi);
        return env->sample(u, v);
    }
    V3 sample(Sampler* s, RGB& emitted, float& pdf) const
    {
        if (kind == AREA) { emitted = emission; return triangle->sample(s, pdf); }                        // Lights.h:36-40
        if (kind == CONSTANT) {                                                                           // Lights.h:94-100
            ORC_DRAW_R2_THEN_R1(s, r1, r2);
            V3 wi = uniformSampleSphere(r1, r2); pdf = uniformSpherePDF(); emitted = emission; return wi;
        }
        V3 wi = tab.sample(s, pdf); emitted = evaluate(wi); return wi;                                    // Lights.h:330-336
    }
    float PDF(V3 wi) const
    {
        if (kind == AREA) return 1.0f / triangle->area;                                                   // Lights.h:50-53
        if (kind == CONSTANT) return uniformSpherePDF();                                                  // Lights.h:105-108
        float u = atan2f(wi.z, wi.x);                                                                     // Lights.h:354-366
        u = (u < 0.0f) ? (float)(u + (2.0f * kPi)) : u;
        u = (float)(u / (2.0f * kPi));
        float v = (float)(acosf(wi.y) / kPi);
        return tab.getPdf(u, v);
    }
    V3 normal(V3 wi) const { return kind == AREA ? triangle->gNormal() : -wi; }                            // Lights.h:60,113,373
    float totalIntegratedPower() const
    {
        if (kind == AREA) return (triangle->area * lum(emission));                                        // Lights.h:65-68
        if (kind == CONSTANT) return (float)(lum(emission) * 4.0f * kPi);                                 // Lights.h:117-120
        return (float)(tab.avgLum * 4.0f * kPi);                                                          // Lights.h:378-380
    }
    V3 samplePositionFromLight(Sampler* s, float& pdf) const
    {
        if (kind == AREA) return triangle->sample(s, pdf);                                                // Lights.h:70-73
        ORC_DRAW_R2_THEN_R1(s, r1, r2);                                                                   // Lights.h:121-128, 395-404
        V3 p = uniformSampleSphere(r1, r2);
        p = p * sceneRadius;
        p = p + sceneCentre;
        if (kind == CONSTANT) pdf = (float)(4 * kPi * sceneRadius * sceneRadius);
        else pdf = (float)(1.0f / (4 * kPi * (sceneRadius * sceneRadius)));
        return p;
    }
    V3 sampleDirectionFromLight(Sampler* s, float& pdf) const
    {
        if (kind == CONSTANT) { ORC_DRAW_R2_THEN_R1(s, r1, r2); V3 wi = uniformSampleSphere(r1, r2); pdf = uniformSpherePDF(); return wi; } // Lights.h:129-134
        ORC_DRAW_R2_THEN_R1(s, r1, r2);
        V3 wi = cosineSampleHemisphere(r1, r2);
        pdf = cosineHemispherePDF(wi);
        if (kind == ENVMAP) return wi;                                                                    // Lights.h:406-412 (not rotated)
        Frame f; f.fromVector(triangle->gNormal());                                                       // Lights.h:75-83
        return f.toWorld(wi);
    }
};

// ---------------------------------------------------------------------------------------
// Camera + Scene — RTBase/Scene.h
// ---------------------------------------------------------------------------------------
struct Camera {                                                   // Scene.h:10-76
    M4 projectionMatrix, inverseProjectionMatrix, camera, cameraToView; float width = 0, height = 0; V3 origin, viewDirection; float Afilm = 0;
    void init(const M4& P, int w, int h)                                             // Scene.h:22-32
    {
        projectionMatrix = P; inverseProjectionMatrix = invert(P); width = (float)w; height = (float)h;
        float Wlens = (2.0f / P.m[5]);
        float aspect = P.m[0] / P.m[5];
        float Hlens = Wlens * aspect;
        Afilm = Wlens * Hlens;
    }
    void updateView(const M4& V)                                                     // Scene.h:33-41
    {
        camera = V; cameraToView = invert(V);
        origin = mulPoint(camera, v3(0, 0, 0));
        viewDirection = mulPointDivide(inverseProjectionMatrix, v3(0, 0, 1));
        viewDirection = mulVec(camera, viewDirection);
        viewDirection = normalize(viewDirection);
    }
    // loadScene's camera block, SceneLoader.h:342-360
    void lookAtSetup(V3 from, V3 to, V3 up, float fov, int w, int h, int flipX)
    {
        M4 P = perspective(0.001f, 10000.0f, (float)w / (float)h, fov);
        M4 V = invert(lookAt(from, to, up));
        if (flipX == 1) P.m[0] = -P.m[0];
        init(P, w, h); updateView(V);
    }
    Ray generateRay(float x, float y) const                                          // Scene.h:49-59
    {
        float xprime = x / width;
        float yprime = 1.0f - (y / height);
        xprime = (xprime * 2.0f) - 1.0f;
        yprime = (yprime * 2.0f) - 1.0f;
        V3 dir = v3(xprime, yprime, 1.0f);
        dir = mulPoint(inverseProjectionMatrix, dir);
        dir = mulVec(camera, dir);
        return Ray(origin, normalize(dir));
    }
    bool projectOntoCamera(V3 p, float& x, float& y) const                           // Scene.h:61-75
    {
        V3 pview = mulPoint(cameraToView, p);
        V3 pproj = mulPointDivide(projectionMatrix, pview);
        x = (pproj.x + 1.0f) * 0.5f;
        y = (pproj.y + 1.0f) * 0.5f;
        if (x < 0 || x > 1.0f || y < 0 || y > 1.0f) return false;
        x = x * width;
        y = 1.0f - y;
        y = y * height;
        return true;
    }
    void load(const rtx_camera& c)
    {
        std::memcpy(projectionMatrix.m, c.proj, 64); std::memcpy(inverseProjectionMatrix.m, c.inv_proj, 64);
        std::memcpy(camera.m, c.camera, 64); std::memcpy(cameraToView.m, c.camera_to_view, 64);
        origin = v3(c.origin[0], c.origin[1], c.origin[2]); viewDirection = v3(c.view_dir[0], c.view_dir[1], c.view_dir[2]);
        Afilm = c.afilm; width = (float)c.width; height = (float)c.height;
    }
    void store(rtx_camera& c) const
    {
        std::memcpy(c.proj, projectionMatrix.m, 64); std::memcpy(c.inv_proj, inverseProjectionMatrix.m, 64);
        std::memcpy(c.camera, camera.m, 64); std::memcpy(c.camera_to_view, cameraToView.m, 64);
        c.origin[0] = origin.x; c.origin[1] = origin.y; c.origin[2] = origin.z;
        c.view_dir[0] = viewDirection.x; c.view_dir[1] = viewDirection.y; c.view_dir[2] = viewDirection.z;
        c.afilm = Afilm; c.width = (int)width; c.height = (int)height;
    }
};

struct Scene {                                                    // Scene.h:78-214
    std::vector<Tri> triangles; std::vector<BSDF> materials; std::vector<Texture> textures; Texture envTex;
    std::vector<Light> lights; Light background; bool backgroundInLights = false; BVH bvh; Camera camera; Box bounds;

    // Scene::init + Scene::build (Scene.h:94-110,151-169) and the scene-bounds sphere of SceneLoader.h:387-388
    void load(const rtx_scene_desc& d)
    {
        textures.resize(d.n_textures);
        for (unsigned i = 0; i < d.n_textures; i++) {
            Texture& t = textures[i]; t.width = d.textures[i].width; t.height = d.textures[i].height;
            t.texels.resize((size_t)t.width * t.height);
            for (size_t j = 0; j < t.texels.size(); j++) t.texels[j] = rgb(d.textures[i].rgb[3 * j], d.textures[i].rgb[3 * j + 1], d.textures[i].rgb[3 * j + 2]);
        }
        materials.resize(d.n_materials);
        for (unsigned i = 0; i < d.n_materials; i++) materials[i].init(d.materials[i], &textures[d.materials[i].texture]);
        triangles.resize(d.n_triangles);
        for (uint64_t i = 0; i < d.n_triangles; i++) {
            Vtx v[3];
            for (int k = 0; k < 3; k++) {
                v[k].p = v3(d.positions[i * 9 + k * 3], d.positions[i * 9 + k * 3 + 1], d.positions[i * 9 + k * 3 + 2]);
                v[k].normal = v3(d.normals[i * 9 + k * 3], d.normals[i * 9 + k * 3 + 1], d.normals[i * 9 + k * 3 + 2]);
                v[k].u = d.uvs[i * 6 + k * 2]; v[k].v = d.uvs[i * 6 + k * 2 + 1];
            }
            triangles[i].init(v[0], v[1], v[2], d.material_index[i]);
            bounds.extend(triangles[i].vertices[0].p); bounds.extend(triangles[i].vertices[1].p); bounds.extend(triangles[i].vertices[2].p);
        }
        if (d.env_rgb) {
            envTex.width = d.env_width; envTex.height = d.env_height; envTex.texels.resize((size_t)d.env_width * d.env_height);
            for (size_t j = 0; j < envTex.texels.size(); j++) envTex.texels[j] = rgb(d.env_rgb[3 * j], d.env_rgb[3 * j + 1], d.env_rgb[3 * j + 2]);
            background.kind = Light::ENVMAP; background.env = &envTex; background.tab.init(envTex);
        } else { background.kind = Light::CONSTANT; background.emission = rgb(d.background[0], d.background[1], d.background[2]); }
        backgroundInLights = background.totalIntegratedPower() > 0;                      // Scene.h:165-168
        bvh.build(triangles, bounds);                                                     // Scene.h:97
        V3 centre = (bounds.max + bounds.min) * 0.5f;                                     // SceneLoader.h:387-388
        float radius = length(bounds.max - centre);
        background.sceneCentre = centre; background.sceneRadius = radius;
        lights.clear();
        if (backgroundInLights) lights.push_back(background);
        for (size_t i = 0; i < triangles.size(); i++) {                                   // Scene.h:101-110
            if (materials[triangles[i].materialIndex].isLight()) {
                Light l; l.kind = Light::AREA; l.triangle = &triangles[i]; l.emission = materials[triangles[i].materialIndex].emission;
                lights.push_back(l);
            }
        }
    }
    Hit traverse(const Ray& r) const { return bvh.traverse(r); }                         // Scene.h:137-141
    const Light* sampleLight(Sampler* s, float& pmf) const                               // Scene.h:143-148
    {
        pmf = 1 / (float)lights.size();
        unsigned i = (unsigned)std::min(s->next() * lights.size(), lights.size() - 1.0f);
        return &lights[i];
    }
    bool visible(V3 p1, V3 p2) const                                                     // Scene.h:170-178
    {
        Ray ray; V3 dir = p2 - p1;
        float maxT = length(dir) - (2.0f * kEps);
        dir = normalize(dir);
        ray.init(p1 + (dir * kEps), dir);
        return bvh.traverseVisible(ray, maxT);
    }
    Shading calculateShadingData(const Hit& hit, const Ray& ray) const                   // Scene.h:183-213
    {
        Shading sd;
        if (hit.t < FLT_MAX) {
            const Tri& tr = triangles[hit.ID];
            sd.x = ray.at(hit.t);
            sd.gNormal = tr.gNormal();
            tr.interpolateAttributes(hit.alpha, hit.beta, hit.gamma, sd.sNormal, sd.tu, sd.tv);
            sd.bsdf = &materials[tr.materialIndex];
            sd.wo = -ray.dir;
            if (sd.bsdf->isTwoSided()) {
                if (dot(sd.wo, sd.sNormal) < 0) sd.sNormal = -sd.sNormal;
                if (dot(sd.wo, sd.gNormal) < 0) sd.gNormal = -sd.gNormal;
            }
            sd.frame.fromVector(sd.sNormal);
            sd.t = hit.t;
        } else { sd.wo = -ray.dir; sd.t = hit.t; }
        return sd;
    }
};

// ---------------------------------------------------------------------------------------
// Film — RTBase/Imaging.h:139-222 (filters), 234-418 (Film)
// ---------------------------------------------------------------------------------------
struct Film {
    std::vector<RGB> film; unsigned width = 0, height = 0; int SPP = 0; int filter = RTX_FT_BOX;
    void init(int w, int h, int f) { width = w; height = h; filter = f; film.assign((size_t)w * h, RGB()); SPP = 0; } // Imaging.h:388-395
    void clear() { std::fill(film.begin(), film.end(), RGB()); SPP = 0; }                // Imaging.h:397-401
    int size() const { return filter == RTX_FT_BOX ? 0 : 2; }                            // Imaging.h:157,180,218
    static float gaussian(float d) { const float alpha = 2.5f; const float t2 = std::exp(-alpha * (2 * 2)); return std::exp(-alpha * (d * d)) - t2; } // Imaging.h:163-173
    static float mitchell(float d)                                                       // Imaging.h:186-211
    {
        const float B = 1.0f / 3.0f, C = 1.0f / 3.0f;
        const float a1 = (1.0f / 6.0f) * (12 - 9 * B - 6 * C), a2 = (-18 + 12 * B + 6 * C), a3 = (6 - 2 * B);
        const float b1 = (1.0f / 6.0f) * (-B - 6 * C), b2 = (6 * B + 30 * C), b3 = (-12 * B - 48 * C), b4 = (8 * B + 24 * C);
        d = (float)fabs(d);
        if (d >= 2) return 0;
        float ds = d * d, dc = ds * d;
        if (d >= 0 && d < 1) return a1 * dc + a2 * ds + a3;
        return b1 * dc + b2 * ds + b3 * d + b4;
    }
    float weight(float x, float y) const
    {
        if (filter == RTX_FT_BOX) return (fabsf(x) <= 1.0f && fabsf(y) <= 1.0f) ? 1.0f : 0.0f;   // Imaging.h:149-156
        if (filter == RTX_FT_GAUSSIAN) return gaussian(x) * gaussian(y);                          // Imaging.h:175-178
        return mitchell(x) * mitchell(y);                                                         // Imaging.h:213-216
    }
    void splat(float x, float y, RGB L)                                                  // Imaging.h:318-340
    {
        float w[25]; unsigned idx[25]; unsigned used = 0; float total = 0; int sz = size();
        for (int i = -sz; i <= sz; i++) for (int j = -sz; j <= sz; j++) {
            int px = (int)x + j, py = (int)y + i;
            if (px >= 0 && px < (int)width && py >= 0 && py < (int)height) {
                idx[used] = (py * width) + px; w[used] = weight(px - x, py - y); total += w[used]; used++;
            }
        }
        for (unsigned i = 0; i < used; i++) film[idx[i]] = film[idx[i]] + (L * w[i] / total);
    }
};

// ---------------------------------------------------------------------------------------
// Integrators + driver — RTBase/Renderer.h
// ---------------------------------------------------------------------------------------
struct VPL { V3 x, n; RGB Le; };                                  // VPL, Renderer.h:19-23 (only x, sNormal and Le are read back)

struct Renderer {
    const Scene* scene = nullptr; Film film; rtx_settings st{}; unsigned totalTiles = 0, totalXTiles = 0; static constexpr unsigned tileSize = 16; // Renderer.h:36-38
    std::vector<unsigned> tileSamples; std::vector<VPL> vpls; unsigned numThreads = 1; bool counterRng = true;
    std::vector<MTSampler> mt;   // per-thread MT19937, seeds 48271*(i+1) (Renderer.h:102-104)

    void init(const Scene* sc, const rtx_settings& s, unsigned threads, bool useCounterRng)     // Renderer.h:69-115
    {
        scene = sc; st = s; counterRng = useCounterRng;
        film.init((int)sc->camera.width, (int)sc->camera.height, s.filter);
        st.adaptive_sampling = st.adaptive_sampling && st.algorithm == RTX_AL_PATH_TRACE;
        if (!st.adaptive_sampling) st.init_spp = st.total_spp;
        numThreads = threads < 1 ? 1 : threads;
        mt.clear(); for (unsigned i = 0; i < numThreads; i++) mt.emplace_back(48271u * (i + 1));
        totalXTiles = (film.width + tileSize - 1) / tileSize;
        unsigned totalYTiles = (film.height + tileSize - 1) / tileSize;
        totalTiles = totalXTiles * totalYTiles;
        tileSamples.assign(totalTiles, 1);
    }
    void clear() { film.clear(); }                                                       // Renderer.h:117-120

    // ---- path tracing: Renderer.h:417-532 ----
    RGB computeDirect(const Shading& sd, Sampler* s, float misPdf) const                 // Renderer.h:417-474
    {
        if (sd.bsdf->isPureSpecular()) return rgb(0, 0, 0);
        if (scene->lights.empty()) return rgb(0, 0, 0);   // the reference indexes an empty vector here; defined as "no direct light"
        float pmf; const Light* light = scene->sampleLight(s, pmf);
        float pdf; RGB emitted; V3 p = light->sample(s, emitted, pdf);
        pdf *= pmf;
        if (light->isArea()) {
            V3 wi = p - sd.x; float lsq = lengthSq(wi); wi = normalize(wi);
            float g = (wmax(dot(wi, sd.sNormal), 0.0f) * wmax(-dot(wi, light->normal(wi)), 0.0f)) / lsq;
            if (g > 0) if (scene->visible(sd.x, p)) return sd.bsdf->evaluate(sd, wi) * emitted * g / pdf;
        } else {
            V3 wi = normalize(p);
            float g = wmax(dot(wi, sd.sNormal), 0.0f);
            if (g > 0) if (scene->visible(sd.x, sd.x + (p * 1000.0f))) {
                float weight = st.use_mis ? pdf / (pdf + misPdf) : 1.0f;
                return sd.bsdf->evaluate(sd, wi) * emitted * g * weight / pdf;
            }
        }
        return rgb(0, 0, 0);
    }
    RGB pathTrace(Ray& r, RGB& T, int depth, Sampler* s, bool canHitLight, float misPdf = 0) const // Renderer.h:476-526
    {
        Hit hit = scene->traverse(r);
        Shading sd = scene->calculateShadingData(hit, r);
        if (sd.t < FLT_MAX) {
            if (sd.bsdf->isLight()) return canHitLight ? T * sd.bsdf->emit() : rgb(0, 0, 0);
            RGB direct = T * computeDirect(sd, s, misPdf);
            if (depth >= (int)st.max_bounces) return direct;
            float rr = wmin(lum(T), 0.9f);
            if (rr < s->next()) return direct;
            T = T / rr;
            RGB bs; float pdf; V3 wi = sd.bsdf->sample(sd, s, bs, pdf);
            T = T * bs * fabsf(dot(wi, sd.sNormal)) / pdf;
            r.init(sd.x + (wi * kEps), wi);
            return direct + pathTrace(r, T, depth + 1, s, sd.bsdf->isPureSpecular(), pdf);
        }
        if (depth == 0) return scene->background.evaluate(r.dir);
        float pdf = scene->background.PDF(r.dir);
        float weight = st.use_mis ? pdf / (pdf + misPdf) : 1.0f;
        return scene->background.evaluate(r.dir) * T * weight / pdf;
    }
    RGB pathTrace(Ray& r, Sampler* s) const { RGB T = rgb(1, 1, 1); return pathTrace(r, T, 0, s, st.can_hit_light != 0); } // Renderer.h:528-532

    // ---- debug / AOV integrators: Renderer.h:536-575 ----
    RGB direct(Ray& r, Sampler* s) const
    {
        Hit hit = scene->traverse(r); Shading sd = scene->calculateShadingData(hit, r);
        if (sd.t < FLT_MAX) { if (sd.bsdf->isLight()) return sd.bsdf->emit(); return computeDirect(sd, s, 0); }
        return scene->background.evaluate(r.dir);
    }
    RGB albedo(Ray& r) const
    {
        Hit hit = scene->traverse(r); Shading sd = scene->calculateShadingData(hit, r);
        if (sd.t < FLT_MAX) { if (sd.bsdf->isLight()) return sd.bsdf->emit(); return sd.bsdf->evaluate(sd, v3(0, 1, 0)); }
        return scene->background.evaluate(r.dir);
    }
    RGB viewNormals(Ray& r) const
    {
        Hit hit = scene->traverse(r);
        if (hit.t < FLT_MAX) { Shading sd = scene->calculateShadingData(hit, r); return rgb(fabsf(sd.sNormal.x), fabsf(sd.sNormal.y), fabsf(sd.sNormal.z)); }
        return rgb(0, 0, 0);
    }

    // RayTracer::createAOV, Renderer.h:786-820: colour = film / spp (per-tile spp under adaptive
    // sampling), albedo and view normals through the pixel centres; W*H*3 floats each
    void createAOV(float* color, float* albedoOut, float* normalOut) const
    {
        for (unsigned y = 0; y < film.height; y++) {
            int sppY = (y / tileSize) * totalXTiles;
            for (unsigned x = 0; x < film.width; x++) {
                unsigned index = y * film.width + x;
                int sppIndex = sppY + x / tileSize;
                int spp = st.adaptive_sampling && (int)st.init_spp < film.SPP ? (int)wmin(tileSamples[sppIndex], (unsigned)film.SPP) : film.SPP;
                RGB col = film.film[index] / (float)spp;
                color[index * 3] = col.r; color[index * 3 + 1] = col.g; color[index * 3 + 2] = col.b;
                float px = x + 0.5f, py = y + 0.5f;
                Ray ray = scene->camera.generateRay(px, py);
                col = albedo(ray);
                albedoOut[index * 3] = col.r; albedoOut[index * 3 + 1] = col.g; albedoOut[index * 3 + 2] = col.b;
                col = viewNormals(ray);
                normalOut[index * 3] = col.r; normalOut[index * 3 + 1] = col.g; normalOut[index * 3 + 2] = col.b;
            }
        }
    }

    // ---- light tracing: Renderer.h:308-411 ----
    void connectToCamera(V3 p, V3 n, RGB col)                                            // Renderer.h:308-334
    {
        float x, y;
        if (scene->camera.projectOntoCamera(p, x, y)) {
            V3 toCam = (scene->camera.origin - p); float lsq = lengthSq(toCam); toCam = normalize(toCam);
            float cosS = dot(toCam, n), cosC = dot(toCam, scene->camera.viewDirection);
            float g = wmax(cosS, 0.0f) * wmax(-cosC, 0.0f) / lsq;
            if (g > 0) if (scene->visible(p, scene->camera.origin)) {
                float c2 = (cosC * cosC);
                float we = 1.0f / (c2 * c2 * scene->camera.Afilm);
                film.splat(x, y, col * we * g);
            }
        }
    }
    void lightTracePath(Ray& r, RGB T, RGB Le, Sampler* s, int depth = 0)                // Renderer.h:336-377
    {
        if (depth >= (int)st.max_bounces) return;
        Hit hit = scene->traverse(r); Shading sd = scene->calculateShadingData(hit, r);
        if (sd.t < FLT_MAX) {
            if (sd.bsdf->isLight() || sd.bsdf->isPureSpecular()) return;
            V3 wi = normalize(scene->camera.origin - sd.x);
            RGB col = T * sd.bsdf->evaluate(sd, wi) * Le;
            connectToCamera(sd.x, sd.sNormal, col);
            float rr = wmin(lum(T), 0.9f);
            if (rr < s->next()) return;
            T = T / rr;
            RGB bs; float pdf; wi = sd.bsdf->sample(sd, s, bs, pdf);
            T = T * bs * fabsf(dot(wi, sd.sNormal)) / pdf;
            r.init(sd.x + (wi * kEps), wi);
            lightTracePath(r, T, Le, s, depth + 1);
        }
    }
    void lightTrace(Sampler* s)                                                          // Renderer.h:379-411
    {
        if (scene->lights.empty()) return;
        float pmf; const Light* light = scene->sampleLight(s, pmf);
        float pdfPos, pdfDir;
        V3 p = light->samplePositionFromLight(s, pdfPos);
        V3 wi = light->sampleDirectionFromLight(s, pdfDir);
        V3 nLight = light->normal(wi);
        float cosTheta = dot(wi, nLight);
        RGB Le = light->evaluate(-wi) / (pmf * pdfPos);
        if (st.can_hit_light) connectToCamera(p, nLight, Le);
        if (light->isArea()) Le = Le * cosTheta;
        Ray ray(p, wi);
        lightTracePath(ray, rgb(1, 1, 1), Le, s);
    }

    // ---- instant radiosity: Renderer.h:124-302 ----
    void VPLTracePath(Ray& r, RGB T, RGB Le, Sampler* s, std::vector<VPL>& out, int depth = 0) const // Renderer.h:124-168
    {
        if (depth >= (int)st.max_bounces) return;
        Hit hit = scene->traverse(r); Shading sd = scene->calculateShadingData(hit, r);
        if (sd.t < FLT_MAX) {
            if (sd.bsdf->isLight() || sd.bsdf->isPureSpecular()) return;
            RGB bs; float pdf; V3 wi = sd.bsdf->sample(sd, s, bs, pdf);
            T = T * bs * fabsf(dot(wi, sd.sNormal)) / pdf;
            VPL v; v.x = sd.x; v.n = sd.sNormal; v.Le = T * Le; out.push_back(v);
            float rr = wmin(lum(T), 0.9f);
            if (rr < s->next()) return;
            T = T / rr;
            r.init(sd.x + (wi * kEps), wi);
            VPLTracePath(r, T, Le, s, out, depth + 1);
        }
    }
    void traceVPLPath(Sampler* s, std::vector<VPL>& out) const                            // body of the loop in traceVPLs, Renderer.h:176-207
    {
        if (scene->lights.empty()) return;
        float pmf; const Light* light = scene->sampleLight(s, pmf);
        float pdfPos, pdfDir;
        V3 p = light->samplePositionFromLight(s, pdfPos);
        V3 wi = light->sampleDirectionFromLight(s, pdfDir);
        V3 nLight = light->normal(wi);
        float cosTheta = dot(wi, nLight);
        RGB Le = light->evaluate(-wi) * cosTheta / (pmf * pdfPos);
        VPL v; v.x = p; v.n = nLight; v.Le = Le; out.push_back(v);
        Ray ray(p, wi);
        VPLTracePath(ray, rgb(1, 1, 1), Le, s, out);
    }
    void radiosityVplPass()                                                               // Renderer.h:280-302 (+ traceVPLs :170-208)
    {
        // numThreads * vplRaysPerTile light paths per pass; lists concatenated in path order
        unsigned paths = st.vpl_rays_per_tile * st.num_threads;
        vpls.clear();
        for (unsigned pidx = 0; pidx < paths; pidx++) {
            if (counterRng) { CounterSampler cs(st.seed, RTX_STREAM_VPL, pidx, globalSample()); traceVPLPath(&cs, vpls); }
            else traceVPLPath(&mt[pidx % numThreads], vpls);
        }
    }
    RGB radiosityComputeDirect(const Shading& sd) const                                   // Renderer.h:210-245
    {
        if (sd.bsdf->isPureSpecular()) return rgb(0, 0, 0);
        if (sd.bsdf->isLight()) return st.can_hit_light ? sd.bsdf->emit() : rgb(0, 0, 0);
        RGB acc = rgb(0, 0, 0);
        unsigned total = (unsigned)vpls.size();
        for (unsigned i = 0; i < total; i++) {
            const VPL& v = vpls[i];
            V3 wi = v.x - sd.x; float lsq = lengthSq(wi); wi = normalize(wi);
            float g = (wmax(dot(wi, sd.sNormal), 0.0f) * wmax(-dot(wi, v.n), 0.0f)) / lsq;
            if (g > 0 && scene->visible(sd.x, v.x)) acc = acc + sd.bsdf->evaluate(sd, wi) * v.Le * g;
        }
        return acc / (float)total;
    }
    RGB radiosityLightPass(Ray r) const                                                   // Renderer.h:262-278
    {
        Hit hit = scene->traverse(r); Shading sd = scene->calculateShadingData(hit, r);
        return sd.t < FLT_MAX ? radiosityComputeDirect(sd) : scene->background.evaluate(r.dir);
    }

    // ---- adaptive sampling: Renderer.h:643-671 (+ Film::getLums, Imaging.h:376-385: x outer, y inner) ----
    void calculateTileSamples()
    {
        for (unsigned i = 0; i < totalTiles; i++) {
            unsigned sx = (i % totalXTiles) * tileSize, sy = (i / totalXTiles) * tileSize;
            unsigned ex = wmin(sx + tileSize, film.width), ey = wmin(sy + tileSize, film.height);
            std::vector<float> lums;
            for (unsigned x = sx; x < ex; x++) for (unsigned y = sy; y < ey; y++) lums.push_back(lum(film.film[y * film.width + x]));
            float total = 0.0f; for (float l : lums) total += l;
            float mean = lums.empty() ? 0.0f : total / lums.size();
            float variance = 0.0f; for (float l : lums) variance += (l - mean) * (l - mean);
            variance = lums.empty() ? 0.0f : variance / lums.size();
            float q = variance / (variance + mean * mean + kEps);
            float weight = (q < kEps) ? kEps : (q > 1.0f) ? 1.0f : q;
            tileSamples[i] = (unsigned)(film.SPP + (st.total_spp - film.SPP) * weight);
        }
    }

    // ---- one pixel of processTile, Renderer.h:695-747 ----
    void pixel(unsigned x, unsigned y, Sampler* s)
    {
        if (st.draw_mode != RTX_DM_ALGORITHM) {
            float px = x + 0.5f, py = y + 0.5f; Ray ray = scene->camera.generateRay(px, py); RGB col;
            switch (st.draw_mode) { case RTX_DM_ALBEDO: col = albedo(ray); break; case RTX_DM_NORMALS: col = viewNormals(ray); break; default: col = direct(ray, s); break; }
            film.splat(px, py, col); return;
        }
        if (st.algorithm == RTX_AL_LIGHT_TRACE) { lightTrace(s); return; }
        float px = x + s->next();
        float py = y + s->next();
        Ray ray = scene->camera.generateRay(px, py);
        RGB col = st.algorithm == RTX_AL_PATH_TRACE ? pathTrace(ray, s) : radiosityLightPass(ray);
        film.splat(px, py, col);
    }
    // shard_mode 2 (samples split over ranks): this rank's k-th pass renders global sample rank+1 + (k-1)*world
    uint32_t globalSample() const
    { return (st.shard_mode == 2 && st.shard_world > 1) ? (uint32_t)(film.SPP - 1) * (uint32_t)st.shard_world + (uint32_t)st.shard_rank + 1u : (uint32_t)film.SPP; }
    bool tileSkipped(unsigned i) const                                                     // Renderer.h:682-687, on the global sample count
    { const int cur = (int)globalSample(); return cur > (int)st.init_spp && cur > (int)tileSamples[i] && st.algorithm != RTX_AL_LIGHT_TRACE; }
    bool sharded(unsigned tile) const { return st.shard_mode == 1 && st.shard_world > 1 && (int)(tile % (unsigned)st.shard_world) != st.shard_rank; }

    // ---- renderMT, Renderer.h:753-782 (tile queue on an atomic counter, Renderer.h:677-751) ----
    void renderPass()
    {
        film.SPP++;                                                                       // Film::incrementSPP, Imaging.h:403
        if (st.algorithm == RTX_AL_INSTANT_RADIOSITY && st.draw_mode == RTX_DM_ALGORITHM) radiosityVplPass();
        std::atomic<unsigned> counter{0};
        auto work = [&](unsigned id) {
            unsigned i; CounterSampler cs;
            while ((i = counter.fetch_add(1)) < totalTiles) {
                if (tileSkipped(i) || sharded(i)) continue;
                unsigned sx = (i % totalXTiles) * tileSize, sy = (i / totalXTiles) * tileSize;
                unsigned ex = wmin(sx + tileSize, film.width), ey = wmin(sy + tileSize, film.height);
                for (unsigned y = sy; y < ey; y++) for (unsigned x = sx; x < ex; x++) {
                    if (counterRng) {
                        uint32_t stream = st.algorithm == RTX_AL_LIGHT_TRACE && st.draw_mode == RTX_DM_ALGORITHM ? RTX_STREAM_LIGHT : RTX_STREAM_CAMERA;
                        cs.reset(st.seed, stream, y * film.width + x, globalSample());
                        pixel(x, y, &cs);
                    } else pixel(x, y, &mt[id]);
                }
            }
        };
        // light tracing splats anywhere on the film (a data race in the reference); keep it single threaded here
        unsigned nt = (st.algorithm == RTX_AL_LIGHT_TRACE && st.draw_mode == RTX_DM_ALGORITHM) || film.filter != RTX_FT_BOX ? 1 : numThreads;
        if (nt <= 1) work(0);
        else { std::vector<std::thread> th; for (unsigned i = 0; i < nt; i++) th.emplace_back(work, i); for (auto& t : th) t.join(); }
        if (film.SPP == (int)st.init_spp && st.algorithm != RTX_AL_LIGHT_TRACE && st.shard_mode != 2) calculateTileSamples();
    }
};

} // namespace orc
