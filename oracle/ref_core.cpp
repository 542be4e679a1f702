// ref_core.cpp — C entry points over the REAL reference headers RTBase/Core.h and
// RTBase/Sampling.h, which compile on Linux exactly as they are (the rest of the
// reference needs Windows headers and is not buildable here).
//
// TEST INFRASTRUCTURE ONLY. Built by oracle/Makefile into oracle/_ref/libref_core.so from
// the sources where they lie under /root/reference (never copied into this repository);
// used by tests to check the oracle's restatement of the math and sampling layer.
#include "Core.h"
#include "Sampling.h"

#include <cstring>

extern "C" {

void ref_matrix_invert(const float* m, float* out) { Matrix a; memcpy(a.m, m, 64); Matrix r = a.invert(); memcpy(out, r.m, 64); }
void ref_look_at(const float* from, const float* to, const float* up, float* out)
{
    Matrix r = Matrix::lookAt(Vec3(from[0], from[1], from[2]), Vec3(to[0], to[1], to[2]), Vec3(up[0], up[1], up[2]));
    memcpy(out, r.m, 64);
}
void ref_perspective(float n, float f, float aspect, float fov, float* out) { Matrix r = Matrix::perspective(n, f, aspect, fov); memcpy(out, r.m, 64); }
void ref_mul_point(const float* m, const float* v, float* out) { Matrix a; memcpy(a.m, m, 64); Vec3 r = a.mulPoint(Vec3(v[0], v[1], v[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void ref_mul_vec(const float* m, const float* v, float* out) { Matrix a; memcpy(a.m, m, 64); Vec3 r = a.mulVec(Vec3(v[0], v[1], v[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void ref_mul_point_divide(const float* m, const float* v, float* out) { Matrix a; memcpy(a.m, m, 64); Vec3 r = a.mulPointAndPerspectiveDivide(Vec3(v[0], v[1], v[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void ref_normalize(const float* v, float* out) { Vec3 r = Vec3(v[0], v[1], v[2]).normalize(); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void ref_cross(const float* a, const float* b, float* out) { Vec3 r = Cross(Vec3(a[0], a[1], a[2]), Vec3(b[0], b[1], b[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
float ref_dot(const float* a, const float* b) { return Dot(Vec3(a[0], a[1], a[2]), Vec3(b[0], b[1], b[2])); }
float ref_lum(const float* c) { return Colour(c[0], c[1], c[2]).Lum(); }
// Frame::fromVector + toLocal + toWorld: out = u(3) v(3) w(3) local(3) world(3)
void ref_frame(const float* n, const float* vec, float* out)
{
    Frame f; f.fromVector(Vec3(n[0], n[1], n[2]));
    Vec3 l = f.toLocal(Vec3(vec[0], vec[1], vec[2])), w = f.toWorld(Vec3(vec[0], vec[1], vec[2]));
    const Vec3 v[5] = {f.u, f.v, f.w, l, w};
    for (int i = 0; i < 5; i++) { out[3 * i] = v[i].x; out[3 * i + 1] = v[i].y; out[3 * i + 2] = v[i].z; }
}
// kind 0: cosineSampleHemisphere, 1: uniformSampleSphere, 2: uniformSampleHemisphere; out = dir(3) pdf
void ref_warp(int kind, float r1, float r2, float* out)
{
    Vec3 d; float pdf;
    if (kind == 0) { d = SamplingDistributions::cosineSampleHemisphere(r1, r2); pdf = SamplingDistributions::cosineHemispherePDF(d); }
    else if (kind == 1) { d = SamplingDistributions::uniformSampleSphere(r1, r2); pdf = SamplingDistributions::uniformSpherePDF(d); }
    else { d = SamplingDistributions::uniformSampleHemisphere(r1, r2); pdf = SamplingDistributions::uniformHemispherePDF(d); }
    out[0] = d.x; out[1] = d.y; out[2] = d.z; out[3] = pdf;
}
void ref_mt_stream(unsigned seed, int n, float* out) { MTRandom r(seed); for (int i = 0; i < n; i++) out[i] = r.next(); }
// The call shape every BSDF and light of the reference uses, f(sampler->next(), sampler->next())
// (RTBase/Materials.h:150,454,610; RTBase/Lights.h:78,96,123,131,324,398,409), through the virtual Sampler
// interface on a real MTRandom: which draw lands in which parameter is the compiler's choice, and this
// build (g++, the oracle of record) fixes it. kind as in ref_warp; out = dir(3).
void ref_warp_from_sampler(int kind, unsigned seed, float* out)
{
    MTRandom r(seed); Sampler* sampler = &r; Vec3 d;
    if (kind == 0) d = SamplingDistributions::cosineSampleHemisphere(sampler->next(), sampler->next());
    else if (kind == 1) d = SamplingDistributions::uniformSampleSphere(sampler->next(), sampler->next());
    else d = SamplingDistributions::uniformSampleHemisphere(sampler->next(), sampler->next());
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
// SphericalCoordinates, RTBase/Core.h:557-572 (used by OrenNayarBSDF): out = sphericalToWorld(theta, phi)(3),
// sphericalTheta(wi), sphericalPhi(wi)
void ref_spherical(float theta, float phi, const float* wi, float* out)
{
    Vec3 w = SphericalCoordinates::sphericalToWorld(theta, phi);
    Vec3 v(wi[0], wi[1], wi[2]);
    out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = SphericalCoordinates::sphericalTheta(v); out[4] = SphericalCoordinates::sphericalPhi(v);
}

} // extern "C"
