// debug_tables.h — function-level parity tables: the device's shading functions evaluated on tables of given inputs,
// so that a test can compare them one by one with the CPU restatement of the reference's functions instead of only
// through whole films. Test support behind the C ABI (rtx_debug_*), no part of a render.
#pragma once

#include "context.h"
#include "shade.h"

namespace rtx {

// Sampler::next() over a list of given numbers, cyclic (the tables' stand-in for the random generator)
template <int N>
struct ListSampler {
    const float* u; int i;
    RTX_DI float next() { return u[i++ % N]; }
};

// Scene::calculateShadingData, RTBase/Scene.h:183-213, per (ray, hit):
// out[i] = x(3) wo(3) sNormal(3) gNormal(3) tu tv frame.u(3) frame.v(3) frame.w(3) material index; zeros on a miss
__global__ void k_debug_shading_data(DeviceScene sc, const rtx_ray* __restrict__ rays, const rtx_hit* __restrict__ hits, uint32_t n, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* q = out + 24 * (size_t)i;
    for (int k = 0; k < 24; k++) q[k] = 0.0f;
    const rtx_hit h = hits[i];
    if (!(h.t < FLT_MAX)) return;
    Shading sd; shading_data(sc, h, ld3(rays[i].o), ld3(rays[i].d), sd);
    const V3 v[7] = {sd.x, sd.wo, sd.sNormal, sd.gNormal, sd.frame.u, sd.frame.v, sd.frame.w};
    int k = 0;
    for (int j = 0; j < 4; j++) { q[k++] = v[j].x; q[k++] = v[j].y; q[k++] = v[j].z; }
    q[k++] = sd.tu; q[k++] = sd.tv;
    for (int j = 4; j < 7; j++) { q[k++] = v[j].x; q[k++] = v[j].y; q[k++] = v[j].z; }
    q[k++] = (float)(sd.mat - sc.materials);
}

// BSDF::sample / evaluate / PDF, RTBase/Materials.h:139-724, of one material on a table of inputs:
// in[i]  = normal(3) wo(3) tu tv wi_eval(3) u0 u1 u2     (u* feed Sampler::next() in call order, cyclic)
// out[i] = wi(3) colour(3) pdf  evaluate(3) PDF
__global__ void k_debug_bsdf_table(DeviceScene sc, uint32_t material, const float* __restrict__ in, uint32_t n, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = in + 14 * (size_t)i; float* q = out + 11 * (size_t)i;
    Shading sd;
    sd.sNormal = ld3(p); sd.gNormal = sd.sNormal; sd.wo = ld3(p + 3); sd.tu = p[6]; sd.tv = p[7]; sd.x = mk(0, 0, 0); sd.t = 0.0f;
    sd.frame.fromVector(sd.sNormal);
    sd.mat = &sc.materials[material];
    sd.alb = tex_sample(sc.texels + 3 * (size_t)sd.mat->tex_offset, sd.mat->tex_w, sd.mat->tex_h, sd.tu, sd.tv);
    ListSampler<3> ls{p + 11, 0};
    C3 c = mkc(0, 0, 0); float pdf = 0.0f;
    const V3 wi = bsdf_sample(sc, sd, ls, c, pdf);
    q[0] = wi.x; q[1] = wi.y; q[2] = wi.z; q[3] = c.r; q[4] = c.g; q[5] = c.b; q[6] = pdf;
    const V3 we = ld3(p + 8);
    const C3 e = bsdf_evaluate(sc, sd, we);
    q[7] = e.r; q[8] = e.g; q[9] = e.b; q[10] = bsdf_pdf(sd, we);
}

// EnvironmentMap::sample on given (u0, u1) pairs (RTBase/Lights.h:252-277,338): out[i] = wi(3) pdf Le(3);
// Light::evaluate / PDF of the background on given directions (Lights.h:101-108, 344-366): out[i] = Le(3) pdf
__global__ void k_debug_env_sample(DeviceScene sc, const float* __restrict__ u01, uint32_t n, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ListSampler<2> ls{u01 + 2 * (size_t)i, 0};
    C3 e; float pdf;
    const V3 wi = env_sample(sc, ls, e, pdf);
    float* q = out + 7 * (size_t)i;
    q[0] = wi.x; q[1] = wi.y; q[2] = wi.z; q[3] = pdf; q[4] = e.r; q[5] = e.g; q[6] = e.b;
}
__global__ void k_debug_env_eval(DeviceScene sc, const float* __restrict__ dirs, uint32_t n, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 d = ld3(dirs + 3 * (size_t)i);
    const C3 e = background_evaluate(sc, d);
    float* q = out + 4 * (size_t)i;
    q[0] = e.r; q[1] = e.g; q[2] = e.b; q[3] = background_pdf(sc, d);
}

// host side: copy the table in, run `launch(in_dev, out_dev)`, copy the result out (blocking)
template <class Launch>
inline int debug_table(rtx_ctx* c, const void* in, size_t in_bytes, const void* in2, size_t in2_bytes, void* out, size_t out_bytes, Launch launch)
{
    const size_t second = (in_bytes + 15) & ~(size_t)15;
    HIP_TRY(c->b_io_a.ensure(second + in2_bytes + 64));
    HIP_TRY(c->b_io_b.ensure(out_bytes + 64));
    char* d_in = (char*)c->b_io_a.p; char* d_in2 = d_in + second;
    HIP_TRY(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, c->stream));
    if (in2_bytes) HIP_TRY(hipMemcpyAsync(d_in2, in2, in2_bytes, hipMemcpyHostToDevice, c->stream));
    launch(d_in, d_in2, c->b_io_b.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->b_io_b.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTX_OK;
}

} // namespace rtx
