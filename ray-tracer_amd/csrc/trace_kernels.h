// trace_kernels.h — ray-query kernels: closest hit, visibility, primary-ray generation.
#pragma once

#include "context.h"
#include "traverse.h"

namespace rtx {

constexpr float kEpsilon = 0.001f;    // EPSILON, RTBase/Geometry.h:54

__device__ __forceinline__ uint32_t global_thread() { return blockIdx.x * kBlock + threadIdx.x; }

// Camera::generateRay, RTBase/Scene.h:49-59
__device__ __forceinline__ void camera_ray(const CameraDev& c, float x, float y, float& dx, float& dy, float& dz)
{
    float xprime = x / c.width;
    float yprime = 1.0f - (y / c.height);
    xprime = (xprime * 2.0f) - 1.0f;
    yprime = (yprime * 2.0f) - 1.0f;
    const float* m = c.inv_proj;
    float px = (xprime * m[0] + yprime * m[1] + 1.0f * m[2]) + m[3];
    float py = (xprime * m[4] + yprime * m[5] + 1.0f * m[6]) + m[7];
    float pz = (xprime * m[8] + yprime * m[9] + 1.0f * m[10]) + m[11];
    const float* v = c.camera;
    float wx = (px * v[0] + py * v[1] + pz * v[2]);
    float wy = (px * v[4] + py * v[5] + pz * v[6]);
    float wz = (px * v[8] + py * v[9] + pz * v[10]);
    float l = 1.0f / sqrtf((wx * wx) + (wy * wy) + (wz * wz));
    dx = wx * l; dy = wy * l; dz = wz * l;
}

// Closest hit for rays[queue ? queue[i] : i], i < n. One ray per lane, grid-stride.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_closest(DeviceScene sc, const rtx_ray* __restrict__ rays, const uint32_t* __restrict__ queue,
                                                    uint32_t n, rtx_hit* __restrict__ hits, uint2* spill, uint32_t spill_stride,
                                                    unsigned long long* counters)
{
    __shared__ uint2 lds[kBlock * kLdsStack];
    LaneStack<kBlock, kLdsStack> stk; stk.lds = lds; stk.spill = spill + global_thread(); stk.stride = spill_stride; stk.sp = 0;
    TravCount cnt{0, 0};
    for (uint32_t i = global_thread(); i < n; i += gridDim.x * kBlock) {
        const uint32_t idx = queue ? queue[i] : i;
        const rtx_ray r = rays[idx];
        HitRec best;
        traverse<false, COUNT, kBlock, kLdsStack>(sc, stk, r.o[0], r.o[1], r.o[2], r.d[0], r.d[1], r.d[2], 0.0f, best, &cnt);
        rtx_hit h;
        if (best.pos != 0xFFFFFFFFu) {
            h.id = __float_as_uint(sc.tris[best.pos].q0.w);
            h.t = best.t; h.alpha = best.u; h.beta = best.v; h.gamma = 1.0f - (best.u + best.v);   // Geometry.h:537-541
        } else { h.id = 0xFFFFFFFFu; h.t = FLT_MAX; h.alpha = 0; h.beta = 0; h.gamma = 0; }
        hits[idx] = h;
    }
    if (COUNT) { atomicAdd(&counters[0], (unsigned long long)cnt.nodes); atomicAdd(&counters[1], (unsigned long long)cnt.tris); }
}

// Closest hit for the live paths of a wavefront stage: rays[queue[i]], i < *n_dev (the
// queue length stays on the device).
__global__ __launch_bounds__(kBlock) void k_closest_q(DeviceScene sc, const rtx_ray* __restrict__ rays, const uint32_t* __restrict__ queue,
                                                      const uint32_t* __restrict__ n_dev, rtx_hit* __restrict__ hits, uint2* spill, uint32_t spill_stride)
{
    __shared__ uint2 lds[kBlock * kLdsStack];
    LaneStack<kBlock, kLdsStack> stk; stk.lds = lds; stk.spill = spill + global_thread(); stk.stride = spill_stride; stk.sp = 0;
    const uint32_t n = *n_dev;
    for (uint32_t i = global_thread(); i < n; i += gridDim.x * kBlock) {
        const uint32_t idx = queue[i];
        const rtx_ray r = rays[idx];
        HitRec best;
        traverse<false, false, kBlock, kLdsStack>(sc, stk, r.o[0], r.o[1], r.o[2], r.d[0], r.d[1], r.d[2], 0.0f, best, nullptr);
        rtx_hit h;
        if (best.pos != 0xFFFFFFFFu) {
            h.id = __float_as_uint(sc.tris[best.pos].q0.w);
            h.t = best.t; h.alpha = best.u; h.beta = best.v; h.gamma = 1.0f - (best.u + best.v);
        } else { h.id = 0xFFFFFFFFu; h.t = FLT_MAX; h.alpha = 0; h.beta = 0; h.gamma = 0; }
        hits[idx] = h;
    }
}

// Scene::visible, RTBase/Scene.h:170-178, for one segment
template <int BLOCK, int LDS_N>
__device__ __forceinline__ bool segment_visible(const DeviceScene& sc, LaneStack<BLOCK, LDS_N>& stk,
                                                float p1x, float p1y, float p1z, float p2x, float p2y, float p2z)
{
    float dx = p2x - p1x, dy = p2y - p1y, dz = p2z - p1z;
    float len2 = (dx * dx) + (dy * dy) + (dz * dz);
    float maxT = sqrtf(len2) - (2.0f * kEpsilon);
    float l = 1.0f / sqrtf(len2);
    dx = dx * l; dy = dy * l; dz = dz * l;
    float ox = p1x + (dx * kEpsilon), oy = p1y + (dy * kEpsilon), oz = p1z + (dz * kEpsilon);
    HitRec best;
    bool root = traverse<true, false, BLOCK, LDS_N>(sc, stk, ox, oy, oz, dx, dy, dz, maxT, best, nullptr);
    return root && best.pos == 0xFFFFFFFFu;
}

__global__ __launch_bounds__(kBlock) void k_visible(DeviceScene sc, const float* __restrict__ p1p2, uint32_t n,
                                                    uint8_t* __restrict__ visible, uint2* spill, uint32_t spill_stride)
{
    __shared__ uint2 lds[kBlock * kLdsStack];
    LaneStack<kBlock, kLdsStack> stk; stk.lds = lds; stk.spill = spill + global_thread(); stk.stride = spill_stride; stk.sp = 0;
    for (uint32_t i = global_thread(); i < n; i += gridDim.x * kBlock) {
        const float* p = p1p2 + (size_t)i * 6;
        visible[i] = segment_visible(sc, stk, p[0], p[1], p[2], p[3], p[4], p[5]) ? 1 : 0;
    }
}

// Position of pixel (x, y) in the 8x8-tile ray order: whole tile rows above, whole tiles to
// the left in this tile row, then row-major inside the (possibly clipped) tile.
__device__ __forceinline__ uint32_t tile8_index(uint32_t x, uint32_t y, uint32_t W, uint32_t H)
{
    uint32_t tx = x >> 3, ty = y >> 3, lx = x & 7, ly = y & 7;
    uint32_t wt = min(8u, W - tx * 8), ht = min(8u, H - ty * 8);
    return ty * 8 * W + tx * 8 * ht + ly * wt + lx;
}

__global__ __launch_bounds__(kBlock) void k_primary_rays(CameraDev cam, rtx_ray* __restrict__ rays, uint32_t* __restrict__ pixel)
{
    uint32_t i = global_thread();
    uint32_t W = cam.w, H = cam.h;
    if (i >= W * H) return;
    uint32_t x = i % W, y = i / W;
    float dx, dy, dz;
    camera_ray(cam, x + 0.5f, y + 0.5f, dx, dy, dz);
    uint32_t q = tile8_index(x, y, W, H);
    rtx_ray r; r.o[0] = cam.origin[0]; r.o[1] = cam.origin[1]; r.o[2] = cam.origin[2]; r.d[0] = dx; r.d[1] = dy; r.d[2] = dz;
    rays[q] = r;
    if (pixel) pixel[q] = i;
}

} // namespace rtx
