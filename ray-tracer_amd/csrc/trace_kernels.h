// trace_kernels.h — ray-query kernels: closest hit, visibility, primary-ray generation.
#pragma once

#include "context.h"
#include "traverse.h"
#include "traverse_dyn.h"
#include "traverse_packet.h"

namespace rtx {

constexpr float kEpsilon = 0.001f;    // EPSILON, RTBase/Geometry.h:54
constexpr int kRefill = 20;           // idle lanes that trigger a fetch of new rays
constexpr int kTriUnroll = 4;         // leaf triangles whose loads are issued together
constexpr int kLeafLanes = 28;        // the wave stops descending and tests triangles once this many lanes hold a leaf
// the closest-hit kernel on its own (no visibility rays mixed in) does 2-3 % better with later refills and longer descents
constexpr int kRefillClosest = 24, kLeafLanesClosest = 36;
constexpr int kShare = 1;             // idle lanes share the pending subtrees of busy ones once the work counters are dry

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

__device__ __forceinline__ rtx_hit make_hit(const DeviceScene& sc, const HitRec& best)
{
    rtx_hit h;
    if (best.pos != 0xFFFFFFFFu) {
        h.id = sc.tri_ids[best.pos];
        h.t = best.t; h.alpha = best.u; h.beta = best.v; h.gamma = 1.0f - (best.u + best.v);   // Geometry.h:537-541
    } else { h.id = 0xFFFFFFFFu; h.t = FLT_MAX; h.alpha = 0; h.beta = 0; h.gamma = 0; }
    return h;
}

#ifdef RTX_TUNING
// Reference schedule: one ray per lane, grid-stride, each wave finishes its 64 rays before
// taking the next 64. Kept as the A/B baseline of the dynamic schedule (RTX_VARIANT=0, tuning build only).
__global__ __launch_bounds__(kBlock) void k_closest_static(DeviceScene sc, const rtx_ray* __restrict__ rays, uint32_t n, rtx_hit* __restrict__ hits,
                                                           uint2* spill, uint32_t spill_stride)
{
    __shared__ unsigned long long lds[kBlock * kLdsStack];
    LaneStack<kBlock, kLdsStack> stk; stk.init(lds, spill, spill_stride);
    for (uint32_t i = global_thread(); i < n; i += gridDim.x * kBlock) {
        const rtx_ray r = rays[i];
        HitRec best;
        traverse<false, false, kBlock, kLdsStack>(sc, stk, r.o[0], r.o[1], r.o[2], r.d[0], r.d[1], r.d[2], 0.0f, best, nullptr);
        hits[i] = make_hit(sc, best);
    }
}
#endif

// ---- closest hit: Scene::traverse, RTBase/Scene.h:137 -> BVHTree::traverse, Geometry.h:506-566 ----
struct ClosestPolicy {
    const DeviceScene& sc; const rtx_ray* rays; const uint32_t* queue; rtx_hit* hits;
    __device__ __forceinline__ uint32_t index(uint32_t k) const { return queue ? queue[k] : k; }
    __device__ __forceinline__ bool fetch(uint32_t k, float o[3], float d[3], float& maxT) const
    {
        const rtx_ray r = rays[index(k)];
        o[0] = r.o[0]; o[1] = r.o[1]; o[2] = r.o[2]; d[0] = r.d[0]; d[1] = r.d[1]; d[2] = r.d[2]; maxT = 0.0f;
        return true;
    }
    __device__ __forceinline__ void retire(uint32_t k, const HitRec& best, bool) const { hits[index(k)] = make_hit(sc, best); }
    // instrumented runs only: the ray's node visits and triangle tests replace two barycentrics of the (scratch) hit record
    __device__ __forceinline__ void note(uint32_t k, uint32_t nodes, uint32_t tris) const { hits[index(k)].alpha = (float)nodes; hits[index(k)].beta = (float)tris; }
};

// rays[queue ? queue[i] : i] for i < (n_dev ? *n_dev : n_host); *work_counter must be 0 at launch
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_closest(DeviceScene sc, const rtx_ray* __restrict__ rays, const uint32_t* __restrict__ queue,
                                                    uint32_t n_host, const uint32_t* __restrict__ n_dev, rtx_hit* __restrict__ hits,
                                                    uint2* spill, uint32_t spill_stride, uint32_t* work_counter, unsigned long long* counters)
{
    __shared__ unsigned long long lds[kBlock * kLdsStack];
    LaneStack<kBlock, kLdsStack> stk; stk.init(lds, spill, spill_stride);
    ClosestPolicy pol{sc, rays, queue, hits};
    // the instrumented pass counts what one lane per ray visits, without helpers' speculative work
    __shared__ TailShare<kBlock> ts;
    traverse_dyn<ClosestPolicy, kModeClosest, COUNT, kRefillClosest, kBlock, kLdsStack, kTriUnroll, kLeafLanesClosest, COUNT ? 0 : kShare>(sc, stk, n_dev ? *n_dev : n_host, work_counter, pol, counters, &ts);
}

#ifdef RTX_TUNING
// (Tuning build only: measured slower than the per-lane kernel on camera rays at 1080p, 0.70 vs 0.58 ms on the
// 1 M-triangle soup and 1.23 vs 1.17 ms per path-traced pass of the mixed scene, faster only at 4K, 1.66 vs 1.89 ms;
// coherent rays do not diverge in the per-lane kernel either, so all a packet saves is the redundant fetches.)
// Closest hit for rays that come in coherent runs (camera rays in pixel-block order): every wave takes 64 consecutive
// items at a time; if they leave one point within a narrow cone it walks the tree once for all of them
// (packet_closest, traverse_packet.h), otherwise it puts the 64 items on `rest`, which the per-lane kernel (k_closest)
// then serves. Same hit records either way. *work and *rest_count must be 0 at launch.
__global__ __launch_bounds__(kBlock) void k_closest_packet(DeviceScene sc, const rtx_ray* __restrict__ rays, const uint32_t* __restrict__ queue, uint32_t n_host, const uint32_t* __restrict__ n_dev,
                                                           rtx_hit* __restrict__ hits, uint2* spill, uint32_t spill_stride, uint32_t* work, uint32_t* rest, uint32_t* rest_count)
{
    __shared__ PacketStacks<kBlock> ps;
    __shared__ unsigned long long s_key[kBlock];
    LaneStack<kBlock, 0> exact_stack; exact_stack.init(nullptr, spill, spill_stride);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t n = n_dev ? *n_dev : n_host, n_packets = (n + 63u) / 64u;
    uint32_t seg = blockIdx.x % (uint32_t)kWorkSegments;
    unsigned long long dry = 0ull;
    for (;;) {
        uint32_t j = 0;
        if (lane == 0) j = atomicAdd(work + seg * (uint32_t)kWorkStride, 1u);
        j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
        const uint32_t g = work_chunk(j, seg);
        if (j >= (1u << 26) || g >= n_packets) {
            dry |= 1ull << seg;
            if (dry == ~0ull) break;
            do { seg = (seg + 1u) & (uint32_t)(kWorkSegments - 1); } while ((dry >> seg) & 1ull);
            continue;
        }
        const uint32_t k = g * 64u + lane;
        const bool has = k < n;
        const uint32_t idx = has ? (queue ? queue[k] : k) : 0u;
        rtx_ray r; r.o[0] = r.o[1] = r.o[2] = 0.0f; r.d[0] = r.d[1] = r.d[2] = 1.0f;
        if (has) r = rays[idx];
        // coherent: one origin (bit for bit) and every direction within ~8 degrees of the first lane's
        const int fl = __builtin_ctzll(__ballot(has));
        const float fox = __shfl(r.o[0], fl, 64), foy = __shfl(r.o[1], fl, 64), foz = __shfl(r.o[2], fl, 64);
        const float fdx = __shfl(r.d[0], fl, 64), fdy = __shfl(r.d[1], fl, 64), fdz = __shfl(r.d[2], fl, 64);
        const bool fits = !has || (r.o[0] == fox && r.o[1] == foy && r.o[2] == foz && (r.d[0] * fdx + r.d[1] * fdy + r.d[2] * fdz) > 0.99f * (fdx * fdx + fdy * fdy + fdz * fdz));
        if (__all(fits)) {
            HitRec best;
            packet_closest<kBlock>(sc, ps, s_key + (wv << 6), exact_stack, has, r.o[0], r.o[1], r.o[2], r.d[0], r.d[1], r.d[2], best);
            if (has) hits[idx] = make_hit(sc, best);
        } else {
            uint32_t at = 0;
            const uint32_t cnt = (uint32_t)__popcll(__ballot(has));
            if (lane == 0) at = atomicAdd(rest_count, cnt);
            at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
            if (has) rest[at + lane] = idx;
        }
    }
}
#endif

#ifdef RTX_TUNING
// Schedules under test (A/B harness, tuning build only)
template <int REFILL, int UNROLL, int SCHED, int LDS_N, int WAVES, int SHARE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_closest_ab(DeviceScene sc, const rtx_ray* __restrict__ rays, uint32_t n, rtx_hit* __restrict__ hits,
                                                       uint2* spill, uint32_t spill_stride, uint32_t* work_counter)
{
    __shared__ unsigned long long lds[kBlock * LDS_N];
    LaneStack<kBlock, LDS_N> stk; stk.init(lds, spill, spill_stride);
    ClosestPolicy pol{sc, rays, nullptr, hits};
    __shared__ TailShare<kBlock> ts;
    traverse_dyn<ClosestPolicy, kModeClosest, false, REFILL, kBlock, LDS_N, UNROLL, SCHED, SHARE>(sc, stk, n, work_counter, pol, nullptr, &ts);
}
#endif

// ---- visibility: Scene::visible, RTBase/Scene.h:170-178 -> traverseVisible, Geometry.h:568-610 ----
// A segment whose ray misses the root box is NOT visible (Geometry.h:571-572).
struct VisiblePolicy {
    const float* segments; uint8_t* visible;
    __device__ __forceinline__ bool fetch(uint32_t k, float o[3], float d[3], float& maxT) const { segment_to_ray(segments + (size_t)k * 6, o, d, maxT); return true; }
    __device__ __forceinline__ void retire(uint32_t k, const HitRec& best, bool reached) const { visible[k] = (reached && best.pos == 0xFFFFFFFFu) ? 1 : 0; }
};

__global__ __launch_bounds__(kBlock) void k_visible(DeviceScene sc, const float* __restrict__ p1p2, uint32_t n, uint8_t* __restrict__ visible,
                                                    uint2* spill, uint32_t spill_stride, uint32_t* work_counter)
{
    __shared__ unsigned long long lds[kBlock * kLdsStack];
    LaneStack<kBlock, kLdsStack> stk; stk.init(lds, spill, spill_stride);
    VisiblePolicy pol{p1p2, visible};
    __shared__ TailShare<kBlock> ts;
    traverse_dyn<VisiblePolicy, kModeAny, false, kRefill, kBlock, kLdsStack, kTriUnroll, kLeafLanes, kShare>(sc, stk, n, work_counter, pol, nullptr, &ts);
}

#ifdef RTX_TUNING
// Schedules under test (A/B harness, tuning build only)
template <int REFILL, int UNROLL, int SCHED, int SHARE>
__global__ __launch_bounds__(kBlock) void k_visible_ab(DeviceScene sc, const float* __restrict__ p1p2, uint32_t n, uint8_t* __restrict__ visible,
                                                       uint2* spill, uint32_t spill_stride, uint32_t* work_counter)
{
    __shared__ unsigned long long lds[kBlock * kLdsStack];
    LaneStack<kBlock, kLdsStack> stk; stk.init(lds, spill, spill_stride);
    VisiblePolicy pol{p1p2, visible};
    __shared__ TailShare<kBlock> ts;
    traverse_dyn<VisiblePolicy, kModeAny, false, REFILL, kBlock, kLdsStack, UNROLL, SCHED, SHARE>(sc, stk, n, work_counter, pol, nullptr, &ts);
}
#endif

// One segment, inside a kernel that loops over segments per lane (the VPL gather)
template <int BLOCK, int LDS_N>
__device__ __forceinline__ bool segment_visible(const DeviceScene& sc, LaneStack<BLOCK, LDS_N>& stk,
                                                float p1x, float p1y, float p1z, float p2x, float p2y, float p2z)
{
    const float s[6] = {p1x, p1y, p1z, p2x, p2y, p2z};
    float o[3], d[3], maxT;
    segment_to_ray(s, o, d, maxT);
    HitRec best;
    bool root = traverse<true, false, BLOCK, LDS_N>(sc, stk, o[0], o[1], o[2], d[0], d[1], d[2], maxT, best, nullptr);
    return root && best.pos == 0xFFFFFFFFu;
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
