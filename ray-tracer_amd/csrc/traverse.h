// traverse.h — device-side BVH traversal: closest hit and any hit (visibility).
//
// Semantics follow BVHTree::traverse / traverseVisible (RTBase/Geometry.h:506-610):
//   * a triangle is a candidate iff the ray passes AABB::rayAABB (Geometry.h:213-228) for
//     every box on the path from the root to the triangle's leaf, and
//     Triangle::rayIntersect (Geometry.h:95-126) accepts it;
//   * closest hit = candidate with the smallest t; equal t is resolved in favour of the
//     triangle the reference visits first (right-first depth-first leaf order, strict '<'
//     at Geometry.h:535). Triangles are stored in that order, so "first" = lower position.
//
// Fast path (all three reciprocal direction components finite): 4-wide nodes, near-first
// order, pruning against the best t so far. Child boxes are exact subsets of their
// ancestors' boxes and IEEE subtraction / multiplication are monotonic, so "child box hit"
// implies "every ancestor box hit" and skipping the collapsed intermediate boxes cannot add
// or remove candidates. Pruning keeps a margin (kCullRel, kCullAbs) because a computed
// triangle t may fall slightly below its box's entry distance.
//
// Exact path (a direction component is +-0 or so small that its reciprocal overflows):
// 0*inf produces NaNs in the slab test and the implication above no longer holds, so these
// rays walk the binary tree with the reference's own box test and no pruning.
//
// All arithmetic here must compile without multiply-add fusion (-ffp-contract=off).
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>

#include "bvh_host.h"
#include "device_scene.h"

namespace rtx {

constexpr float kTriEps = 1e-8f;     // Triangle::epsilon, RTBase/Geometry.h:72
constexpr float kCullRel = 1.0f / 64.0f;
constexpr float kCullAbs = 1e-6f;

struct HitRec { float t, u, v; uint32_t pos; };   // pos = leaf-order position, 0xFFFFFFFF = none

// Work counters of an instrumented traversal (algorithmic traffic of the layout)
struct TravCount { uint32_t nodes, tris, groups; };   // groups: leaf-triangle fetches (up to four triangles each)

// Triangle::rayIntersect, RTBase/Geometry.h:95-126, same operation order.
__device__ __forceinline__ bool tri_test(const TriIsect& T, float ox, float oy, float oz, float dx, float dy, float dz,
                                         float& t, float& u, float& v)
{
    const float e1x = T.e1[0], e1y = T.e1[1], e1z = T.e1[2];
    const float e2x = T.e2[0], e2y = T.e2[1], e2z = T.e2[2];
    // p = Cross(dir, e2)
    float px = (dy * e2z) - (dz * e2y);
    float py = (dz * e2x) - (dx * e2z);
    float pz = (dx * e2y) - (dy * e2x);
    float det = ((px * e1x) + (py * e1y)) + (pz * e1z);
    if (fabsf(det) < kTriEps) return false;
    float invDet = 1.0f / det;
    float Tx = ox - T.v2[0], Ty = oy - T.v2[1], Tz = oz - T.v2[2];
    u = (((Tx * px) + (Ty * py)) + (Tz * pz)) * invDet;
    if ((u < 0 && fabsf(u) > kTriEps) || (u > 1 && fabsf(u - 1) > kTriEps)) return false;
    // q = Cross(T, e1)
    float qx = (Ty * e1z) - (Tz * e1y);
    float qy = (Tz * e1x) - (Tx * e1z);
    float qz = (Tx * e1y) - (Ty * e1x);
    v = (((dx * qx) + (dy * qy)) + (dz * qz)) * invDet;
    if ((v < 0 && fabsf(v) > kTriEps) || (u + v > 1 && fabsf(u + v - 1) > kTriEps)) return false;
    t = (((e2x * qx) + (e2y * qy)) + (e2z * qz)) * invDet;
    if (t < kTriEps) return false;
    return true;
}

// AABB::rayAABB(const Ray&), RTBase/Geometry.h:213-228, NaN behaviour included: Min/Max of
// Core.h:199-207 return the second operand unless the comparison holds; std::max(a,b) is
// (a<b)?b:a and std::min(a,b) is (b<a)?b:a.
__device__ __forceinline__ bool slab_reference(float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                               float ox, float oy, float oz, float ix, float iy, float iz)
{
    float t0x = (mnx - ox) * ix, t0y = (mny - oy) * iy, t0z = (mnz - oz) * iz;
    float t1x = (mxx - ox) * ix, t1y = (mxy - oy) * iy, t1z = (mxz - oz) * iz;
    float enx = t0x < t1x ? t0x : t1x, eny = t0y < t1y ? t0y : t1y, enz = t0z < t1z ? t0z : t1z;
    float exx = t0x > t1x ? t0x : t1x, exy = t0y > t1y ? t0y : t1y, exz = t0z > t1z ? t0z : t1z;
    float m = (enx < eny) ? eny : enx;
    float tentry = (m < enz) ? enz : m;
    float n = (exy < exx) ? exy : exx;
    float texit = (exz < n) ? exz : n;
    return !(tentry > texit || texit < 0);
}

__device__ __forceinline__ void consider(HitRec& best, float t, float u, float v, uint32_t pos)
{
    if (t < best.t || (t == best.t && pos < best.pos)) { best.t = t; best.u = u; best.v = v; best.pos = pos; }
}

// Traversal stack: the first LDS_N entries of each lane live in LDS (entry i of lane l at
// lds[i*BLOCK + l], 8 bytes: child reference in the low word, entry distance in the high
// word), deeper entries in a per-lane slab of global memory. The two parts are addressed
// through pointers that carry their address space: with plain pointers the compiler merges
// the two stores of push() (and the two loads of pop()) into one FLAT access on a selected
// pointer, which is slower than ds_write / global_store and waits on both memory counters.
typedef __attribute__((address_space(3))) unsigned long long LdsWord64;
typedef __attribute__((address_space(1))) unsigned long long GlobalWord64;

template <int BLOCK, int LDS_N>
struct LaneStack {
    LdsWord64* lds;        // this block's LDS array, BLOCK*LDS_N entries
    GlobalWord64* spill;   // this lane's first global entry; consecutive entries are `stride` apart
    uint32_t stride;
    int sp;
    __device__ __forceinline__ void init(unsigned long long* lds_array, uint2* spill_slab, uint32_t spill_stride)
    {
        lds = (LdsWord64*)lds_array;
        spill = (GlobalWord64*)(spill_slab + blockIdx.x * (uint32_t)BLOCK + threadIdx.x);
        stride = spill_stride; sp = 0;
    }
    // entry i of the lane `delta` lanes away in the same wave
    __device__ __forceinline__ uint2 peek_other(int i, int delta) const
    {
        unsigned long long e;
        if (i < LDS_N) e = lds[i * BLOCK + (int)threadIdx.x + delta];
        else e = (spill + delta)[(size_t)(i - LDS_N) * stride];
        return make_uint2((uint32_t)e, (uint32_t)(e >> 32));
    }
    __device__ __forceinline__ void push(uint32_t ref, float tentry)
    {
        const unsigned long long e = ((unsigned long long)__float_as_uint(tentry) << 32) | ref;
        if (sp < LDS_N) lds[sp * BLOCK + threadIdx.x] = e;
        else spill[(size_t)(sp - LDS_N) * stride] = e;
        sp++;
    }
    __device__ __forceinline__ uint2 pop()
    {
        sp--;
        unsigned long long e;
        if (sp < LDS_N) e = lds[sp * BLOCK + threadIdx.x];
        else e = spill[(size_t)(sp - LDS_N) * stride];
        return make_uint2((uint32_t)e, (uint32_t)(e >> 32));
    }
};

// Exact path: binary tree, reference box test, no pruning. ANY: returns as soon as a
// candidate with t <= maxT is found (best.pos != none signals occlusion).
template <bool ANY, int BLOCK, int LDS_N>
__device__ void traverse_exact(const DeviceScene& sc, LaneStack<BLOCK, LDS_N>& stk, float ox, float oy, float oz,
                               float dx, float dy, float dz, float ix, float iy, float iz, float maxT, HitRec& best)
{
    // binary depth is capped at 50 (Geometry.h:306): at most 51 pending nodes
    stk.sp = 0;
    stk.push(0, 0.0f);
    while (stk.sp > 0) {
        uint32_t ni = stk.pop().x;
        float4 a = sc.nodes2[2 * (size_t)ni], b = sc.nodes2[2 * (size_t)ni + 1];   // bmin.xyz bmax.x | bmax.yz a b
        (void)a;
        uint32_t ra = __float_as_uint(b.z), rb = __float_as_uint(b.w);
        if (ra & kLeafBit) {
            uint32_t first = ra & ~kLeafBit;
            for (uint32_t k = 0; k < rb; k++) {
                TriIsect T = sc.tris[first + k];
                float t, u, v;
                if (tri_test(T, ox, oy, oz, dx, dy, dz, t, u, v)) {
                    if (ANY) { if (t <= maxT) { best.pos = first + k; best.t = t; return; } }
                    else consider(best, t, u, v, first + k);
                }
            }
        } else {
            for (int c = 0; c < 2; c++) {
                uint32_t ci = c == 0 ? ra : rb;
                float4 ca = sc.nodes2[2 * (size_t)ci], cb = sc.nodes2[2 * (size_t)ci + 1];
                if (slab_reference(ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, ox, oy, oz, ix, iy, iz)) stk.push(ci, 0.0f);
            }
        }
    }
}

// Closest hit (ANY = false) or visibility (ANY = true) for one ray per lane.
// maxT is only read when ANY. COUNT adds work counters.
// Returns false when the ray misses the root box (the reference reports such a
// segment as NOT visible, Geometry.h:571-572; closest hit: a miss).
template <bool ANY, bool COUNT, int BLOCK, int LDS_N>
__device__ __forceinline__ bool traverse(const DeviceScene& sc, LaneStack<BLOCK, LDS_N>& stk,
                                         float ox, float oy, float oz, float dx, float dy, float dz,
                                         float maxT, HitRec& best, TravCount* cnt)
{
    best.t = FLT_MAX; best.u = 0; best.v = 0; best.pos = 0xFFFFFFFFu;
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;    // Ray::init, Geometry.h:25
    // root box first, as Geometry.h:509 / :571 do
    if (!slab_reference(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2],
                        ox, oy, oz, ix, iy, iz))
        return false;
    const bool finite = (fabsf(ix) <= FLT_MAX) && (fabsf(iy) <= FLT_MAX) && (fabsf(iz) <= FLT_MAX);
    if (!finite) { traverse_exact<ANY, BLOCK, LDS_N>(sc, stk, ox, oy, oz, dx, dy, dz, ix, iy, iz, maxT, best); return true; }

    stk.sp = 0;
    uint32_t cur = 0;                       // root Node4
    // pruning bound: boxes that start beyond it cannot hold the answer
    float bound = ANY ? (maxT + (fabsf(maxT) * kCullRel + kCullAbs)) : FLT_MAX;
    for (;;) {
        if (!(cur & kLeafBit)) {
            const float4* np = sc.nodes4 + (size_t)cur * 8;
            const float4 lox = np[0], loy = np[1], loz = np[2], hix = np[3], hiy = np[4], hiz = np[5];
            const float4 cf = np[6];
            if (COUNT) cnt->nodes++;
            const uint32_t ref[4] = {__float_as_uint(cf.x), __float_as_uint(cf.y), __float_as_uint(cf.z), __float_as_uint(cf.w)};
            const float lx[4] = {lox.x, lox.y, lox.z, lox.w}, ly[4] = {loy.x, loy.y, loy.z, loy.w}, lz[4] = {loz.x, loz.y, loz.z, loz.w};
            const float hx[4] = {hix.x, hix.y, hix.z, hix.w}, hy[4] = {hiy.x, hiy.y, hiy.z, hiy.w}, hz[4] = {hiz.x, hiz.y, hiz.z, hiz.w};
            float key[4]; uint32_t kref[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                // no NaN can occur here (finite reciprocals), so plain min/max equal the
                // reference's selects
                float t0x = (lx[c] - ox) * ix, t1x = (hx[c] - ox) * ix;
                float t0y = (ly[c] - oy) * iy, t1y = (hy[c] - oy) * iy;
                float t0z = (lz[c] - oz) * iz, t1z = (hz[c] - oz) * iz;
                float tentry = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
                float texit = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
                bool hit = !(tentry > texit || texit < 0) && (ref[c] != kEmptySlot) && !(tentry > bound);
                key[c] = hit ? tentry : FLT_MAX;
                kref[c] = hit ? ref[c] : kEmptySlot;
            }
            // sort the four slots by entry distance (5-comparator network), misses sink
#define RTX_CSWAP(a, b) { bool s = key[b] < key[a]; float tk = s ? key[b] : key[a]; float uk = s ? key[a] : key[b]; \
                          uint32_t tr = s ? kref[b] : kref[a]; uint32_t ur = s ? kref[a] : kref[b]; key[a] = tk; key[b] = uk; kref[a] = tr; kref[b] = ur; }
            RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
            // far ones first onto the stack, nearest continues
            if (kref[3] != kEmptySlot) stk.push(kref[3], key[3]);
            if (kref[2] != kEmptySlot) stk.push(kref[2], key[2]);
            if (kref[1] != kEmptySlot) stk.push(kref[1], key[1]);
            if (kref[0] != kEmptySlot) { cur = kref[0]; continue; }
        } else {
            const uint32_t first = cur & 0x07FFFFFFu, n = ((cur >> 27) & 15u) + 1u;
            for (uint32_t k = 0; k < n; k++) {
                TriIsect T = sc.tris[first + k];
                if (COUNT) cnt->tris++;
                float t, u, v;
                if (tri_test(T, ox, oy, oz, dx, dy, dz, t, u, v)) {
                    if (ANY) { if (t <= maxT) { best.pos = first + k; best.t = t; return true; } }
                    else consider(best, t, u, v, first + k);
                }
            }
            if (!ANY) bound = best.t + (fabsf(best.t) * kCullRel + kCullAbs);   // FLT_MAX stays +inf-like: FLT_MAX + x rounds to FLT_MAX or inf
        }
        // next pending subtree that can still matter
        for (;;) {
            if (stk.sp == 0) return true;
            uint2 e = stk.pop();
            if (!(__uint_as_float(e.y) > bound)) { cur = e.x; break; }
        }
    }
}


// ---- building blocks of the dynamic schedule (traverse_dyn.h) ---------------------------------
constexpr uint32_t kDone = kEmptySlot;

template <int BLOCK, int LDS_N>
__device__ __forceinline__ uint32_t pop_next(LaneStack<BLOCK, LDS_N>& stk, float bound)
{
    for (;;) {
        if (stk.sp == 0) return kDone;
        uint2 e = stk.pop();
        if (!(__uint_as_float(e.y) > bound)) return e.x;
    }
}

// One inner node: slab-test its four slots, push the far hits, return the nearest (or the
// next pending subtree when nothing is hit).
template <bool COUNT, int BLOCK, int LDS_N, int PARTIAL = 0>
__device__ __forceinline__ uint32_t step_inner(const DeviceScene& sc, LaneStack<BLOCK, LDS_N>& stk, uint32_t cur,
                                               float ox, float oy, float oz, float ix, float iy, float iz, float bound, TravCount* cnt)
{
    // 32-bit byte offset from the (wave-uniform) array base: the load then takes a scalar base and one
    // vector offset register instead of a 64-bit address computed per lane (node indices stay below 2^25)
    // The near plane of an axis is the low one for a ray that travels up that axis and the high one otherwise, so the
    // lane loads "near" and "far" planes by address instead of loading low and high and taking minima and maxima:
    // (near - o) * inv <= (far - o) * inv holds in floating point too (subtraction and multiplication by a finite
    // reciprocal of one sign are monotone), so tentry and texit are the same numbers as the reference's
    // min / max of the two products, 24 instructions per node fewer.
    // 32-bit byte offsets from the (wave-uniform) array base: the loads then take a scalar base and one vector offset
    // register instead of a 64-bit address computed per lane (node indices stay below 2^25)
    const char* nodes = (const char*)sc.nodes4;
    const uint32_t b = cur << 7;
    const uint32_t nx = ix < 0.0f ? 48u : 0u, ny = iy < 0.0f ? 64u : 16u, nz = iz < 0.0f ? 80u : 32u;
    const float4 nearx = *(const float4*)(nodes + (b | nx)), neary = *(const float4*)(nodes + (b | ny)), nearz = *(const float4*)(nodes + (b | nz));
    const float4 farx = *(const float4*)(nodes + (b | (nx ^ 48u))), fary = *(const float4*)(nodes + (b | (ny ^ 80u))), farz = *(const float4*)(nodes + (b | (nz ^ 112u)));
    const float4 cf = *(const float4*)(nodes + (b | 96u));
    if (COUNT) cnt->nodes++;
    const uint32_t ref[4] = {__float_as_uint(cf.x), __float_as_uint(cf.y), __float_as_uint(cf.z), __float_as_uint(cf.w)};
    // all four child references are needed whatever the boxes say: tested up front (and combined
    // without short-circuit below) so that their load stays with the other six instead of becoming
    // a dependent load behind the first box test
    const bool valid[4] = {ref[0] != kEmptySlot, ref[1] != kEmptySlot, ref[2] != kEmptySlot, ref[3] != kEmptySlot};
    // scalar subtract / multiply on purpose: the packed two-float forms cost more issue time than two scalar
    // instructions on this hardware (the library is also built without SLP vectorisation for that reason)
    const float O[3] = {ox, oy, oz}, I[3] = {ix, iy, iz};
    const float4 lo4[3] = {nearx, neary, nearz}, hi4[3] = {farx, fary, farz};
    float t0[3][4], t1[3][4];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        t0[a][0] = (lo4[a].x - O[a]) * I[a]; t0[a][1] = (lo4[a].y - O[a]) * I[a]; t0[a][2] = (lo4[a].z - O[a]) * I[a]; t0[a][3] = (lo4[a].w - O[a]) * I[a];
        t1[a][0] = (hi4[a].x - O[a]) * I[a]; t1[a][1] = (hi4[a].y - O[a]) * I[a]; t1[a][2] = (hi4[a].z - O[a]) * I[a]; t1[a][3] = (hi4[a].w - O[a]) * I[a];
    }
    float key[4]; uint32_t kref[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        // no NaN can occur here (finite reciprocals), so plain min/max equal the reference's selects
        float tentry = fmaxf(fmaxf(t0[0][c], t0[1][c]), t0[2][c]);
        float texit = fminf(fminf(t1[0][c], t1[1][c]), t1[2][c]);
        const bool hit = valid[c] & !(tentry > texit || texit < 0) & !(tentry > bound);
        key[c] = hit ? tentry : FLT_MAX;
        kref[c] = hit ? ref[c] : kEmptySlot;
    }
#define RTX_CSWAP(a, b) { bool s = key[b] < key[a]; float tk = s ? key[b] : key[a]; float uk = s ? key[a] : key[b]; \
                          uint32_t tr = s ? kref[b] : kref[a]; uint32_t ur = s ? kref[a] : kref[b]; key[a] = tk; key[b] = uk; kref[a] = tr; kref[b] = ur; }
    // closest hit: all four by entry distance (the order of the pending entries decides how much the best hit
    // prunes later). PARTIAL (visibility, whose answer does not depend on the order): the nearest only.
    RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2)
    if (PARTIAL == 0) { RTX_CSWAP(1, 3) RTX_CSWAP(1, 2) }
#undef RTX_CSWAP
    if (kref[3] != kEmptySlot) stk.push(kref[3], key[3]);
    if (kref[2] != kEmptySlot) stk.push(kref[2], key[2]);
    if (kref[1] != kEmptySlot) stk.push(kref[1], key[1]);
    if (kref[0] != kEmptySlot) return kref[0];
    return pop_next(stk, bound);
}

} // namespace rtx
