// traverse_packet.h — visibility (any hit) for a wave of COHERENT segments: one traversal per wave instead of one per lane.
//
// traverse_dyn.h gives every lane its own ray, stack and node fetches; that is the right schedule for rays that go
// different ways. Segments that leave neighbouring surface points for ONE far point (a pixel block gathering from a
// virtual point light, RTBase/Renderer.h:227-244) pass the same nodes, and then 64 lanes fetching the same 128-byte
// node through the vector memory path and keeping 64 identical stacks is wasted work. Here the wave walks the tree
// once: the stack is wave-uniform (node reference + the mask of lanes whose segment reached that node), a node or a
// triangle is fetched once through the scalar cache, and every lane tests its own segment against it.
//
// What a lane's segment is tested against is exactly what traverse_dyn / traverse() test it against: a lane is in
// the mask of a node only if its own slab test passed for that node's box and for every box above it (same
// arithmetic, same pruning against maxT), and it tests the triangles of exactly the leaves it reaches that way.
// Visibility is an existence question, so the order in which leaves are visited cannot change the answer; lanes
// that have found an occluder drop out of all masks.
//
// Segments with a non-finite reciprocal direction component take the exact path over the binary tree
// (traverse_exact, traverse.h), one lane at a time after the packet walk, as in the other kernels.
#pragma once

#include "traverse.h"

namespace rtx {

constexpr int kPacketStack = 96;     // wave-uniform stack entries; scenes that can need more use the per-lane kernels

typedef const __attribute__((address_space(4))) uint32_t ConstWord;
typedef const __attribute__((address_space(4))) float ConstFloat;

template <int BLOCK>
struct PacketStacks {
    uint32_t ref[BLOCK / 64][kPacketStack];
    unsigned long long mask[BLOCK / 64][kPacketStack];
    uint32_t src[BLOCK / 64][64];        // leaf step: the lanes of the current mask, packed
    uint32_t hit[BLOCK / 64][64];        // leaf step: lane l's segment was found occluded by another lane (zero between uses)
};

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
{
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// ceil(2^16 / n) for n = 1 .. 16: floor(q / n) == (q * kInv16[n]) >> 16 for q < 4096
__device__ const uint32_t kInv16[17] = {0, 65536, 32768, 21846, 16384, 13108, 10923, 9363, 8192, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4096};

// One segment per lane (has == false: the lane holds none). Returns true for lanes whose segment is NOT visible:
// occluded, or the ray misses the root box (Geometry.h:571-572). All lanes of the wave must call this together.
// COUNT: nodes / tris receive what the lane's own segment visited (the numbers a per-lane walk would count).
// wave_stats (tuning builds, may be null): [0] node steps, [1] lanes in their masks, [2] triangle steps, [3] lanes in theirs,
// [7] node steps with at most 16 lanes
template <int BLOCK, bool COUNT>
__device__ __forceinline__ bool packet_occluded(const DeviceScene& sc, PacketStacks<BLOCK>& ps, LaneStack<BLOCK, 0>& exact_stack, bool has,
                                                float ox, float oy, float oz, float dx, float dy, float dz, float maxT, uint32_t* nodes, uint32_t* tris,
                                                uint32_t* wave_stats = nullptr)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;                                    // Ray::init, Geometry.h:25
    const bool root = has && slab_reference(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], ox, oy, oz, ix, iy, iz);
    const bool finite = (fabsf(ix) <= FLT_MAX) && (fabsf(iy) <= FLT_MAX) && (fabsf(iz) <= FLT_MAX);
    const float bound = maxT + (fabsf(maxT) * kCullRel + kCullAbs);
    bool occluded = false;
    uint32_t* s_ref = ps.ref[wv]; unsigned long long* s_mask = ps.mask[wv];
    uint32_t sp = 0;
    unsigned long long done = 0ull;                          // lanes that have found an occluder
    const unsigned long long m0 = __ballot(root && finite);
    if (m0 != 0ull) { s_ref[0] = 0u; s_mask[0] = m0; sp = 1; }
    // Segments of one packet mostly travel the same way along every axis. Then the near plane of a box (the low one
    // for a ray that travels up the axis, the high one otherwise) is the same plane for the whole wave and is picked
    // by the address the scalar load reads, and (near - o) * inv <= (far - o) * inv holds in floating point too
    // (subtraction and multiplication by a finite reciprocal of one sign are monotone), so tentry / texit are the
    // same numbers as the minima and maxima of the two products that the general form takes.
    const unsigned long long negx = __ballot(ix < 0.0f) & m0, negy = __ballot(iy < 0.0f) & m0, negz = __ballot(iz < 0.0f) & m0;
    const bool one_way = (negx == 0ull || negx == m0) && (negy == 0ull || negy == m0) && (negz == 0ull || negz == m0);
    const uint32_t offx = negx ? 12u : 0u, offy = negy ? 16u : 4u, offz = negz ? 20u : 8u;     // float offsets of the near planes inside a Node4
    ConstFloat* nodes4 = (ConstFloat*)sc.nodes4;
    ConstFloat* trif = (ConstFloat*)sc.tris;
    while (sp > 0) {
        sp--;
        const uint32_t ref = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ref[sp]);
        const unsigned long long mask = uniform64(s_mask[sp]) & ~done;
        if (mask == 0ull) continue;
        const bool in = ((mask >> lane) & 1ull) != 0ull;
        if (!(ref & kLeafBit)) {
            ConstFloat* np = nodes4 + (size_t)ref * 32;          // Node4: lo[axis][slot], hi[axis][slot], child[4]
            ConstWord* cp = (ConstWord*)(np + 24);
            if (COUNT && in) (*nodes)++;
            if (wave_stats) { wave_stats[0]++; wave_stats[1] += (uint32_t)__popcll(mask); if (__popcll(mask) <= 16) wave_stats[7]++; }
            const uint32_t cr[4] = {cp[0], cp[1], cp[2], cp[3]};
            if (one_way) {
                ConstFloat* pnx = np + offx; ConstFloat* pny = np + offy; ConstFloat* pnz = np + offz;
                ConstFloat* pfx = np + (offx ^ 12u); ConstFloat* pfy = np + (offy ^ 20u); ConstFloat* pfz = np + (offz ^ 28u);
                const float nx[4] = {pnx[0], pnx[1], pnx[2], pnx[3]}, ny[4] = {pny[0], pny[1], pny[2], pny[3]}, nz[4] = {pnz[0], pnz[1], pnz[2], pnz[3]};
                const float fx[4] = {pfx[0], pfx[1], pfx[2], pfx[3]}, fy[4] = {pfy[0], pfy[1], pfy[2], pfy[3]}, fz[4] = {pfz[0], pfz[1], pfz[2], pfz[3]};
                // all seven loads of the node go out together (left alone, the compiler loads the planes of each slot
                // behind the test of its child reference: four dependent round trips per node instead of one)
                asm volatile("" :: "s"(nx[0]), "s"(nx[1]), "s"(nx[2]), "s"(nx[3]), "s"(ny[0]), "s"(ny[1]), "s"(ny[2]), "s"(ny[3]), "s"(nz[0]), "s"(nz[1]), "s"(nz[2]), "s"(nz[3]));
                asm volatile("" :: "s"(fx[0]), "s"(fx[1]), "s"(fx[2]), "s"(fx[3]), "s"(fy[0]), "s"(fy[1]), "s"(fy[2]), "s"(fy[3]), "s"(fz[0]), "s"(fz[1]), "s"(fz[2]), "s"(fz[3]));
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (cr[c] == kEmptySlot) continue;          // (wave-uniform)
                    const float tentry = fmaxf(fmaxf((nx[c] - ox) * ix, (ny[c] - oy) * iy), (nz[c] - oz) * iz);
                    const float texit = fminf(fminf((fx[c] - ox) * ix, (fy[c] - oy) * iy), (fz[c] - oz) * iz);
                    const bool hit = in && !(tentry > texit || texit < 0) && !(tentry > bound);
                    const unsigned long long hm = __ballot(hit);
                    if (hm != 0ull) { s_ref[sp] = cr[c]; s_mask[sp] = hm; sp++; }
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (cr[c] == kEmptySlot) continue;
                    // no NaN can occur here (finite reciprocals), so plain min / max equal the reference's selects
                    const float t0x = (np[c] - ox) * ix, t1x = (np[12 + c] - ox) * ix;
                    const float t0y = (np[4 + c] - oy) * iy, t1y = (np[16 + c] - oy) * iy;
                    const float t0z = (np[8 + c] - oz) * iz, t1z = (np[20 + c] - oz) * iz;
                    const float tentry = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
                    const float texit = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
                    const bool hit = in && !(tentry > texit || texit < 0) && !(tentry > bound);
                    const unsigned long long hm = __ballot(hit);
                    if (hm != 0ull) { s_ref[sp] = cr[c]; s_mask[sp] = hm; sp++; }
                }
            }
        } else {
            const uint32_t first = ref & 0x07FFFFFFu, nt = ((ref >> 27) & 15u) + 1u;
            const uint32_t na = (uint32_t)__popcll(mask), passes = (na * nt + 63u) / 64u;
            if (passes * 4u < nt * 3u) {      // (a packed pass costs about a third more than a step of the loop over triangles)
                // Few lanes reach this leaf: one lane per (segment, triangle) pair instead of one step per triangle with
                // most lanes idle. Pair q = (the q / nt-th lane of the mask, triangle q % nt); the segment comes over by
                // lane permutes, the result goes back through a flag per lane in LDS.
                uint32_t* s_src = ps.src[wv]; uint32_t* s_hit = ps.hit[wv];
                if (in) s_src[__popcll(mask & ((1ull << lane) - 1ull))] = lane;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                const uint32_t pairs = na * nt, inv = kInv16[nt];
                bool any_hit = false;
                for (uint32_t q0 = 0; q0 < pairs; q0 += 64u) {
                    const uint32_t q = q0 + lane;
                    const bool valid = q < pairs;
                    const uint32_t r = valid ? (q * inv) >> 16 : 0u, ti = q - r * nt;
                    const int sl = (int)s_src[r];
                    const float sox = __shfl(ox, sl, 64), soy = __shfl(oy, sl, 64), soz = __shfl(oz, sl, 64);
                    const float sdx = __shfl(dx, sl, 64), sdy = __shfl(dy, sl, 64), sdz = __shfl(dz, sl, 64);
                    const float smax = __shfl(maxT, sl, 64);
                    if (wave_stats) { wave_stats[2]++; wave_stats[3] += min(64u, pairs - q0); }
                    bool h = false;
                    if (valid) {
                        const TriIsect T = sc.tris[first + ti];
                        float t, u, v;
                        h = tri_test(T, sox, soy, soz, sdx, sdy, sdz, t, u, v) && t <= smax;
                        if (h) s_hit[sl] = 1u;
                    }
                    any_hit = any_hit || __any(h);
                }
                if (COUNT && in) (*tris) += nt;
                if (any_hit) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    if (s_hit[lane]) { occluded = true; s_hit[lane] = 0u; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    done = __ballot(occluded);
                }
            } else {
                for (uint32_t k = 0; k < nt; k++) {
                    if ((mask & ~done) == 0ull) break;
                    ConstFloat* tp = trif + (size_t)(first + k) * 9;
                    TriIsect T;
#pragma unroll
                    for (int q = 0; q < 3; q++) { T.e1[q] = tp[q]; T.e2[q] = tp[3 + q]; T.v2[q] = tp[6 + q]; }
                    if (wave_stats) { wave_stats[2]++; wave_stats[3] += (uint32_t)__popcll(mask & ~done); }
                    if (in && !occluded) {
                        if (COUNT) (*tris)++;
                        float t, u, v;
                        if (tri_test(T, ox, oy, oz, dx, dy, dz, t, u, v) && t <= maxT) occluded = true;
                    }
                    done = __ballot(occluded);
                }
            }
        }
    }
    if (root && !finite) {
        HitRec best; best.t = FLT_MAX; best.u = best.v = 0; best.pos = 0xFFFFFFFFu;
        traverse_exact<true, BLOCK, 0>(sc, exact_stack, ox, oy, oz, dx, dy, dz, ix, iy, iz, maxT, best);
        occluded = best.pos != 0xFFFFFFFFu;
    }
    return has && (occluded || !root);
}

// ---- closest hit for a wave of coherent rays (camera rays of a pixel block) ---------------------------------
// The same walk with a per-lane bound: a lane joins a child's mask only if its own slab test passes and the box
// does not start beyond its best hit so far (the pruning rule of traverse(), same margin), so every lane tests a
// superset-free, order-independent candidate set: exactly the triangles traverse() would test or could prune, and
// the minimum over (t, position) is the same whatever the order. Children are pushed far to near by the entry
// distance of the first lane that hits them (rays of a packet see the children in the same order, and the order
// only decides how early the bounds shrink).
// During the walk a lane keeps (t, position) only; the barycentrics of the winner are recomputed at the end by one
// more triangle test (same inputs, same instructions, same bits).
__device__ __forceinline__ unsigned long long pack_tp(float t, uint32_t pos) { return ((unsigned long long)__float_as_uint(t) << 32) | pos; }

template <int BLOCK>
__device__ __forceinline__ void packet_closest(const DeviceScene& sc, PacketStacks<BLOCK>& ps, unsigned long long* s_key /* [64], this wave's */,
                                               LaneStack<BLOCK, 0>& exact_stack, bool has,
                                               float ox, float oy, float oz, float dx, float dy, float dz, HitRec& best)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;                                    // Ray::init, Geometry.h:25
    const bool root = has && slab_reference(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], ox, oy, oz, ix, iy, iz);
    const bool finite = (fabsf(ix) <= FLT_MAX) && (fabsf(iy) <= FLT_MAX) && (fabsf(iz) <= FLT_MAX);
    best.t = FLT_MAX; best.u = best.v = 0; best.pos = 0xFFFFFFFFu;
    float bound = FLT_MAX;
    uint32_t* s_ref = ps.ref[wv]; unsigned long long* s_mask = ps.mask[wv];
    uint32_t sp = 0;
    const unsigned long long m0 = __ballot(root && finite);
    if (m0 != 0ull) { s_ref[0] = 0u; s_mask[0] = m0; sp = 1; }
    const unsigned long long negx = __ballot(ix < 0.0f) & m0, negy = __ballot(iy < 0.0f) & m0, negz = __ballot(iz < 0.0f) & m0;
    const bool one_way = (negx == 0ull || negx == m0) && (negy == 0ull || negy == m0) && (negz == 0ull || negz == m0);
    const uint32_t offx = negx ? 12u : 0u, offy = negy ? 16u : 4u, offz = negz ? 20u : 8u;     // float offsets of the near planes inside a Node4
    ConstFloat* nodes4 = (ConstFloat*)sc.nodes4;
    ConstFloat* trif = (ConstFloat*)sc.tris;
    while (sp > 0) {
        sp--;
        const uint32_t ref = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ref[sp]);
        const unsigned long long mask = uniform64(s_mask[sp]);
        const bool in = ((mask >> lane) & 1ull) != 0ull;
        if (!(ref & kLeafBit)) {
            ConstFloat* np = nodes4 + (size_t)ref * 32;          // Node4: lo[axis][slot], hi[axis][slot], child[4]
            ConstWord* cp = (ConstWord*)(np + 24);
            uint32_t cr[4] = {cp[0], cp[1], cp[2], cp[3]};
            unsigned long long hm[4]; uint32_t key[4];
            if (one_way) {
                ConstFloat* pnx = np + offx; ConstFloat* pny = np + offy; ConstFloat* pnz = np + offz;
                ConstFloat* pfx = np + (offx ^ 12u); ConstFloat* pfy = np + (offy ^ 20u); ConstFloat* pfz = np + (offz ^ 28u);
                const float nx[4] = {pnx[0], pnx[1], pnx[2], pnx[3]}, ny[4] = {pny[0], pny[1], pny[2], pny[3]}, nz[4] = {pnz[0], pnz[1], pnz[2], pnz[3]};
                const float fx[4] = {pfx[0], pfx[1], pfx[2], pfx[3]}, fy[4] = {pfy[0], pfy[1], pfy[2], pfy[3]}, fz[4] = {pfz[0], pfz[1], pfz[2], pfz[3]};
                asm volatile("" :: "s"(nx[0]), "s"(nx[1]), "s"(nx[2]), "s"(nx[3]), "s"(ny[0]), "s"(ny[1]), "s"(ny[2]), "s"(ny[3]), "s"(nz[0]), "s"(nz[1]), "s"(nz[2]), "s"(nz[3]));
                asm volatile("" :: "s"(fx[0]), "s"(fx[1]), "s"(fx[2]), "s"(fx[3]), "s"(fy[0]), "s"(fy[1]), "s"(fy[2]), "s"(fy[3]), "s"(fz[0]), "s"(fz[1]), "s"(fz[2]), "s"(fz[3]));
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float tentry = fmaxf(fmaxf((nx[c] - ox) * ix, (ny[c] - oy) * iy), (nz[c] - oz) * iz);
                    const float texit = fminf(fminf((fx[c] - ox) * ix, (fy[c] - oy) * iy), (fz[c] - oz) * iz);
                    const bool hit = in && !(tentry > texit || texit < 0) && !(tentry > bound);
                    hm[c] = cr[c] == kEmptySlot ? 0ull : __ballot(hit);
                    const int fl = hm[c] ? __builtin_ctzll(hm[c]) : 0;
                    key[c] = hm[c] ? (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(fmaxf(tentry, 0.0f)), fl) : 0xFFFFFFFFu;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    // no NaN can occur here (finite reciprocals), so plain min / max equal the reference's selects
                    const float t0x = (np[c] - ox) * ix, t1x = (np[12 + c] - ox) * ix;
                    const float t0y = (np[4 + c] - oy) * iy, t1y = (np[16 + c] - oy) * iy;
                    const float t0z = (np[8 + c] - oz) * iz, t1z = (np[20 + c] - oz) * iz;
                    const float tentry = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
                    const float texit = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
                    const bool hit = in && !(tentry > texit || texit < 0) && !(tentry > bound);
                    hm[c] = cr[c] == kEmptySlot ? 0ull : __ballot(hit);
                    const int fl = hm[c] ? __builtin_ctzll(hm[c]) : 0;
                    key[c] = hm[c] ? (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(fmaxf(tentry, 0.0f)), fl) : 0xFFFFFFFFu;
                }
            }
            // far to near onto the stack (wave-uniform values: scalar compares and selects); misses carry the largest key
#define RTX_PSWAP(a, b) { const bool sw = key[a] < key[b]; const uint32_t k0 = sw ? key[b] : key[a], k1 = sw ? key[a] : key[b]; const uint32_t r0 = sw ? cr[b] : cr[a], r1 = sw ? cr[a] : cr[b]; \
                          const unsigned long long h0 = sw ? hm[b] : hm[a], h1 = sw ? hm[a] : hm[b]; key[a] = k0; key[b] = k1; cr[a] = r0; cr[b] = r1; hm[a] = h0; hm[b] = h1; }
            RTX_PSWAP(0, 1) RTX_PSWAP(2, 3) RTX_PSWAP(0, 2) RTX_PSWAP(1, 3) RTX_PSWAP(1, 2)      // descending keys
#undef RTX_PSWAP
#pragma unroll
            for (int c = 0; c < 4; c++) if (hm[c] != 0ull) { s_ref[sp] = cr[c]; s_mask[sp] = hm[c]; sp++; }
        } else {
            const uint32_t first = ref & 0x07FFFFFFu, nt = ((ref >> 27) & 15u) + 1u;
            const uint32_t na = (uint32_t)__popcll(mask), passes = (na * nt + 63u) / 64u;
            if (passes * 4u < nt * 3u) {
                // one lane per (ray, triangle) pair; the candidates meet in the ray's (t, position) word in LDS, whose unsigned
                // minimum is the reference's rule (smaller t, then the triangle it visits first)
                uint32_t* s_src = ps.src[wv];
                if (in) { s_src[__popcll(mask & ((1ull << lane) - 1ull))] = lane; s_key[lane] = pack_tp(best.t, best.pos); }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                const uint32_t pairs = na * nt, inv = kInv16[nt];
                for (uint32_t q0 = 0; q0 < pairs; q0 += 64u) {
                    const uint32_t q = q0 + lane;
                    const bool valid = q < pairs;
                    const uint32_t r = valid ? (q * inv) >> 16 : 0u, ti = q - r * nt;
                    const int sl = (int)s_src[r];
                    const float sox = __shfl(ox, sl, 64), soy = __shfl(oy, sl, 64), soz = __shfl(oz, sl, 64);
                    const float sdx = __shfl(dx, sl, 64), sdy = __shfl(dy, sl, 64), sdz = __shfl(dz, sl, 64);
                    if (valid) {
                        const TriIsect T = sc.tris[first + ti];
                        float t, u, v;
                        if (tri_test(T, sox, soy, soz, sdx, sdy, sdz, t, u, v)) atomicMin(&s_key[sl], pack_tp(t, first + ti));
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                if (in) { const unsigned long long k = s_key[lane]; best.t = __uint_as_float((uint32_t)(k >> 32)); best.pos = (uint32_t)k; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            } else {
                for (uint32_t k = 0; k < nt; k++) {
                    ConstFloat* tp = trif + (size_t)(first + k) * 9;
                    TriIsect T;
#pragma unroll
                    for (int q = 0; q < 3; q++) { T.e1[q] = tp[q]; T.e2[q] = tp[3 + q]; T.v2[q] = tp[6 + q]; }
                    if (in) {
                        float t, u, v;
                        if (tri_test(T, ox, oy, oz, dx, dy, dz, t, u, v) && (t < best.t || (t == best.t && first + k < best.pos))) { best.t = t; best.pos = first + k; }
                    }
                }
            }
            bound = best.t + (fabsf(best.t) * kCullRel + kCullAbs);
        }
    }
    if (root && !finite) traverse_exact<false, BLOCK, 0>(sc, exact_stack, ox, oy, oz, dx, dy, dz, ix, iy, iz, 0.0f, best);
    else if (best.pos != 0xFFFFFFFFu) { float t; tri_test(sc.tris[best.pos], ox, oy, oz, dx, dy, dz, t, best.u, best.v); }
}

} // namespace rtx
