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
};

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
{
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// One segment per lane (has == false: the lane holds none). Returns true for lanes whose segment is NOT visible:
// occluded, or the ray misses the root box (Geometry.h:571-572). All lanes of the wave must call this together.
// COUNT: nodes / tris receive what the lane's own segment visited (the numbers a per-lane walk would count).
template <int BLOCK, bool COUNT>
__device__ __forceinline__ bool packet_occluded(const DeviceScene& sc, PacketStacks<BLOCK>& ps, LaneStack<BLOCK, 0>& exact_stack, bool has,
                                                float ox, float oy, float oz, float dx, float dy, float dz, float maxT, uint32_t* nodes, uint32_t* tris)
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
    {
        const unsigned long long m0 = __ballot(root && finite);
        if (m0 != 0ull) { s_ref[0] = 0u; s_mask[0] = m0; sp = 1; }
    }
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
            const uint32_t cr[4] = {cp[0], cp[1], cp[2], cp[3]};
            const float lx[4] = {np[0], np[1], np[2], np[3]}, ly[4] = {np[4], np[5], np[6], np[7]}, lz[4] = {np[8], np[9], np[10], np[11]};
            const float hx[4] = {np[12], np[13], np[14], np[15]}, hy[4] = {np[16], np[17], np[18], np[19]}, hz[4] = {np[20], np[21], np[22], np[23]};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (cr[c] == kEmptySlot) continue;          // (wave-uniform)
                // no NaN can occur here (finite reciprocals), so plain min / max equal the reference's selects
                const float t0x = (lx[c] - ox) * ix, t1x = (hx[c] - ox) * ix;
                const float t0y = (ly[c] - oy) * iy, t1y = (hy[c] - oy) * iy;
                const float t0z = (lz[c] - oz) * iz, t1z = (hz[c] - oz) * iz;
                const float tentry = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
                const float texit = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
                const bool hit = in && !(tentry > texit || texit < 0) && !(tentry > bound);
                const unsigned long long hm = __ballot(hit);
                if (hm != 0ull) { s_ref[sp] = cr[c]; s_mask[sp] = hm; sp++; }
            }
        } else {
            const uint32_t first = ref & 0x07FFFFFFu, nt = ((ref >> 27) & 15u) + 1u;
            for (uint32_t k = 0; k < nt; k++) {
                if ((mask & ~done) == 0ull) break;
                ConstFloat* tp = trif + (size_t)(first + k) * 9;
                TriIsect T;
#pragma unroll
                for (int q = 0; q < 3; q++) { T.e1[q] = tp[q]; T.e2[q] = tp[3 + q]; T.v2[q] = tp[6 + q]; }
                if (in && !occluded) {
                    if (COUNT) (*tris)++;
                    float t, u, v;
                    if (tri_test(T, ox, oy, oz, dx, dy, dz, t, u, v) && t <= maxT) occluded = true;
                }
                done = __ballot(occluded);
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

} // namespace rtx
