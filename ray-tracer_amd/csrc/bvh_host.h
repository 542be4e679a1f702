// bvh_host.h — host-side acceleration-structure build of the MI355X renderer core.
//
// Two stages:
//   1. build_bvh2: the reference's binary tree (BVHTree::build, RTBase/Geometry.h:309-504)
//      reproduced node for node — same node numbering, bounds, index permutation — but
//      with one binning sweep per node instead of the reference's 45 full sweeps.
//   2. flatten_bvh4: that tree collapsed into 128-byte 4-wide nodes for the device, with the
//      triangles re-ordered into the reference's leaf visit order so that the position of a
//      triangle in device memory doubles as its tie-break rank.
#pragma once

#include <cstdint>
#include <functional>
#include <vector>

namespace rtx {

// Triangle::init, RTBase/Geometry.h:74-92: the values the build and the ray test read.
struct TriPrep {
    float e1[3], e2[3], v2[3];          // e1 = v0 - v2, e2 = v1 - v2 (Geometry.h:82-83)
    float centre[3], minP[3], maxP[3];  // Geometry.h:89-91
};

// BVHTree::Node, RTBase/Geometry.h:293-300 (leaf <=> child_l == 0 && child_r == 0).
struct Bvh2Node {
    float bmin[3], bmax[3];
    uint32_t start, end, child_l, child_r;
};

struct Bvh2 {
    std::vector<Bvh2Node> nodes;
    std::vector<uint32_t> indices;  // BVHTree::indices, Geometry.h:304
    uint32_t depth = 0;             // value returned by BVHTree::split, Geometry.h:432-449
};

// 4-wide device node, one 128-byte cache line. child[k]:
//   0x7FFFFFFF           empty slot
//   bit31 = 0            index of another Node4
//   bit31 = 1            leaf: bits 0..26 first triangle position, bits 27..30 count-1 (1..16)
struct Node4 {
    float lo[3][4];   // lo[axis][slot]
    float hi[3][4];
    uint32_t child[4];
    uint32_t pad[4];
};
static_assert(sizeof(Node4) == 128, "Node4 must be one cache line");

constexpr uint32_t kEmptySlot = 0x7FFFFFFFu;
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kLeafMaxTris = 16;
constexpr uint32_t kMaxTriangles = 1u << 27;

// Binary node as the device's exact fallback path reads it (32 bytes):
// inner: a = child_l, b = child_r; leaf: a = kLeafBit | first position, b = triangle count.
struct Node2Dev {
    float bmin[3], bmax[3];
    uint32_t a, b;
};

struct Bvh4 {
    std::vector<Node4> nodes;
    std::vector<Node2Dev> nodes2;
    std::vector<uint32_t> order;   // order[position] = original triangle index
    uint32_t depth = 0;            // levels of Node4 below (and including) the root
    float root_min[3], root_max[3];
};

// fn(begin, end) over [0, n) on up to 16 host threads
void parallel_ranges(uint64_t n, const std::function<void(uint64_t, uint64_t)>& fn);

void prepare_triangles(const float* positions, uint64_t n, std::vector<TriPrep>& out,
                       float scene_min[3], float scene_max[3]);
void build_bvh2(const std::vector<TriPrep>& tris, const float scene_min[3], const float scene_max[3], Bvh2& out);
void flatten_bvh4(const Bvh2& in, Bvh4& out);

} // namespace rtx
