// bvh_device.h — the reference's BVH (BVHTree::build, RTBase/Geometry.h:309-504) built on the
// GPU, node for node and index for index what build_bvh2() builds on the host.
//
// The tree is grown one level at a time; every kernel works on all nodes of the level at once.
//
//   binning     evaluateSplit's 3 x 15 candidate planes (Geometry.h:314-343, 402-423) need, per
//               plane, the boxes and counts of the triangles on either side. Plane positions
//               of one axis do not decrease, so a triangle is "left" of plane i exactly when
//               i >= k, k = first plane with centre < pos[k]: 16 bins per axis, filled with
//               integer atomics (min / max of order-preserving float keys, counts), merged as
//               prefixes / suffixes by one thread per node. min, max and counts do not depend
//               on the order of accumulation, so the float costs equal the reference's.
//   partition   createSplit (Geometry.h:345-358) walks the range once and swaps every triangle
//               with centre > pos to the front. That leaves the "true" triangles in arrival
//               order at the front, and the others in the order of a queue that appends each
//               arrival at the back and, for every "true" seen after the first "false", moves
//               its front element to the back. Unrolled on a tape — arrival = new cell,
//               rotation number j = copy of cell j — the final queue is the cells from
//               (#rotations) on, and each cell's triangle is found by following "copy of"
//               links, which log2(n) rounds of pointer jumping resolve for all cells at once.
//   numbering   nodes are created level by level here and renumbered on the host into the
//               order split() creates them (children side by side, then the left subtree,
//               then the right, Geometry.h:381-386, 444-445).
//
// Known limit: when a box face is at +0.0 for some triangles and -0.0 for others, the
// reference's result carries the sign of whichever came last; atomics pick -0.0 for a
// minimum and +0.0 for a maximum. The two zeros compare equal everywhere on the path.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <vector>

#include "bvh_host.h"
#include "context.h"

namespace rtx {
namespace devbuild {

constexpr int kMaxLeafTris = 8;     // MAXNODE_TRIANGLES, Geometry.h:285
constexpr int kBins = 15;           // BUILD_BINS, Geometry.h:288
constexpr int kMaxDepth = 50;       // Geometry.h:306
constexpr int kBlk = 256;
constexpr uint32_t kNoNode = 0xFFFFFFFFu;
constexpr int kBinWords = 3 * (kBins + 1) * 7;    // per node: axis x bin x (min xyz, max xyz, count)

struct Prep { float centre[3], minP[3], maxP[3]; };

struct Node {
    uint32_t start, end;
    uint32_t mn[3], mx[3];      // ordered keys of the bounds
    uint32_t child_l, child_r, depth;
    uint32_t active;            // splits are attempted this level
    uint32_t axis; float pos;   // chosen plane
    uint32_t first_false;       // first slot whose triangle stays right (centre <= pos)
    uint32_t slot;              // index into the level's bin storage
};

// order-preserving map float -> uint (for atomicMin / atomicMax)
__host__ __device__ inline uint32_t fkey(float f) { uint32_t b; memcpy(&b, &f, 4); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__host__ __device__ inline float funkey(uint32_t k) { uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float f; memcpy(&f, &b, 4); return f; }

struct Level { uint32_t begin, end; };   // node ids of the level being split

// ---- kernels ------------------------------------------------------------------------------------

// per node of the level: decide whether it is split, plane positions, clear bins
__global__ void k_level_setup(Node* nodes, Level lv, float* planes, uint32_t* bins, uint32_t* counters)
{
    const uint32_t i = lv.begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lv.end) return;
    Node& n = nodes[i];
    n.first_false = 0xFFFFFFFFu; n.axis = 0; n.pos = 0.0f; n.slot = kNoNode;
    n.active = (n.end - n.start > (uint32_t)kMaxLeafTris && n.depth < (uint32_t)kMaxDepth) ? 1u : 0u;   // Geometry.h:394, 435
    if (!n.active) return;
    const uint32_t s = atomicAdd(&counters[0], 1u);
    n.slot = s;
    atomicMax(&counters[1], n.end - n.start);
    const float invBins = 1.0f / (float)kBins;                                                              // Geometry.h:307
    for (int axis = 0; axis < 3; axis++) {
        const float b0 = funkey(n.mn[axis]), b1 = funkey(n.mx[axis]);
        for (int k = 0; k < kBins; k++) { float splitT = (k + 1) * invBins; planes[(size_t)s * 45 + axis * kBins + k] = b0 + (b1 - b0) * splitT; }   // Geometry.h:411-413
    }
    uint32_t* b = bins + (size_t)s * kBinWords;
    for (int a = 0; a < 3; a++) for (int k = 0; k <= kBins; k++) {
        uint32_t* e = b + (a * (kBins + 1) + k) * 7;
        e[0] = e[1] = e[2] = fkey(FLT_MAX); e[3] = e[4] = e[5] = fkey(-FLT_MAX); e[6] = 0;                 // AABB::reset, Geometry.h:176
    }
}

__device__ inline void bin_add(uint32_t* e, const Prep& t)
{
    // two AABB::extend calls (maxP, minP), Geometry.h:329-330
    for (int a = 0; a < 3; a++) {
        atomicMin(&e[a], min(fkey(t.minP[a]), fkey(t.maxP[a])));
        atomicMax(&e[3 + a], max(fkey(t.minP[a]), fkey(t.maxP[a])));
    }
    atomicAdd(&e[6], 1u);
}

// per slot: drop the triangle into its bin on each axis. A workgroup walks kChunks consecutive
// chunks of 256 slots; while they all belong to one node it collects in LDS and flushes once (the
// top levels would otherwise queue on a handful of addresses), mixed chunks go straight to memory.
constexpr int kChunks = 16;
__global__ __launch_bounds__(kBlk) void k_bin(const Node* __restrict__ nodes, const uint32_t* __restrict__ slot_node, const uint32_t* __restrict__ idx,
                                              const Prep* __restrict__ prep, const float* __restrict__ planes, uint32_t* bins, uint32_t n)
{
    __shared__ uint32_t s_bins[kBinWords];
    __shared__ uint32_t s_first, s_uniform, s_held;      // s_held: node whose bins are in LDS (kNoNode: none)
    auto clear = [&]() { for (int k = threadIdx.x; k < kBinWords; k += kBlk) { const int f = k % 7; s_bins[k] = f < 3 ? fkey(FLT_MAX) : f < 6 ? fkey(-FLT_MAX) : 0u; } };
    auto flush = [&](uint32_t node) {
        uint32_t* dst = bins + (size_t)nodes[node].slot * kBinWords;
        for (int k = threadIdx.x; k < kBinWords; k += kBlk) {
            const int f = k % 7; const uint32_t v = s_bins[k];
            if (f < 3) { if (v != fkey(FLT_MAX)) atomicMin(&dst[k], v); }
            else if (f < 6) { if (v != fkey(-FLT_MAX)) atomicMax(&dst[k], v); }
            else if (v) atomicAdd(&dst[k], v);
        }
    };
    if (threadIdx.x == 0) s_held = kNoNode;
    clear();
    __syncthreads();
    for (int c = 0; c < kChunks; c++) {
        const uint32_t base = (blockIdx.x * (uint32_t)kChunks + (uint32_t)c) * kBlk;
        if (base >= n) break;
        const uint32_t i = base + threadIdx.x;
        const uint32_t node = i < n ? slot_node[i] : kNoNode;
        if (threadIdx.x == 0) { s_first = node; s_uniform = 1; }
        __syncthreads();
        if (i < n && node != s_first) s_uniform = 0;
        __syncthreads();
        const bool uniform = s_uniform != 0 && s_first != kNoNode && nodes[s_first].active;
        const uint32_t held = s_held;
        __syncthreads();                                             // everyone has read s_held before it changes below
        if (held != kNoNode && !(uniform && s_first == held)) {      // the LDS bins belong to another node: write them out
            flush(held);
            __syncthreads();
            clear();
            if (threadIdx.x == 0) s_held = kNoNode;
            __syncthreads();
        }
        if (uniform && threadIdx.x == 0) s_held = s_first;
        const bool live = i < n && node != kNoNode && nodes[node].active;
        if (live) {
            const Node& nd = nodes[node];
            const Prep t = prep[idx[i]];
            const float* pl = planes + (size_t)nd.slot * 45;
            for (int a = 0; a < 3; a++) {
                const float cc = t.centre[a];
                int k = 0;
                while (k < kBins && !(cc < pl[a * kBins + k])) k++;                                          // Geometry.h:327
                if (uniform) bin_add(s_bins + (a * (kBins + 1) + k) * 7, t);
                else bin_add(bins + (size_t)nd.slot * kBinWords + (a * (kBins + 1) + k) * 7, t);
            }
        }
        __syncthreads();
    }
    if (s_held != kNoNode) flush(s_held);
}

struct Box { float mn[3], mx[3]; uint32_t n; };
__device__ inline void box_reset(Box& b) { for (int a = 0; a < 3; a++) { b.mn[a] = FLT_MAX; b.mx[a] = -FLT_MAX; } b.n = 0; }
__device__ inline void box_merge(Box& b, const uint32_t* e)
{
    for (int a = 0; a < 3; a++) { const float lo = funkey(e[a]), hi = funkey(e[3 + a]); b.mx[a] = b.mx[a] > hi ? b.mx[a] : hi; b.mn[a] = b.mn[a] < lo ? b.mn[a] : lo; }
    b.n += e[6];
}
__device__ inline float box_area(const Box& b)                                                             // AABB::area, Geometry.h:231-235
{
    float sx = b.mx[0] - b.mn[0], sy = b.mx[1] - b.mn[1], sz = b.mx[2] - b.mn[2];
    return ((sx * sy) + (sy * sz) + (sx * sz)) * 2.0f;
}

// per active node: splitNode's search, Geometry.h:397-423
__global__ void k_best(Node* nodes, Level lv, const float* __restrict__ planes, const uint32_t* __restrict__ bins)
{
    const uint32_t i = lv.begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lv.end || !nodes[i].active) return;
    Node& n = nodes[i];
    const uint32_t* b = bins + (size_t)n.slot * kBinWords;
    const float* pl = planes + (size_t)n.slot * 45;
    Box parent; for (int a = 0; a < 3; a++) { parent.mn[a] = funkey(n.mn[a]); parent.mx[a] = funkey(n.mx[a]); }
    const float parentArea = box_area(parent);
    uint32_t bestAxis = 0; float bestPos = 0, bestCost = FLT_MAX;
    for (int axis = 0; axis < 3; axis++) {
        Box left; box_reset(left);
        for (int p = 0; p < kBins; p++) {
            box_merge(left, b + (axis * (kBins + 1) + p) * 7);
            Box right; box_reset(right);
            for (int k = kBins; k > p; k--) box_merge(right, b + (axis * (kBins + 1) + k) * 7);
            float cost = 1.0f + 2.0f * (box_area(left) * left.n + box_area(right) * right.n) / parentArea;   // calcCost, Geometry.h:309-312
            if (cost < bestCost) { bestCost = cost; bestPos = pl[axis * kBins + p]; bestAxis = axis; }
        }
    }
    n.axis = bestAxis; n.pos = bestPos;
}

// per slot: 1 when createSplit moves the triangle to the front (Geometry.h:351-352)
__global__ __launch_bounds__(kBlk) void k_flags(Node* nodes, const uint32_t* __restrict__ slot_node, const uint32_t* __restrict__ idx, const Prep* __restrict__ prep,
                                                uint32_t* flags, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
    if (i > n) return;
    if (i == n) { flags[i] = 0; return; }
    const uint32_t node = slot_node[i];
    uint32_t f = 0;
    if (node != kNoNode && nodes[node].active) {
        f = prep[idx[i]].centre[nodes[node].axis] > nodes[node].pos ? 1u : 0u;
    }
    flags[i] = f;
}

// per slot: the first "false" of a node is the one slot that is false while everything before it in
// the node is true — exactly one writer per node, no atomics
__global__ __launch_bounds__(kBlk) void k_first_false(Node* nodes, const uint32_t* __restrict__ slot_node, const uint32_t* __restrict__ flags,
                                                      const uint32_t* __restrict__ tpre, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= n || flags[i]) return;
    const uint32_t node = slot_node[i];
    if (node == kNoNode || !nodes[node].active) return;
    const uint32_t start = nodes[node].start;
    if (tpre[i] - tpre[start] == i - start) nodes[node].first_false = i;
}

// exclusive prefix sum over n + 1 flags: per-block scan, scan of the block totals, add back
__global__ __launch_bounds__(kBlk) void k_scan_blocks(const uint32_t* __restrict__ in, uint32_t* out, uint32_t* block_sums, uint32_t n)
{
    __shared__ uint32_t s[kBlk];
    const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
    const uint32_t v = i < n ? in[i] : 0u;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < kBlk; o <<= 1) { uint32_t t = threadIdx.x >= (uint32_t)o ? s[threadIdx.x - o] : 0u; __syncthreads(); s[threadIdx.x] += t; __syncthreads(); }
    if (i < n) out[i] = s[threadIdx.x] - v;
    if (threadIdx.x == kBlk - 1) block_sums[blockIdx.x] = s[kBlk - 1];
}
__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t* sums, uint32_t count)
{
    __shared__ uint32_t s[1024];
    const uint32_t per = (count + 1023u) / 1024u, b = threadIdx.x * per;
    uint32_t t = 0;
    for (uint32_t k = 0; k < per && b + k < count; k++) t += sums[b + k];
    s[threadIdx.x] = t;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { uint32_t v = threadIdx.x >= (uint32_t)o ? s[threadIdx.x - o] : 0u; __syncthreads(); s[threadIdx.x] += v; __syncthreads(); }
    uint32_t run = s[threadIdx.x] - t;
    for (uint32_t k = 0; k < per && b + k < count; k++) { const uint32_t v = sums[b + k]; sums[b + k] = run; run += v; }
}
__global__ __launch_bounds__(kBlk) void k_scan_add(uint32_t* out, const uint32_t* __restrict__ block_sums, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
    if (i < n) out[i] += block_sums[blockIdx.x];
}

// per slot: where does the tape cell of this slot get its triangle from?
__global__ __launch_bounds__(kBlk) void k_links(const Node* __restrict__ nodes, const uint32_t* __restrict__ slot_node, const uint32_t* __restrict__ flags,
                                                const uint32_t* __restrict__ tpre, uint32_t* link, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= n) return;
    const uint32_t node = slot_node[i];
    uint32_t l = i;
    if (node != kNoNode && nodes[node].active) {
        const uint32_t ff = nodes[node].first_false;
        // a "true" after the first "false" is rotation number (trues in [ff, i)): a copy of that cell
        if (ff != 0xFFFFFFFFu && i > ff && flags[i]) l = ff + (tpre[i] - tpre[ff]);
    }
    link[i] = l;
}
__global__ __launch_bounds__(kBlk) void k_jump(const uint32_t* __restrict__ in, uint32_t* out, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
    if (i < n) out[i] = in[in[i]];
}

// per slot: final place of createSplit's pass
__global__ __launch_bounds__(kBlk) void k_scatter(const Node* __restrict__ nodes, const uint32_t* __restrict__ slot_node, const uint32_t* __restrict__ flags,
                                                  const uint32_t* __restrict__ tpre, const uint32_t* __restrict__ link, const uint32_t* __restrict__ idx_in,
                                                  uint32_t* idx_out, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
    if (i >= n) return;
    const uint32_t node = slot_node[i];
    if (node == kNoNode || !nodes[node].active) { idx_out[i] = idx_in[i]; return; }
    const Node& nd = nodes[node];
    const uint32_t ff = nd.first_false;
    const uint32_t nTrue = tpre[nd.end] - tpre[nd.start], rStart = nd.start + nTrue;
    if (ff == 0xFFFFFFFFu) { idx_out[i] = idx_in[i]; return; }                      // every triangle moves "to the front": self-swaps only
    if (i < ff) { idx_out[i] = idx_in[i]; return; }                                 // leading trues stay where they are
    const uint32_t rotations = tpre[nd.end] - tpre[ff];                             // trues after the first false
    if (flags[i]) idx_out[nd.start + (tpre[i] - tpre[nd.start])] = idx_in[i];       // stable among the trues
    const uint32_t cell = i - ff;                                                   // this slot's tape cell
    if (cell >= rotations) idx_out[rStart + (cell - rotations)] = idx_in[link[i]];  // cells that survive form the queue
}

// per active node: create the children (Geometry.h:360-388)
__global__ void k_children(Node* nodes, Level lv, const uint32_t* __restrict__ tpre, uint32_t* counters)
{
    const uint32_t i = lv.begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lv.end || !nodes[i].active) return;
    Node& n = nodes[i];
    const uint32_t rStart = n.start + (tpre[n.end] - tpre[n.start]);
    if (rStart == n.start || rStart == n.end) { n.active = 0; return; }             // no split: stays a leaf
    const uint32_t c = lv.end + 2u * atomicAdd(&counters[2], 1u);
    n.child_l = c; n.child_r = c + 1;
    for (int k = 0; k < 2; k++) {
        Node& ch = nodes[c + k];
        ch.start = k == 0 ? n.start : rStart; ch.end = k == 0 ? rStart : n.end;
        for (int a = 0; a < 3; a++) { ch.mn[a] = fkey(FLT_MAX); ch.mx[a] = fkey(-FLT_MAX); }
        ch.child_l = ch.child_r = 0; ch.depth = n.depth + 1; ch.active = 0; ch.slot = kNoNode;
    }
}

// per slot: move to the child node and extend its box (Geometry.h:369-378)
__global__ __launch_bounds__(kBlk) void k_child_bounds(Node* nodes, uint32_t* slot_node, const uint32_t* __restrict__ idx, const Prep* __restrict__ prep, uint32_t n)
{
    __shared__ uint32_t s_first, s_uniform;
    __shared__ uint32_t s_box[6];
    const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
    uint32_t child = kNoNode;
    if (i < n) {
        const uint32_t node = slot_node[i];
        if (node != kNoNode) {
            if (nodes[node].active) { child = i < nodes[nodes[node].child_r].start ? nodes[node].child_l : nodes[node].child_r; }
            slot_node[i] = nodes[node].active ? child : kNoNode;                    // leaves are finished
        }
    }
    if (threadIdx.x == 0) { s_first = child; s_uniform = 1; for (int a = 0; a < 3; a++) { s_box[a] = fkey(FLT_MAX); s_box[3 + a] = fkey(-FLT_MAX); } }
    __syncthreads();
    if (i < n && child != s_first) s_uniform = 0;
    __syncthreads();
    const bool uniform = s_uniform != 0;
    if (child != kNoNode) {
        const Prep t = prep[idx[i]];
        uint32_t* mn = uniform ? s_box : nodes[child].mn; uint32_t* mx = uniform ? s_box + 3 : nodes[child].mx;
        for (int a = 0; a < 3; a++) { atomicMin(&mn[a], min(fkey(t.minP[a]), fkey(t.maxP[a]))); atomicMax(&mx[a], max(fkey(t.minP[a]), fkey(t.maxP[a]))); }
    }
    __syncthreads();
    if (uniform && s_first != kNoNode && threadIdx.x < 6) {
        if (threadIdx.x < 3) atomicMin(&nodes[s_first].mn[threadIdx.x], s_box[threadIdx.x]);
        else atomicMax(&nodes[s_first].mx[threadIdx.x - 3], s_box[threadIdx.x]);
    }
}

__global__ __launch_bounds__(kBlk) void k_fill(uint32_t* p, uint32_t v, uint32_t n, bool iota)
{
    const uint32_t i = blockIdx.x * kBlk + threadIdx.x;
    if (i < n) p[i] = iota ? i : v;
}

} // namespace devbuild

// Builds the tree of build_bvh2() on the device behind `stream`. Returns hipSuccess or the
// first HIP error; `out` is complete only on success.
inline hipError_t build_bvh2_device(const std::vector<TriPrep>& tris, const float smin[3], const float smax[3], Bvh2& out, hipStream_t stream)
{
    using namespace devbuild;
    const uint32_t n = (uint32_t)tris.size();
    out.nodes.clear(); out.indices.resize(n); out.depth = 0;
    Bvh2Node root{};
    for (int a = 0; a < 3; a++) { root.bmin[a] = smin[a]; root.bmax[a] = smax[a]; }
    root.start = 0; root.end = n;
    if (n <= (uint32_t)kMaxLeafTris) { for (uint32_t i = 0; i < n; i++) out.indices[i] = i; out.nodes.push_back(root); return hipSuccess; }

#define RTX_DB(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)
    const size_t maxNodes = 2 * (size_t)n + 2, maxActive = (size_t)n / (kMaxLeafTris + 1) + 2;
    DevBuf b_prep, b_idx[2], b_slot, b_flags, b_tpre, b_sums, b_link[2], b_nodes, b_planes, b_bins, b_cnt;
    RTX_DB(b_prep.ensure((size_t)n * sizeof(Prep))); RTX_DB(b_idx[0].ensure((size_t)n * 4)); RTX_DB(b_idx[1].ensure((size_t)n * 4));
    RTX_DB(b_slot.ensure((size_t)n * 4)); RTX_DB(b_flags.ensure(((size_t)n + 1) * 4)); RTX_DB(b_tpre.ensure(((size_t)n + 1) * 4));
    const uint32_t scanBlocks = (n + 1 + kBlk - 1) / kBlk;
    RTX_DB(b_sums.ensure((size_t)scanBlocks * 4)); RTX_DB(b_link[0].ensure((size_t)n * 4)); RTX_DB(b_link[1].ensure((size_t)n * 4));
    RTX_DB(b_nodes.ensure(maxNodes * sizeof(Node))); RTX_DB(b_planes.ensure(maxActive * 45 * 4)); RTX_DB(b_bins.ensure(maxActive * kBinWords * 4));
    RTX_DB(b_cnt.ensure(16));

    std::vector<Prep> hp(n);
    for (uint32_t i = 0; i < n; i++) for (int a = 0; a < 3; a++) { hp[i].centre[a] = tris[i].centre[a]; hp[i].minP[a] = tris[i].minP[a]; hp[i].maxP[a] = tris[i].maxP[a]; }
    RTX_DB(hipMemcpyAsync(b_prep.p, hp.data(), (size_t)n * sizeof(Prep), hipMemcpyHostToDevice, stream));
    Node hroot{}; hroot.start = 0; hroot.end = n;
    for (int a = 0; a < 3; a++) { hroot.mn[a] = fkey(smin[a]); hroot.mx[a] = fkey(smax[a]); }
    RTX_DB(hipMemcpyAsync(b_nodes.p, &hroot, sizeof(Node), hipMemcpyHostToDevice, stream));
    const dim3 blk(kBlk), gridN((n + kBlk - 1) / kBlk), gridN1(scanBlocks);
    hipLaunchKernelGGL(k_fill, gridN, blk, 0, stream, b_idx[0].as<uint32_t>(), 0u, n, true);
    hipLaunchKernelGGL(k_fill, gridN, blk, 0, stream, b_slot.as<uint32_t>(), 0u, n, false);

    Node* nodes = b_nodes.as<Node>();
    Level lv{0, 1};
    int cur = 0;
    for (int level = 0; level <= kMaxDepth; level++) {
        const uint32_t count = lv.end - lv.begin;
        const dim3 gridL((count + 127) / 128), blkL(128);
        uint32_t* cnt = b_cnt.as<uint32_t>();
        RTX_DB(hipMemsetAsync(cnt, 0, 16, stream));
        hipLaunchKernelGGL(k_level_setup, gridL, blkL, 0, stream, nodes, lv, b_planes.as<float>(), b_bins.as<uint32_t>(), cnt);
        uint32_t hc[4];
        RTX_DB(hipMemcpyAsync(hc, cnt, 16, hipMemcpyDeviceToHost, stream));
        RTX_DB(hipStreamSynchronize(stream));
        if (hc[0] == 0) break;                                                      // nothing left to split
        uint32_t* idx_in = b_idx[cur].as<uint32_t>(); uint32_t* idx_out = b_idx[cur ^ 1].as<uint32_t>();
        hipLaunchKernelGGL(k_bin, dim3((n + kBlk * kChunks - 1) / (kBlk * kChunks)), blk, 0, stream, nodes, b_slot.as<uint32_t>(), idx_in, b_prep.as<Prep>(), b_planes.as<float>(), b_bins.as<uint32_t>(), n);
        hipLaunchKernelGGL(k_best, gridL, blkL, 0, stream, nodes, lv, b_planes.as<float>(), b_bins.as<uint32_t>());
        hipLaunchKernelGGL(k_flags, gridN1, blk, 0, stream, nodes, b_slot.as<uint32_t>(), idx_in, b_prep.as<Prep>(), b_flags.as<uint32_t>(), n);
        hipLaunchKernelGGL(k_scan_blocks, gridN1, blk, 0, stream, b_flags.as<uint32_t>(), b_tpre.as<uint32_t>(), b_sums.as<uint32_t>(), n + 1);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, b_sums.as<uint32_t>(), scanBlocks);
        hipLaunchKernelGGL(k_scan_add, gridN1, blk, 0, stream, b_tpre.as<uint32_t>(), b_sums.as<uint32_t>(), n + 1);
        hipLaunchKernelGGL(k_first_false, gridN, blk, 0, stream, nodes, b_slot.as<uint32_t>(), b_flags.as<uint32_t>(), b_tpre.as<uint32_t>(), n);
        hipLaunchKernelGGL(k_links, gridN, blk, 0, stream, nodes, b_slot.as<uint32_t>(), b_flags.as<uint32_t>(), b_tpre.as<uint32_t>(), b_link[0].as<uint32_t>(), n);
        int lk = 0;
        for (uint32_t span = 1; span < hc[1]; span <<= 1) {                         // chains are shorter than the largest node
            hipLaunchKernelGGL(k_jump, gridN, blk, 0, stream, b_link[lk].as<uint32_t>(), b_link[lk ^ 1].as<uint32_t>(), n);
            lk ^= 1;
        }
        hipLaunchKernelGGL(k_scatter, gridN, blk, 0, stream, nodes, b_slot.as<uint32_t>(), b_flags.as<uint32_t>(), b_tpre.as<uint32_t>(), b_link[lk].as<uint32_t>(), idx_in, idx_out, n);
        hipLaunchKernelGGL(k_children, gridL, blkL, 0, stream, nodes, lv, b_tpre.as<uint32_t>(), cnt);
        hipLaunchKernelGGL(k_child_bounds, gridN, blk, 0, stream, nodes, b_slot.as<uint32_t>(), idx_out, b_prep.as<Prep>(), n);
        RTX_DB(hipMemcpyAsync(hc, cnt, 16, hipMemcpyDeviceToHost, stream));
        RTX_DB(hipStreamSynchronize(stream));
        cur ^= 1;
        if (hc[2] == 0) break;
        lv = Level{lv.end, lv.end + 2 * hc[2]};
        if ((size_t)lv.end > maxNodes) return hipErrorOutOfMemory;
    }
    RTX_DB(hipGetLastError());
    std::vector<Node> hn(lv.end);
    RTX_DB(hipMemcpyAsync(hn.data(), nodes, (size_t)lv.end * sizeof(Node), hipMemcpyDeviceToHost, stream));
    RTX_DB(hipMemcpyAsync(out.indices.data(), b_idx[cur].p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    RTX_DB(hipStreamSynchronize(stream));
#undef RTX_DB

    // renumber into the order split() creates nodes: children side by side, left subtree, right subtree
    out.nodes.reserve(hn.size());
    auto emit = [&](const Node& s) {
        Bvh2Node d{};
        for (int a = 0; a < 3; a++) { d.bmin[a] = funkey(s.mn[a]); d.bmax[a] = funkey(s.mx[a]); }
        d.start = s.start; d.end = s.end; d.child_l = 0; d.child_r = 0;
        out.nodes.push_back(d); return (uint32_t)out.nodes.size() - 1;
    };
    emit(hn[0]);
    for (int a = 0; a < 3; a++) { out.nodes[0].bmin[a] = smin[a]; out.nodes[0].bmax[a] = smax[a]; }
    struct Item { uint32_t src, dst; };
    std::vector<Item> stack; stack.push_back({0, 0});
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        const Node& s = hn[it.src];
        out.depth = std::max(out.depth, s.depth);
        if (s.child_l == 0 && s.child_r == 0) continue;
        const uint32_t l = emit(hn[s.child_l]), r = emit(hn[s.child_r]);
        out.nodes[it.dst].child_l = l; out.nodes[it.dst].child_r = r;
        stack.push_back({s.child_r, r}); stack.push_back({s.child_l, l});           // left subtree is numbered first
    }
    return hipSuccess;
}

} // namespace rtx
