// bvh_host.cpp — see bvh_host.h.
#include "bvh_host.h"

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <cstdlib>
#include <deque>
#include <functional>
#include <thread>

namespace rtx {

namespace {

constexpr int kMaxLeafTris = 8;   // MAXNODE_TRIANGLES, RTBase/Geometry.h:285
constexpr int kBins = 15;         // BUILD_BINS, RTBase/Geometry.h:288
constexpr int kMaxDepth = 50;     // BVHTree::maxDepth, RTBase/Geometry.h:306

// Max / Min of RTBase/Core.h:199-207 pick the SECOND operand unless the first compares
// strictly greater / smaller; kept so that bounds agree with the reference bit for bit.
inline float refMax(float a, float b) { return a > b ? a : b; }
inline float refMin(float a, float b) { return a < b ? a : b; }

struct GrowBox {
    float mn[3], mx[3];
    uint32_t n;
    void reset() { for (int a = 0; a < 3; a++) { mn[a] = FLT_MAX; mx[a] = -FLT_MAX; } n = 0; }  // AABB::reset, Geometry.h:176
    void add(const TriPrep& t)                                                                    // two AABB::extend calls, Geometry.h:329-330
    {
        for (int a = 0; a < 3; a++) {
            mx[a] = refMax(refMax(mx[a], t.maxP[a]), t.minP[a]);
            mn[a] = refMin(refMin(mn[a], t.maxP[a]), t.minP[a]);
        }
        n++;
    }
    void merge(const GrowBox& o)
    {
        for (int a = 0; a < 3; a++) { mx[a] = refMax(mx[a], o.mx[a]); mn[a] = refMin(mn[a], o.mn[a]); }
        n += o.n;
    }
    float area() const                                                                            // AABB::area, Geometry.h:231-235
    {
        float sx = mx[0] - mn[0], sy = mx[1] - mn[1], sz = mx[2] - mn[2];
        return ((sx * sy) + (sy * sz) + (sx * sz)) * 2.0f;
    }
};

inline float nodeArea(const Bvh2Node& n)
{
    float sx = n.bmax[0] - n.bmin[0], sy = n.bmax[1] - n.bmin[1], sz = n.bmax[2] - n.bmin[2];
    return ((sx * sy) + (sy * sz) + (sx * sz)) * 2.0f;
}

// Builds into a node vector of its own (the subtree's root at index 0) over a range of the
// shared index array; subtrees own disjoint index ranges, so they can be built concurrently.
struct SubTree { std::vector<Bvh2Node> nodes; std::vector<uint32_t>& indices; };

struct Builder {
    const TriPrep* tris;
    SubTree& out;
    const float invBins = 1.0f / (float)kBins;   // Geometry.h:307

    // splitNode + createSplit, Geometry.h:345-430. The reference evaluates each of the
    // 3 x 15 candidate planes with a full sweep that sorts triangles by `centre < pos`.
    // The plane positions of one axis are non-decreasing, so a triangle belongs to the
    // "left" set of plane i exactly when i >= k, k = first plane with centre < pos[k];
    // binning by k and merging prefixes / suffixes gives the same boxes and counts, hence
    // the same float costs, with one sweep.
    bool splitNode(uint32_t node)
    {
        const uint32_t start = out.nodes[node].start, end = out.nodes[node].end;
        if (end - start <= (uint32_t)kMaxLeafTris) return false;

        float pos[3][kBins];
        for (int axis = 0; axis < 3; axis++) {
            float b0 = out.nodes[node].bmin[axis], b1 = out.nodes[node].bmax[axis];
            for (int i = 0; i < kBins; i++) {
                float splitT = (i + 1) * invBins;
                pos[axis][i] = b0 + (b1 - b0) * splitT;
            }
        }
        GrowBox bins[3][kBins + 1];
        for (int axis = 0; axis < 3; axis++) for (int k = 0; k <= kBins; k++) bins[axis][k].reset();
        for (uint32_t i = start; i < end; i++) {
            const TriPrep& t = tris[out.indices[i]];
            for (int axis = 0; axis < 3; axis++) {
                const float c = t.centre[axis];
                int k = 0;
                while (k < kBins && !(c < pos[axis][k])) k++;
                bins[axis][k].add(t);
            }
        }
        const float parentArea = nodeArea(out.nodes[node]);
        uint32_t bestAxis = 0; float bestPos = 0, bestCost = FLT_MAX;
        for (int axis = 0; axis < 3; axis++) {
            GrowBox suffix[kBins + 2];
            suffix[kBins + 1].reset();
            for (int k = kBins; k >= 0; k--) { suffix[k] = suffix[k + 1]; suffix[k].merge(bins[axis][k]); }
            GrowBox left; left.reset();
            for (int i = 0; i < kBins; i++) {
                left.merge(bins[axis][i]);
                const GrowBox& right = suffix[i + 1];
                // calcCost, Geometry.h:309-312
                float cost = 1.0f + 2.0f * (left.area() * left.n + right.area() * right.n) / parentArea;
                if (cost < bestCost) { bestCost = cost; bestPos = pos[axis][i]; bestAxis = axis; }
            }
        }
        // createSplit, Geometry.h:345-389
        uint32_t rStart = start;
        for (uint32_t i = start; i < end; i++) {
            if (tris[out.indices[i]].centre[bestAxis] > bestPos) { std::swap(out.indices[i], out.indices[rStart]); rStart++; }
        }
        if (rStart == start || rStart == end) return false;
        GrowBox l, r; l.reset(); r.reset();
        for (uint32_t i = start; i < rStart; i++) l.add(tris[out.indices[i]]);
        for (uint32_t i = rStart; i < end; i++) r.add(tris[out.indices[i]]);
        Bvh2Node cl{}, cr{};
        for (int a = 0; a < 3; a++) { cl.bmin[a] = l.mn[a]; cl.bmax[a] = l.mx[a]; cr.bmin[a] = r.mn[a]; cr.bmax[a] = r.mx[a]; }
        cl.start = start; cl.end = rStart; cr.start = rStart; cr.end = end;
        out.nodes.push_back(cl); out.nodes.push_back(cr);
        out.nodes[node].child_l = (uint32_t)out.nodes.size() - 2;
        out.nodes[node].child_r = (uint32_t)out.nodes.size() - 1;
        return true;
    }
    uint32_t split(uint32_t node, int depth)                                                       // Geometry.h:432-449
    {
        if (depth >= kMaxDepth) return depth;
        if (!splitNode(node)) return depth;
        depth++;
        const uint32_t cl = out.nodes[node].child_l, cr = out.nodes[node].child_r;
        uint32_t d1 = split(cl, depth);
        uint32_t d2 = split(cr, depth);
        return std::max(d1, d2);
    }
};

} // namespace

// fn(begin, end) over [0, n) on up to 16 threads (RTX_BUILD_THREADS overrides)
void parallel_ranges(uint64_t n, const std::function<void(uint64_t, uint64_t)>& fn)
{
    int threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("RTX_BUILD_THREADS")) threads = std::max(1, atoi(e));
    if (n < 100000 || threads == 1) { fn((uint64_t)0, n); return; }
    std::vector<std::thread> pool;
    const uint64_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) { const uint64_t b = t * per, e = std::min(n, b + per); if (b < e) pool.emplace_back([=]() { fn(b, e); }); }
    for (auto& th : pool) th.join();
}

void prepare_triangles(const float* p, uint64_t n, std::vector<TriPrep>& out, float smin[3], float smax[3])
{
    out.resize(n);
    parallel_ranges(n, [&](uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; i++) {
            const float* v0 = p + i * 9; const float* v1 = v0 + 3; const float* v2 = v0 + 6;
            TriPrep& t = out[i];
            for (int a = 0; a < 3; a++) {
                t.e1[a] = v0[a] - v2[a];
                t.e2[a] = v1[a] - v2[a];
                t.v2[a] = v2[a];
                t.maxP[a] = refMax(v0[a], refMax(v1[a], v2[a]));
                t.minP[a] = refMin(v0[a], refMin(v1[a], v2[a]));
                t.centre[a] = t.minP[a] + (t.maxP[a] - t.minP[a]) * 0.5f;
            }
        }
    });
    // Scene::init extends the scene box by the three vertices of every triangle in turn, Scene.h:156-158
    // (kept sequential: the sign of a zero bound depends on the order)
    for (int a = 0; a < 3; a++) { smin[a] = FLT_MAX; smax[a] = -FLT_MAX; }
    for (uint64_t i = 0; i < n; i++) {
        const float* v0 = p + i * 9; const float* v1 = v0 + 3; const float* v2 = v0 + 6;
        for (int a = 0; a < 3; a++) {
            smax[a] = refMax(refMax(refMax(smax[a], v0[a]), v1[a]), v2[a]);
            smin[a] = refMin(refMin(refMin(smin[a], v0[a]), v1[a]), v2[a]);
        }
    }
}

namespace {

// The reference numbers nodes in the order split() creates them (Geometry.h:381-386, 444-445):
// a node's two children side by side, then everything below the left child, then everything
// below the right child. A subtree built on its own therefore only needs its ids shifted.
uint32_t build_subtree(const TriPrep* tris, SubTree& st, int depth, int threads)
{
    Builder b{tris, st};
    const uint32_t count = st.nodes[0].end - st.nodes[0].start;
    if (threads <= 1 || count < 50000u) return b.split(0, depth);
    if (depth >= kMaxDepth) return (uint32_t)depth;
    if (!b.splitNode(0)) return (uint32_t)depth;
    depth++;
    SubTree left{{st.nodes[1]}, st.indices}, right{{st.nodes[2]}, st.indices};
    uint32_t dl = 0, dr = 0;
    const int tl = threads / 2, tr = threads - tl;
    std::thread worker([&]() { dl = build_subtree(tris, left, depth, tl); });
    dr = build_subtree(tris, right, depth, tr);
    worker.join();
    auto place = [&](const std::vector<Bvh2Node>& sub, uint32_t self, uint32_t shift) {
        auto moved = [&](Bvh2Node n) { if (n.child_l != 0 || n.child_r != 0) { n.child_l += shift; n.child_r += shift; } return n; };
        st.nodes[self] = moved(sub[0]);
        for (size_t j = 1; j < sub.size(); j++) st.nodes.push_back(moved(sub[j]));
    };
    st.nodes.reserve(3 + left.nodes.size() + right.nodes.size());
    place(left.nodes, 1, 2);                                   // local id j >= 1 of the left block lands at 3 + (j - 1)
    place(right.nodes, 2, (uint32_t)left.nodes.size() + 1);    // ... of the right block behind the left block
    return std::max(dl, dr);
}

} // namespace

void build_bvh2(const std::vector<TriPrep>& tris, const float smin[3], const float smax[3], Bvh2& out)
{
    out.nodes.clear();
    out.indices.resize(tris.size());
    for (uint32_t i = 0; i < tris.size(); i++) out.indices[i] = i;
    Bvh2Node root{};
    for (int a = 0; a < 3; a++) { root.bmin[a] = smin[a]; root.bmax[a] = smax[a]; }
    root.start = 0; root.end = (uint32_t)tris.size();
    int threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("RTX_BUILD_THREADS")) threads = std::max(1, atoi(e));
    SubTree st{{root}, out.indices};
    st.nodes.reserve(tris.size() / 2 + 16);
    out.depth = tris.empty() ? 0 : build_subtree(tris.data(), st, 0, threads);
    out.nodes = std::move(st.nodes);
}

void flatten_bvh4(const Bvh2& in, Bvh4& out)
{
    const auto& N = in.nodes;
    const size_t nTris = in.indices.size();
    out.nodes.clear(); out.nodes2.clear(); out.order.assign(nTris, 0); out.depth = 0;
    for (int a = 0; a < 3; a++) { out.root_min[a] = N[0].bmin[a]; out.root_max[a] = N[0].bmax[a]; }
    auto isLeaf = [&](uint32_t i) { return N[i].child_l == 0 && N[i].child_r == 0; };

    // Leaf visit order of BVHTree::traverse (Geometry.h:512-553): child_l is pushed first,
    // so child_r is popped first — a right-first depth-first walk.
    std::vector<uint32_t> leafPos(N.size(), 0);
    {
        std::vector<uint32_t> stack; stack.push_back(0); uint32_t cursor = 0;
        while (!stack.empty()) {
            uint32_t i = stack.back(); stack.pop_back();
            if (isLeaf(i)) { leafPos[i] = cursor; for (uint32_t k = N[i].start; k < N[i].end; k++) out.order[cursor++] = in.indices[k]; }
            else { stack.push_back(N[i].child_l); stack.push_back(N[i].child_r); }
        }
    }
    out.nodes2.resize(N.size());
    for (size_t i = 0; i < N.size(); i++) {
        Node2Dev& d = out.nodes2[i];
        for (int a = 0; a < 3; a++) { d.bmin[a] = N[i].bmin[a]; d.bmax[a] = N[i].bmax[a]; }
        if (isLeaf((uint32_t)i)) { d.a = kLeafBit | leafPos[i]; d.b = N[i].end - N[i].start; }
        else { d.a = N[i].child_l; d.b = N[i].child_r; }
    }

    auto allocNode = [&]() {
        Node4 n; std::memset(&n, 0, sizeof n);
        for (int k = 0; k < 4; k++) n.child[k] = kEmptySlot;
        out.nodes.push_back(n); return (uint32_t)out.nodes.size() - 1;
    };
    auto setBox = [&](uint32_t node, int slot, const float* mn, const float* mx) {
        for (int a = 0; a < 3; a++) { out.nodes[node].lo[a][slot] = mn[a]; out.nodes[node].hi[a][slot] = mx[a]; }
    };
    // A reference leaf may hold any number of triangles (unsplittable sets, depth limit).
    // Ranges above 16 are cut into slots that all carry the leaf's own box, which keeps the
    // set of rays that reach each triangle unchanged.
    struct Chunk { uint32_t first, count; const float* mn; const float* mx; uint32_t node; uint32_t depth; };
    std::deque<Chunk> chunkQueue;
    auto leafRef = [&](uint32_t first, uint32_t count, const float* mn, const float* mx, uint32_t depth) -> uint32_t {
        if (count <= kLeafMaxTris) return kLeafBit | ((count - 1) << 27) | first;
        uint32_t idx = allocNode();
        chunkQueue.push_back({first, count, mn, mx, idx, depth + 1});
        return idx;
    };
    auto drainChunks = [&]() {
        while (!chunkQueue.empty()) {
            Chunk c = chunkQueue.front(); chunkQueue.pop_front();
            out.depth = std::max(out.depth, c.depth);
            uint32_t per = (c.count + 3) / 4, first = c.first, left = c.count;
            for (int k = 0; k < 4 && left > 0; k++) {
                uint32_t cnt = std::min(per, left);
                setBox(c.node, k, c.mn, c.mx);
                uint32_t ref = leafRef(first, cnt, c.mn, c.mx, c.depth);
                out.nodes[c.node].child[k] = ref;
                first += cnt; left -= cnt;
            }
        }
    };

    struct Work { uint32_t bnode, node4, depth; };
    std::deque<Work> queue;
    queue.push_back({0, allocNode(), 1});
    while (!queue.empty()) {
        Work w = queue.front(); queue.pop_front();
        out.depth = std::max(out.depth, w.depth);
        uint32_t slots[4]; int ns = 0;
        if (isLeaf(w.bnode)) slots[ns++] = w.bnode;
        else {
            slots[ns++] = N[w.bnode].child_l; slots[ns++] = N[w.bnode].child_r;
            while (ns < 4) {   // open the inner slot with the largest box
                int best = -1; float bestArea = -1.0f;
                for (int k = 0; k < ns; k++) if (!isLeaf(slots[k])) { float a = nodeArea(N[slots[k]]); if (best < 0 || a > bestArea) { best = k; bestArea = a; } }
                if (best < 0) break;
                uint32_t b = slots[best];
                slots[best] = N[b].child_l; slots[ns++] = N[b].child_r;
            }
        }
        for (int k = 0; k < ns; k++) {
            const Bvh2Node& s = N[slots[k]];
            setBox(w.node4, k, s.bmin, s.bmax);
            if (isLeaf(slots[k])) {
                uint32_t cnt = s.end - s.start;
                if (cnt == 0) { out.nodes[w.node4].child[k] = kEmptySlot; continue; }
                uint32_t ref = leafRef(leafPos[slots[k]], cnt, s.bmin, s.bmax, w.depth);
                out.nodes[w.node4].child[k] = ref;
            } else {
                uint32_t ci = allocNode();
                out.nodes[w.node4].child[k] = ci;
                queue.push_back({slots[k], ci, w.depth + 1});
            }
        }
        drainChunks();
    }
}

} // namespace rtx
