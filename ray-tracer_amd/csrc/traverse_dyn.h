// traverse_dyn.h — BVH traversal with dynamic work fetch, the schedule every ray-query kernel
// of the core runs on.
//
// One ray per lane, but a lane whose ray has finished does not idle until the slowest ray of
// its wave is done: as soon as REFILL lanes of the wave are idle they take the next items of
// the wave's reserved chunk (wave64 ballot; one atomic per chunk). Items are handed out in
// queue order, so the lanes that start together hold neighbouring rays, walk the same top
// nodes and hit the same cache lines. Inside the loop the wave alternates two phases:
// every lane descends inner nodes until it holds a leaf, then the wave tests leaf
// triangles, TRI_UNROLL at a time so that their loads are in flight together.
//
// What is traversed and what is returned is identical to traverse() in traverse.h (same
// candidates, same arithmetic, same tie-break); only the assignment of rays to lanes and
// the order of independent steps differ.
//
// A Policy supplies the items and consumes the results:
//   bool  fetch(uint32_t k, float o[3], float d[3], float& maxT)   load item k (k < n); false: the item needs no walk and is done
//                                                                   (the policy has recorded its result itself)
//   void  retire(uint32_t k, const HitRec& best, bool reached)     reached == false: the ray missed the root box
// MODE selects what a ray asks for: kModeClosest (closest hit), kModeAny (visibility: any hit with t <= maxT ends
// the ray) or kModeMixed, where every item says which of the two it is (fetch takes a further `bool& any`): one
// launch then serves the shadow rays of one path depth and the closest-hit rays of the next.
#pragma once

#include "traverse.h"

namespace rtx {

// Work counters of one traversal launch: kWorkSegments counters (one per lane of a wave,
// so that a wave can read them all with one load), kWorkStride words apart, all zero at
// launch.
constexpr int kWorkSegments = 64;
constexpr int kWorkStride = 32;
constexpr int kWorkMaskCopies = 8;      // after the counters: kWorkMaskCopies lines with the "told dry" words (chunk counters, item counters)
constexpr int kWorkSlotWords = (kWorkSegments + kWorkMaskCopies) * kWorkStride;

// The chunk that counter `seg` hands out as its j-th: chunk j * kWorkSegments + c, where the eight
// counters of one XCD (workgroup b runs on XCD b % 8 and draws from counter b % 64) own eight
// neighbouring chunks c, so that neighbouring rays meet in the same L2. Successive chunks of one
// counter lie kWorkSegments chunks apart: the rays a wave has in flight then come from distant
// places of the queue, which evens out their cost (neighbouring rays cost alike).
__device__ __forceinline__ uint32_t work_chunk(uint32_t j, uint32_t seg)
{
    return j * (uint32_t)kWorkSegments + (seg & 7u) * 8u + (seg >> 3);
}

// Tail sharing. When the work counters have run dry, the launch lasts as long as the longest
// ray still in flight (a few rays walk ten times the mean number of nodes), with most lanes
// of their waves idle. From then on an idle lane takes pending subtrees off the traversal
// stack of a busy lane of its wave, newest (nearest) first — the entries that lane would visit
// next itself, so the helpers work on the ray's likely critical path and a hit they find prunes
// the far entries — and walks them for that lane's ray. Per-lane mailboxes in LDS collect what the helpers find: `key` is the best candidate of
// the lane's ray so far, packed so that an unsigned minimum is the reference's rule (smaller t,
// then earlier triangle); `pending` counts helpers still at work. The ray retires when its own
// lane and all helpers are done. Candidates and arithmetic are unchanged, the minimum is
// order-independent, so results are the same as without sharing.
struct __attribute__((packed, aligned(4))) TriQuad { float f[36]; };

template <int BLOCK>
struct TailShare {
    unsigned long long key[BLOCK];
    uint32_t pending[BLOCK];
};

// mailbox words are written by other lanes of the wave: read them as atomics so that the load is neither hoisted nor merged
template <class T> __device__ __forceinline__ T lds_read(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__device__ __forceinline__ unsigned long long pack_hit(const HitRec& b) { return ((unsigned long long)__float_as_uint(b.t) << 32) | b.pos; }

// position of the n-th set bit of m (n < popcount(m))
__device__ __forceinline__ uint32_t nth_set_bit(unsigned long long m, uint32_t n)
{
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t w = 32; w > 0; w >>= 1) {
        const uint32_t c = (uint32_t)__popcll((m >> pos) & ((1ull << w) - 1ull));
        if (n >= c) { n -= c; pos += w; }
    }
    return pos;
}

constexpr int kModeClosest = 0, kModeAny = 1, kModeMixed = 2;

// LOCAL: the n items belong to the calling WAVE alone (n is wave-uniform and may differ between the waves of a workgroup;
// `work` is not used): lanes take them in order as they fall idle, and once they are handed out the idle lanes help (SHARE).
// The path tracer's tail kernel walks the few rays of a wave's own paths this way, depth after depth, without a launch.
template <class Policy, int MODE, bool COUNT, int REFILL, int BLOCK, int LDS_N, int TRI_UNROLL, int SCHED, int SHARE = 0, bool LOCAL = false>
__device__ __forceinline__ void traverse_dyn(const DeviceScene& sc, LaneStack<BLOCK, LDS_N>& stk, uint32_t n, uint32_t* work,
                                             Policy& pol, unsigned long long* counters, TailShare<BLOCK>* ts = nullptr)
{
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    // item values from kHelper up (except kNone): the lane helps the ray of lane (item & 63) of its wave
    constexpr uint32_t kHelper = 0xFFFFFF00u;
    const uint32_t wave0 = threadIdx.x & ~63u;
    bool tail = false;   // SHARE: the mailboxes are set up
    TravCount cnt{0, 0, 0}, mark{0, 0, 0};   // mark: the counters when the lane's current ray started (COUNT only)
    uint32_t wave_iters[3] = {0, 0, 0};   // COUNT only: inner-loop, triangle-loop and outer-loop trips of this lane
    uint32_t cur = kDone, item = kNone;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0, bound = FLT_MAX, maxT = 0;
    bool any_lane = MODE == kModeAny;      // kModeMixed: set per item
#define RTX_IS_ANY (MODE == kModeMixed ? any_lane : MODE == kModeAny)
    HitRec best; best.t = FLT_MAX; best.u = best.v = 0; best.pos = kNone;
    // Work distribution. Items are handed out in chunks of 64. Chunk g belongs to counter
    // g % kWorkSegments, and counter s hands out its chunks in ascending order, so that the
    // whole grid sweeps through the items on one front (neighbouring rays are in flight
    // together and find each other's nodes in the caches) while the fetch atomics are spread
    // over kWorkSegments addresses instead of queueing on one. A wave draws from the counter
    // of its workgroup; when that one has run dry it moves to one that nobody has reported
    // dry yet. Wave-uniform state.
    //
    // Small launches (fewer than half a wave of items per wave of the grid) are spread instead: chunks of 1 .. 32 items,
    // so that every wave of the grid gets some, a wave fetches again only when all its lanes are free, and the
    // lanes it does not need for rays of their own help from the first step on (see TailShare). Such a launch is
    // not limited by throughput but by the dependent node fetches of its longest rays, and the helpers walk
    // the pending subtrees of a ray side by side.
    uint32_t chunk = 64u, refill = (uint32_t)REFILL;
    constexpr uint32_t GL = 4u, GM = (1u << GL) - 1u;      // the item counters hand out groups of 16 neighbouring items
    bool spread = false, eager = false;   // eager: most lanes are free, so pending children are handed to them after every level
    if constexpr (SHARE != 0 && !LOCAL) {
        const uint32_t waves = gridDim.x * (uint32_t)(BLOCK / 64);
        if ((unsigned long long)n < (unsigned long long)waves * sc.spread_below) {
            const uint32_t per = (n + waves - 1u) / waves;
            chunk = 1u; while (chunk < per) chunk <<= 1;
            refill = 64u; spread = true; eager = chunk <= 8u;
            ts->key[threadIdx.x] = pack_hit(best); ts->pending[threadIdx.x] = 0u; tail = true;
        }
    }
    // The end of the queue is handed out item by item. With chunks, a wave that finds the counters dry still has
    // up to a chunk of unstarted items parked, another one none, and the launch ends with the unluckiest wave. So the
    // last fine_per_wave items per wave of the grid (whole chunks before them) are drawn from a second set of
    // counters (word 1 of every segment; groups of 16 items, interleaved over the counters), exactly as many as the wave has free
    // lanes for: one atomic per refill instead of one per chunk, but only for that last part, and nothing is parked
    // when the work runs out.
    uint32_t n_coarse = n, fine_range = 0;             // items [0, n_coarse) go out in chunks; every counter then has fine_range positions of the rest
    if (!LOCAL && !spread && sc.fine_per_wave && n <= 0xFFFF0000u) {      // (beyond that the 32-bit item arithmetic below could wrap)
        const unsigned long long want = (unsigned long long)gridDim.x * (uint32_t)(BLOCK / 64) * sc.fine_per_wave;
        n_coarse = want >= n ? 0u : (uint32_t)((n - want) / 64u) * 64u;
        fine_range = ((n - n_coarse + ((uint32_t)kWorkSegments << GL) - 1u) / ((uint32_t)kWorkSegments << GL)) << GL;
    }
    const uint32_t n_chunks = (n_coarse + chunk - 1u) / chunk;
    uint32_t seg = blockIdx.x % (uint32_t)kWorkSegments, loc_next = 0, loc_end = 0;
    uint32_t hop = (blockIdx.x * (uint32_t)(BLOCK / 64) + (threadIdx.x >> 6)) * 2654435761u;   // wave-uniform
    bool exhausted = false, fine = n_coarse == 0u && fine_range != 0u;
    // counters of the current part that are known to be dry: those that never had a chunk, then what this wave finds out
    unsigned long long dry_seen = fine ? 0ull : ~__ballot(work_chunk(0u, lane) < n_chunks);
    if constexpr (LOCAL) { loc_end = n; exhausted = n == 0u; }
    else {
        if (!fine && dry_seen == ~0ull) { if (fine_range != 0u) { fine = true; dry_seen = 0ull; } else exhausted = true; }
        if (!fine && !exhausted && ((dry_seen >> seg) & 1ull)) seg = nth_set_bit(~dry_seen, (hop >> 16) % (uint32_t)__popcll(~dry_seen));   // (a launch of a few chunks)
    }
    for (;;) {
        if constexpr (SHARE) {
            if (tail && item != kNone) {
                const uint32_t own = wave0 + (item >= kHelper ? (item & 63u) : lane);
                if (item >= kHelper && cur == kDone) {
                    // a helper hands in what it found
                    if (best.pos != kNone) atomicMin(&ts->key[own], pack_hit(best));
                    atomicSub(&ts->pending[own], 1u);
                    item = kNone;
                } else if (RTX_IS_ANY && cur != kDone && (uint32_t)lds_read(&ts->key[own]) != kNone) {
                    cur = kDone; stk.sp = 0;     // somebody found an occluder for this ray
                }
            }
        }
        // retire finished rays
        if (cur == kDone && item != kNone && (!SHARE || item < kHelper)) {
            bool ready = true;
            if constexpr (SHARE) {
                if (tail) {
                    ready = lds_read(&ts->pending[threadIdx.x]) == 0u;
                    const unsigned long long kk = lds_read(&ts->key[threadIdx.x]);
                    if (ready && kk < pack_hit(best)) {
                        best.pos = (uint32_t)kk; best.t = __uint_as_float((uint32_t)(kk >> 32));
                        if (!RTX_IS_ANY) { float t; tri_test(sc.tris[best.pos], ox, oy, oz, dx, dy, dz, t, best.u, best.v); }
                    }
                }
            }
            if (ready) {
                pol.retire(item, best, true);
                if constexpr (COUNT) pol.note(item, cnt.nodes - mark.nodes, cnt.tris - mark.tris);
                item = kNone;
            }
        }
        unsigned long long m = __ballot(cur == kDone && item == kNone);
        uint32_t n_idle = (uint32_t)__popcll(m);
        while (!exhausted && n_idle >= refill) {
            if (LOCAL && loc_next == loc_end) { exhausted = true; break; }
            if (loc_next == loc_end) {
                const uint32_t ask = fine ? n_idle : 1u;
                uint32_t j = 0;
                if (lane == 0) j = atomicAdd(work + seg * (uint32_t)kWorkStride + (fine ? 1u : 0u), ask);
                j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
                // chunk index j * kWorkSegments + seg; j is far below 2^26 because n < 2^32
                const uint32_t g = work_chunk(j, seg);
                const bool dry = fine ? j >= fine_range : (j >= (1u << 26) || g >= n_chunks);
                // The wave whose request crosses the end of a counter tells everybody (a bit per counter, one word per part,
                // a copy per XCD): a wave that finds its counter dry then knows where not to ask. The counters themselves are
                // never read: a load of a line that atomics are working on takes tens of microseconds when thousands of waves do it.
                const bool last = fine ? (j < fine_range && j + ask >= fine_range) : (!dry && work_chunk(j + 1u, seg) >= n_chunks);
                unsigned long long* told = (unsigned long long*)(work + (uint32_t)kWorkSegments * (uint32_t)kWorkStride) + (fine ? 1 : 0);
                if (last && lane < (uint32_t)kWorkMaskCopies) atomicOr(told + lane * (uint32_t)(kWorkStride / 2), 1ull << seg);
                if (last) dry_seen |= 1ull << seg;
                if (dry) {
                    dry_seen |= 1ull << seg;
                    // (lane 0's view for everybody: bits arrive while the lanes of one load are served, and the wave must agree on what it saw)
                    unsigned long long mk = __hip_atomic_load(told + (blockIdx.x % (uint32_t)kWorkMaskCopies) * (uint32_t)(kWorkStride / 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mk = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mk >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mk);
                    const unsigned long long live = ~(dry_seen | mk);
                    if (live == 0ull) {
                        // this part of the queue is handed out
                        if (!fine && fine_range != 0u) { fine = true; dry_seen = 0ull; seg = blockIdx.x % (uint32_t)kWorkSegments; } else exhausted = true;
                        continue;
                    }
                    // a random live one: the waves of a dry counter must not all land on the same next one
                    hop = hop * 1664525u + 1013904223u;
                    seg = nth_set_bit(live, (hop >> 16) % (uint32_t)__popcll(live));
                    continue;
                }
                if (fine) { loc_next = j; loc_end = min(j + ask, fine_range); }
                else { loc_next = g * chunk; loc_end = min(loc_next + chunk, n_coarse); }
            }
            const uint32_t take = min(n_idle, loc_end - loc_next);
            const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (cur == kDone && item == kNone && rank < take && (!fine || ((((loc_next + rank) >> GL) * (uint32_t)kWorkSegments + seg) << GL | ((loc_next + rank) & GM)) < n - n_coarse)) {
                // fine part: position x of counter seg is item x % 16 of that counter's (x / 16)-th group of 16, and the groups of the counters interleave
                const uint32_t x = loc_next + rank;
                const uint32_t k = fine ? n_coarse + ((((x >> GL) * (uint32_t)kWorkSegments + seg) << GL) | (x & GM)) : x;
                float o[3], d[3];
                bool walk;
                if constexpr (MODE == kModeMixed) walk = pol.fetch(k, o, d, maxT, any_lane); else walk = pol.fetch(k, o, d, maxT);
                ox = o[0]; oy = o[1]; oz = o[2]; dx = d[0]; dy = d[1]; dz = d[2];
                ix = 1.0f / dx; iy = 1.0f / dy; iz = 1.0f / dz;                                    // Ray::init, Geometry.h:25
                best.t = FLT_MAX; best.u = best.v = 0; best.pos = kNone; stk.sp = 0;
                if constexpr (SHARE != 0) { if (tail) { ts->key[threadIdx.x] = pack_hit(best); ts->pending[threadIdx.x] = 0u; } }   // a fresh mailbox for the new ray
                bound = RTX_IS_ANY ? (maxT + (fabsf(maxT) * kCullRel + kCullAbs)) : FLT_MAX;
                // root box first (Geometry.h:509 / :571)
                if (!walk) {}
                else if (slab_reference(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], ox, oy, oz, ix, iy, iz)) {
                    const bool finite = (fabsf(ix) <= FLT_MAX) && (fabsf(iy) <= FLT_MAX) && (fabsf(iz) <= FLT_MAX);
                    if (finite) {
                        cur = 0; item = k; mark = cnt;
                    }
                    else {
                        if (RTX_IS_ANY) traverse_exact<true, BLOCK, LDS_N>(sc, stk, ox, oy, oz, dx, dy, dz, ix, iy, iz, maxT, best);
                        else traverse_exact<false, BLOCK, LDS_N>(sc, stk, ox, oy, oz, dx, dy, dz, ix, iy, iz, maxT, best);
                        pol.retire(k, best, true);
                    }
                } else pol.retire(k, best, false);
            }
            loc_next += take;
            m = __ballot(cur == kDone && item == kNone);
            n_idle = (uint32_t)__popcll(m);
        }
        if constexpr (SHARE) {
            if (exhausted || spread) {
                if (!tail) { ts->key[threadIdx.x] = pack_hit(best); ts->pending[threadIdx.x] = 0u; tail = true; }
                const unsigned long long freeM = __ballot(cur == kDone && item == kNone);
                const unsigned long long donM = __ballot(cur != kDone && stk.sp > 0);
                if (freeM != 0ull && donM != 0ull) {
                    const uint32_t nF = (uint32_t)__popcll(freeM), nD = (uint32_t)__popcll(donM);
                    const unsigned long long below = (1ull << lane) - 1ull;
                    // free lane r helps donor r % nD and takes that donor's (r / nD)-th newest entry
                    const uint32_t r = (uint32_t)__popcll(freeM & below);
                    const uint32_t dl = nth_set_bit(donM, r % nD), q = r / nD;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the donors' stack stores have landed
                    const int d_sp = __shfl(stk.sp, (int)dl, 64);
                    const uint32_t d_item = (uint32_t)__shfl((int)item, (int)dl, 64);
                    const float sox = __shfl(ox, (int)dl, 64), soy = __shfl(oy, (int)dl, 64), soz = __shfl(oz, (int)dl, 64);
                    const float sdx = __shfl(dx, (int)dl, 64), sdy = __shfl(dy, (int)dl, 64), sdz = __shfl(dz, (int)dl, 64);
                    const float six = __shfl(ix, (int)dl, 64), siy = __shfl(iy, (int)dl, 64), siz = __shfl(iz, (int)dl, 64);
                    const float smax = __shfl(maxT, (int)dl, 64), sbound = __shfl(bound, (int)dl, 64);
                    const bool sany = MODE == kModeMixed ? (__shfl((int)any_lane, (int)dl, 64) != 0) : (MODE == kModeAny);
                    const bool is_free = (cur == kDone && item == kNone);
                    if (cur != kDone && stk.sp > 0) {
                        const uint32_t rd = (uint32_t)__popcll(donM & below);
                        const uint32_t offered = rd < nF ? (nF - rd + nD - 1u) / nD : 0u;
                        stk.sp -= (int)min(offered, (uint32_t)stk.sp);
                    } else if (is_free && (int)q < d_sp) {
                        const uint2 e = stk.peek_other(d_sp - 1 - (int)q, (int)dl - (int)lane);
                        ox = sox; oy = soy; oz = soz; dx = sdx; dy = sdy; dz = sdz; ix = six; iy = siy; iz = siz; maxT = smax; bound = sbound; any_lane = sany;
                        best.t = FLT_MAX; best.u = best.v = 0; best.pos = kNone; stk.sp = 0;
                        const uint32_t own = d_item >= kHelper ? (d_item & 63u) : dl;
                        item = kHelper | own;
                        atomicAdd(&ts->pending[wave0 + own], 1u);
                        cur = (__uint_as_float(e.y) > bound) ? kDone : e.x;
                        if constexpr (COUNT) mark = cnt;
                    }
                }
            }
        }
        if (exhausted && __all(cur == kDone && item == kNone)) break;
        if (COUNT) wave_iters[2]++;
        // every lane descends until it holds a leaf
        if (SCHED == 0) {
            while (!(cur & kLeafBit) && cur != kDone) { if (COUNT) wave_iters[0]++; cur = step_inner<COUNT, BLOCK, LDS_N, MODE == kModeAny ? 1 : 0>(sc, stk, cur, ox, oy, oz, ix, iy, iz, bound, &cnt); }
        } else if (SCHED == 1) {
            if (!(cur & kLeafBit) && cur != kDone) { if (COUNT) wave_iters[0]++; cur = step_inner<COUNT, BLOCK, LDS_N, MODE == kModeAny ? 1 : 0>(sc, stk, cur, ox, oy, oz, ix, iy, iz, bound, &cnt); }
        } else {
            // descend until SCHED lanes hold a leaf (or nobody has an inner node left)
            for (;;) {
                const bool inner = !(cur & kLeafBit) && cur != kDone;
                if (!__any(inner)) break;
                if (inner) { if (COUNT) wave_iters[0]++; cur = step_inner<COUNT, BLOCK, LDS_N, MODE == kModeAny ? 1 : 0>(sc, stk, cur, ox, oy, oz, ix, iy, iz, bound, &cnt); }
                if (__popcll(__ballot((cur & kLeafBit) != 0)) >= SCHED) break;
                if (eager) break;      // one level at a time: the children just pushed go to free lanes at once
            }
        }
        if (cur & kLeafBit) {
            const uint32_t first = cur & 0x07FFFFFFu, nt = ((cur >> 27) & 15u) + 1u;
            bool occluded = false;
            for (uint32_t k = 0; k < nt; k += TRI_UNROLL) {
                TriIsect T[TRI_UNROLL];
                if constexpr (TRI_UNROLL == 4) {
                    // four consecutive 36-byte records are 144 bytes = nine 16-byte loads instead of twelve
                    // (4-byte aligned; records past the leaf's end are loaded but not tested, the array is padded)
                    const TriQuad Q = *(const TriQuad*)(sc.tris + first + k);
                    // the first triangle's third vertex is only read once its determinant has passed, and the compiler would
                    // move its load there: a second memory round trip in every triangle phase. Ask for it here, with the others.
                    asm volatile("" :: "v"(Q.f[6]), "v"(Q.f[7]), "v"(Q.f[8]));
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int q = 0; q < 3; q++) { T[j].e1[q] = Q.f[j * 9 + q]; T[j].e2[q] = Q.f[j * 9 + 3 + q]; T[j].v2[q] = Q.f[j * 9 + 6 + q]; }
                } else {
#pragma unroll
                    for (int j = 0; j < TRI_UNROLL; j++) T[j] = sc.tris[first + min(k + (uint32_t)j, nt - 1u)];
                }
                if (COUNT) { wave_iters[1]++; cnt.groups++; }
#pragma unroll
                for (int j = 0; j < TRI_UNROLL; j++) {
                    if (k + (uint32_t)j < nt) {
                        if (COUNT) cnt.tris++;
                        float t, u, v;
                        if (tri_test(T[j], ox, oy, oz, dx, dy, dz, t, u, v)) {
                            if (RTX_IS_ANY) { if (t <= maxT && !occluded) { occluded = true; best.pos = first + k + (uint32_t)j; best.t = t; } }
                            else consider(best, t, u, v, first + k + (uint32_t)j);
                        }
                    }
                }
                if (RTX_IS_ANY && occluded) break;
            }
            if (RTX_IS_ANY && occluded) {
                cur = kDone;
                if constexpr (SHARE) { if (tail) { atomicMin(&ts->key[wave0 + (item >= kHelper ? (item & 63u) : lane)], pack_hit(best)); stk.sp = 0; } }
            } else {
                if (!RTX_IS_ANY) {
                    float bt = best.t;
                    if constexpr (SHARE) {
                        if (tail) {
                            // publish this lane's best and prune against the best anyone has found for the ray
                            const uint32_t own = wave0 + (item >= kHelper ? (item & 63u) : lane);
                            const unsigned long long mine = pack_hit(best);
                            const unsigned long long old = best.pos != kNone ? atomicMin(&ts->key[own], mine) : lds_read(&ts->key[own]);
                            bt = __uint_as_float((uint32_t)((old < mine ? old : mine) >> 32));
                        }
                    }
                    bound = bt + (fabsf(bt) * kCullRel + kCullAbs);
                }
                cur = pop_next(stk, bound);
            }
        }
    }
#undef RTX_IS_ANY
    if (COUNT) {
        atomicAdd(&counters[0], (unsigned long long)cnt.nodes); atomicAdd(&counters[1], (unsigned long long)cnt.tris); atomicAdd(&counters[5], (unsigned long long)cnt.groups);
        // wave-level trip counts: the maximum over the lanes of a wave is what the wave executed
        for (int q = 0; q < 3; q++) {
            uint32_t v = wave_iters[q];
            for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
            if (lane == 0) atomicAdd(&counters[2 + q], (unsigned long long)v);
        }
    }
}

// Scene::visible's segment -> ray set-up, RTBase/Scene.h:170-178
__device__ __forceinline__ void segment_to_ray(const float* s, float o[3], float d[3], float& maxT)
{
    float dx = s[3] - s[0], dy = s[4] - s[1], dz = s[5] - s[2];
    float len2 = (dx * dx) + (dy * dy) + (dz * dz);
    maxT = sqrtf(len2) - (2.0f * 0.001f);
    float l = 1.0f / sqrtf(len2);
    dx = dx * l; dy = dy * l; dz = dz * l;
    d[0] = dx; d[1] = dy; d[2] = dz;
    o[0] = s[0] + (dx * 0.001f); o[1] = s[1] + (dy * 0.001f); o[2] = s[2] + (dz * 0.001f);
}

} // namespace rtx
