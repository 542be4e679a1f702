// traverse_dyn2.h — the dynamic-fetch traversal of traverse_dyn.h with TWO rays per lane.
//
// One ray per lane leaves lanes idle inside a wave for a structural reason: the wave alternates between a phase in
// which lanes step through inner nodes and a phase in which lanes test leaf triangles, and a lane whose ray is in the
// other state waits (on the headline workload about two thirds of the lanes take part in a node step and half of them
// in a triangle step, while the vector ALU is the unit the kernel keeps busiest). Here every lane holds a second ray
// "parked" in registers beside the active one, with a traversal stack of its own. Before each step of the node phase a lane
// whose active ray has no inner node to visit, but whose parked ray has, swaps the two; before the triangle phase
// likewise for leaves. What is traversed per ray, the arithmetic and the tie-break are those of traverse_dyn: only
// which ray a lane works on at a given moment changes, so the results are identical.
//
// Launches that are spread thin (traverse_dyn.h) have lanes to spare and gain nothing from a second ray: they run on
// traverse_dyn. This variant has no instrumentation.
#pragma once

#include "traverse_dyn.h"

namespace rtx {

// mailboxes of the tail sharing, one set per ray slot of a lane
template <int BLOCK>
struct TailShare2 {
    unsigned long long key[2][BLOCK];
    uint32_t pending[2][BLOCK];
};

template <class Policy, int MODE, int REFILL, int BLOCK, int LDS_N, int SCHED, int SHARE>
__device__ __forceinline__ void traverse_dyn2(const DeviceScene& sc, unsigned long long* lds_array, uint2* spill_slab, uint32_t spill_stride, uint32_t spill_slab_entries,
                                              uint32_t n, uint32_t* work, Policy& pol, TailShare2<BLOCK>* ts)
{
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    // item values from kHelper up (except kNone): the lane helps the ray in slot ((item >> 6) & 1) of lane (item & 63) of its wave
    constexpr uint32_t kHelper = 0xFFFFFF00u;
    const uint32_t wave0 = threadIdx.x & ~63u;
    bool tail = false;

    // ---- the two ray contexts of the lane: A is the one the code below works on, B is parked -------------------
    struct Ctx {
        float ox, oy, oz, dx, dy, dz, ix, iy, iz, bound, maxT;
        HitRec best; uint32_t cur, item, slot; bool any;
    };
    Ctx A, B;
    A.cur = B.cur = kDone; A.item = B.item = kNone; A.slot = 0u; B.slot = 1u;
    A.best.t = B.best.t = FLT_MAX; A.best.u = A.best.v = B.best.u = B.best.v = 0; A.best.pos = B.best.pos = kNone;
    A.ox = A.oy = A.oz = A.dx = A.dy = A.dz = A.ix = A.iy = A.iz = A.maxT = 0; A.bound = FLT_MAX; A.any = MODE == kModeAny;
    B = A; B.slot = 1u;
    LaneStack<BLOCK, LDS_N> stk, stkB;            // stk belongs to A, stkB to B (swapped with them)
    stk.init(lds_array, spill_slab, spill_stride);
    stkB.init(lds_array + (size_t)BLOCK * LDS_N, spill_slab + (size_t)spill_slab_entries * spill_stride, spill_stride);
#define RTX_ANY_OF(C) (MODE == kModeMixed ? (C).any : MODE == kModeAny)
    auto swap_ctx = [&]() {
        Ctx t = A; A = B; B = t;
        LaneStack<BLOCK, LDS_N> s = stk; stk = stkB; stkB = s;
    };
    auto idle = [&](const Ctx& c) { return c.cur == kDone && c.item == kNone; };
    auto inner = [&](const Ctx& c) { return !(c.cur & kLeafBit) && c.cur != kDone; };
    auto leaf = [&](const Ctx& c) { return (c.cur & kLeafBit) != 0u; };   // kDone carries no leaf bit

    const uint32_t n_chunks = (n + 63u) / 64u;
    uint32_t seg = blockIdx.x % (uint32_t)kWorkSegments, loc_next = 0, loc_end = 0;
    bool exhausted = false;

    // retire the active ray if it is done (and, in the tail, all its helpers have handed in)
    auto settle_active = [&]() {
        if constexpr (SHARE != 0) {
            if (tail && A.item != kNone) {
                if (A.item >= kHelper && A.cur == kDone) {
                    const uint32_t hs = (A.item >> 6) & 1u, hl = wave0 + (A.item & 63u);
                    if (A.best.pos != kNone) atomicMin(&ts->key[hs][hl], pack_hit(A.best));
                    atomicSub(&ts->pending[hs][hl], 1u);
                    A.item = kNone;
                } else if (RTX_ANY_OF(A) && A.cur != kDone) {
                    const uint32_t hs = A.item >= kHelper ? ((A.item >> 6) & 1u) : A.slot, hl = wave0 + (A.item >= kHelper ? (A.item & 63u) : lane);
                    if ((uint32_t)lds_read(&ts->key[hs][hl]) != kNone) { A.cur = kDone; stk.sp = 0; }
                }
            }
        }
        if (A.cur == kDone && A.item != kNone && (SHARE == 0 || A.item < kHelper)) {
            bool ready = true;
            if constexpr (SHARE != 0) {
                if (tail) {
                    ready = lds_read(&ts->pending[A.slot][threadIdx.x]) == 0u;
                    const unsigned long long kk = lds_read(&ts->key[A.slot][threadIdx.x]);
                    if (ready && kk < pack_hit(A.best)) {
                        A.best.pos = (uint32_t)kk; A.best.t = __uint_as_float((uint32_t)(kk >> 32));
                        if (!RTX_ANY_OF(A)) { float t; tri_test(sc.tris[A.best.pos], A.ox, A.oy, A.oz, A.dx, A.dy, A.dz, t, A.best.u, A.best.v); }
                    }
                }
            }
            if (ready) { pol.retire(A.item, A.best, true); A.item = kNone; }
        }
    };

    for (;;) {
        settle_active();
        // the parked ray cannot finish while parked, but a helper's donor may have been waiting on it: give it its turn
        if (B.cur == kDone && B.item != kNone) { swap_ctx(); settle_active(); }
        // refill: a lane takes a new ray whenever one of its two slots is empty (the filled slot becomes the active one)
        unsigned long long m = __ballot(idle(A) || idle(B));
        uint32_t n_free = (uint32_t)__popcll(m);
        while (!exhausted && n_free >= (uint32_t)REFILL) {
            if (loc_next == loc_end) {
                uint32_t j = 0;
                if (lane == 0) j = atomicAdd(work + seg * (uint32_t)kWorkStride, 1u);
                j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
                const uint32_t g = work_chunk(j, seg);
                if (j >= (1u << 26) || g >= n_chunks) {
                    const uint32_t cj = __hip_atomic_load(work + lane * (uint32_t)kWorkStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long live = __ballot(cj < (1u << 26) && work_chunk(cj, lane) < n_chunks);
                    if (live == 0ull) { exhausted = true; continue; }
                    const unsigned long long rot = seg ? ((live >> seg) | (live << (64u - seg))) : live;
                    seg = (seg + (uint32_t)__ffsll((long long)rot) - 1u) & 63u;
                    continue;
                }
                loc_next = g * 64u; loc_end = min(loc_next + 64u, n);
            }
            const uint32_t take = min(n_free, loc_end - loc_next);
            const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if ((idle(A) || idle(B)) && rank < take) {
                if (!idle(A)) swap_ctx();
                const uint32_t k = loc_next + rank;
                float o[3], d[3];
                if constexpr (MODE == kModeMixed) pol.fetch(k, o, d, A.maxT, A.any); else pol.fetch(k, o, d, A.maxT);
                A.ox = o[0]; A.oy = o[1]; A.oz = o[2]; A.dx = d[0]; A.dy = d[1]; A.dz = d[2];
                A.ix = 1.0f / A.dx; A.iy = 1.0f / A.dy; A.iz = 1.0f / A.dz;                            // Ray::init, Geometry.h:25
                A.best.t = FLT_MAX; A.best.u = A.best.v = 0; A.best.pos = kNone; stk.sp = 0;
                A.bound = RTX_ANY_OF(A) ? (A.maxT + (fabsf(A.maxT) * kCullRel + kCullAbs)) : FLT_MAX;
                if (slab_reference(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], A.ox, A.oy, A.oz, A.ix, A.iy, A.iz)) {
                    const bool finite = (fabsf(A.ix) <= FLT_MAX) && (fabsf(A.iy) <= FLT_MAX) && (fabsf(A.iz) <= FLT_MAX);
                    if (finite) { A.cur = 0; A.item = k; }
                    else {
                        if (RTX_ANY_OF(A)) traverse_exact<true, BLOCK, LDS_N>(sc, stk, A.ox, A.oy, A.oz, A.dx, A.dy, A.dz, A.ix, A.iy, A.iz, A.maxT, A.best);
                        else traverse_exact<false, BLOCK, LDS_N>(sc, stk, A.ox, A.oy, A.oz, A.dx, A.dy, A.dz, A.ix, A.iy, A.iz, A.maxT, A.best);
                        pol.retire(k, A.best, true);
                    }
                } else pol.retire(k, A.best, false);
            }
            loc_next += take;
            m = __ballot(idle(A) || idle(B));
            n_free = (uint32_t)__popcll(m);
        }
        // a lane whose active slot is empty works on its parked ray
        if (idle(A) && !idle(B)) swap_ctx();
        if constexpr (SHARE != 0) {
            if (exhausted) {
                if (!tail) {
                    ts->key[A.slot][threadIdx.x] = pack_hit(A.best); ts->pending[A.slot][threadIdx.x] = 0u;
                    ts->key[B.slot][threadIdx.x] = pack_hit(B.best); ts->pending[B.slot][threadIdx.x] = 0u;
                    tail = true;
                }
                // free lanes (both slots empty) take pending subtrees off the ACTIVE stacks of busy lanes, newest first
                const unsigned long long freeM = __ballot(idle(A) && idle(B));
                const unsigned long long donM = __ballot(A.cur != kDone && stk.sp > 0);
                if (freeM != 0ull && donM != 0ull) {
                    const uint32_t nF = (uint32_t)__popcll(freeM), nD = (uint32_t)__popcll(donM);
                    const unsigned long long below = (1ull << lane) - 1ull;
                    const uint32_t r = (uint32_t)__popcll(freeM & below);
                    const uint32_t dl = nth_set_bit(donM, r % nD), q = r / nD;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    const int d_sp = __shfl(stk.sp, (int)dl, 64);
                    const uint32_t d_item = (uint32_t)__shfl((int)A.item, (int)dl, 64);
                    const uint32_t d_slot = (uint32_t)__shfl((int)A.slot, (int)dl, 64);
                    const float sox = __shfl(A.ox, (int)dl, 64), soy = __shfl(A.oy, (int)dl, 64), soz = __shfl(A.oz, (int)dl, 64);
                    const float sdx = __shfl(A.dx, (int)dl, 64), sdy = __shfl(A.dy, (int)dl, 64), sdz = __shfl(A.dz, (int)dl, 64);
                    const float six = __shfl(A.ix, (int)dl, 64), siy = __shfl(A.iy, (int)dl, 64), siz = __shfl(A.iz, (int)dl, 64);
                    const float smax = __shfl(A.maxT, (int)dl, 64), sbound = __shfl(A.bound, (int)dl, 64);
                    const bool sany = MODE == kModeMixed ? (__shfl((int)A.any, (int)dl, 64) != 0) : (MODE == kModeAny);
                    const bool is_free = idle(A) && idle(B);
                    if (A.cur != kDone && stk.sp > 0) {
                        const uint32_t rd = (uint32_t)__popcll(donM & below);
                        const uint32_t offered = rd < nF ? (nF - rd + nD - 1u) / nD : 0u;
                        stk.sp -= (int)min(offered, (uint32_t)stk.sp);
                    } else if (is_free && (int)q < d_sp) {
                        // the donor's active stack lives in the half of the LDS array (and the slab) of its slot
                        const LaneStack<BLOCK, LDS_N>& mine = (A.slot == d_slot) ? stk : stkB;
                        const uint2 e = mine.peek_other(d_sp - 1 - (int)q, (int)dl - (int)lane);
                        A.ox = sox; A.oy = soy; A.oz = soz; A.dx = sdx; A.dy = sdy; A.dz = sdz; A.ix = six; A.iy = siy; A.iz = siz; A.maxT = smax; A.bound = sbound; A.any = sany;
                        A.best.t = FLT_MAX; A.best.u = A.best.v = 0; A.best.pos = kNone; stk.sp = 0;
                        const uint32_t own = d_item >= kHelper ? (d_item & 127u) : ((d_slot << 6) | dl);   // slot bit + lane of the ray's owner
                        A.item = kHelper | own;
                        atomicAdd(&ts->pending[(own >> 6) & 1u][wave0 + (own & 63u)], 1u);
                        A.cur = (__uint_as_float(e.y) > A.bound) ? kDone : e.x;
                    }
                }
            }
        }
        if (exhausted && __all(idle(A) && idle(B))) break;
        // node phase: every lane steps whichever of its rays has an inner node to visit, until SCHED lanes hold a leaf
        for (;;) {
            if (!inner(A) && inner(B)) swap_ctx();
            const bool go = inner(A);
            if (!__any(go)) break;
            if (go) A.cur = step_inner<false, BLOCK, LDS_N, MODE == kModeAny ? 1 : 0>(sc, stk, A.cur, A.ox, A.oy, A.oz, A.ix, A.iy, A.iz, A.bound, nullptr);
            if (__popcll(__ballot(leaf(A) || leaf(B))) >= SCHED) break;
        }
        // triangle phase on whichever ray holds a leaf
        if (!leaf(A) && leaf(B)) swap_ctx();
        if (leaf(A)) {
            const uint32_t first = A.cur & 0x07FFFFFFu, nt = ((A.cur >> 27) & 15u) + 1u;
            bool occluded = false;
            for (uint32_t k = 0; k < nt; k += 4) {
                TriIsect T[4];
                const TriQuad Q = *(const TriQuad*)(sc.tris + first + k);
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int q = 0; q < 3; q++) { T[j].e1[q] = Q.f[j * 9 + q]; T[j].e2[q] = Q.f[j * 9 + 3 + q]; T[j].v2[q] = Q.f[j * 9 + 6 + q]; }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (k + (uint32_t)j < nt) {
                        float t, u, v;
                        if (tri_test(T[j], A.ox, A.oy, A.oz, A.dx, A.dy, A.dz, t, u, v)) {
                            if (RTX_ANY_OF(A)) { if (t <= A.maxT && !occluded) { occluded = true; A.best.pos = first + k + (uint32_t)j; A.best.t = t; } }
                            else consider(A.best, t, u, v, first + k + (uint32_t)j);
                        }
                    }
                }
                if (RTX_ANY_OF(A) && occluded) break;
            }
            const uint32_t hs = A.item >= kHelper ? ((A.item >> 6) & 1u) : A.slot, hl = wave0 + (A.item >= kHelper ? (A.item & 63u) : lane);
            if (RTX_ANY_OF(A) && occluded) {
                A.cur = kDone;
                if constexpr (SHARE != 0) { if (tail) { atomicMin(&ts->key[hs][hl], pack_hit(A.best)); stk.sp = 0; } }
            } else {
                if (!RTX_ANY_OF(A)) {
                    float bt = A.best.t;
                    if constexpr (SHARE != 0) {
                        if (tail) {
                            const unsigned long long mine = pack_hit(A.best);
                            const unsigned long long old = A.best.pos != kNone ? atomicMin(&ts->key[hs][hl], mine) : lds_read(&ts->key[hs][hl]);
                            bt = __uint_as_float((uint32_t)((old < mine ? old : mine) >> 32));
                        }
                    }
                    A.bound = bt + (fabsf(bt) * kCullRel + kCullAbs);
                }
                A.cur = pop_next(stk, A.bound);
            }
        }
    }
#undef RTX_ANY_OF
}

} // namespace rtx
