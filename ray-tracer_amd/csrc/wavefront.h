// wavefront.h — progressive rendering as a wavefront of kernels.
//
// One call of wavefront_render_pass() is one RayTracer::renderMT() (RTBase/Renderer.h:753-782):
// one sample per active pixel for path tracing and instant radiosity, one light path per
// pixel for light tracing. Instead of recursing per pixel, every stage is a kernel over a
// queue of live paths:
//
//   ray generation -> [ closest hit -> shade (NEE + BSDF sample) -> shadow rays ]* -> film
//
// Queues are compacted with a wave64 ballot + one atomic per wave; queue lengths stay on
// the device (kernels read them), so a pass needs no host synchronisation.
// Path state is structure-of-arrays, indexed by path slot; slot i belongs to the 16x16
// reference tile i/256 (RTBase/Renderer.h:38), wave i/64 covers an 8x8 pixel block.
#pragma once

#include "context.h"
#include "shade.h"
#include "trace_kernels.h"
#include "traverse_packet.h"

// statistics of the tuning build (librtx_tuning.so): extra kernel arguments and the code that fills them
#ifdef RTX_TUNING
#define RTX_PACKET_STATS(...) __VA_ARGS__
#else
#define RTX_PACKET_STATS(...)
#endif

namespace rtx {

constexpr uint32_t kFlagCanHitLight = 1u;   // emission counts at the next hit (Renderer.h:487,516)
constexpr uint32_t kFlagStarted = 2u;       // slot carries a sample this pass
// counter words, zeroed once per pass: queue lengths [0,96) live paths per depth, [96,192) shadow /
// connection segments per depth, every one on a line of its own (a kernel reads the length of its input queue
// while the workgroups that started earlier push onto its output queue with atomics, and a load of a line that
// atomics are working on waits behind them); then the work-counter slots of the closest-hit and shadow-ray
// launches of each depth (dynamic ray fetch)
constexpr int kShadowCounterBase = 96;
constexpr int kQueueStride = 32;                                            // queue length q lives in word q * kQueueStride: a 128-byte line of its own
constexpr int kClosestWorkBase = 192 * kQueueStride;                        // 96 work-counter slots of the closest-hit launches
constexpr int kShadowWorkBase = kClosestWorkBase + 96 * kWorkSlotWords;     // 96 slots of the shadow / splat launches
// Path tracing keeps every queue behind depth 0 as kQueueStreams streams with a length of their own (see stream_push):
// queue q, stream c at word stream_word(q, c), a line each like the dense queue lengths
constexpr int kQueueStreams = 16;
constexpr int kStreamCounterBase = kShadowWorkBase + 96 * kWorkSlotWords;
constexpr int kCounterWords = kStreamCounterBase + 192 * kQueueStreams * kQueueStride;
constexpr int kQueueSlack = kQueueStreams * 64;                             // queue arrays hold P + kQueueSlack entries: streamed queues round up to whole blocks of 64 per stream
__host__ __device__ constexpr uint32_t stream_word(uint32_t q, uint32_t c) { return (uint32_t)kStreamCounterBase + (q * (uint32_t)kQueueStreams + c) * (uint32_t)kQueueStride; }

struct WfPtrs {
    rtx_ray* ray; rtx_hit* hit;
    float* T;          // [P][3] path throughput
    float* L;          // [P][3] radiance of the sample (light tracing / VPL: emitted Le of the path)
    float* xy;         // [P][2] film position of the sample
    uint32_t* key; uint32_t* dim; float* mis; uint32_t* flags;
    uint32_t* qa; uint32_t* qb;       // live-path queues (ping-pong)
    uint32_t* cnt;                    // [kCounterWords] queue lengths, zeroed per pass
    float* sseg;       // [P][6] shadow segment p1 -> p2
    float* sval;       // [P][3] contribution carried by the shadow ray
    float* sxy;        // [P][2] splat position (light tracing)
    uint32_t* spath;   // [P]    path slot the contribution belongs to
    float* vpl;        // [V][10] x(3) n(3) Le(3) valid, slot j * (maxBounces + 1) + d of light path j
    float* vplc;       // [<= V][10] the valid ones, packed in list order
    uint32_t* vpl_total;
    float* ir_bounds;  // [P / 256][12] boxes of the gather points and of their shading normals, per block of 256 places of the gather queue
    float* film; uint32_t* tile_samples;
    uint32_t* tile_off;   // [n_tiles] queue offset of the tile's first camera sample this pass
    uint32_t P, W, H, tiles_x, n_tiles, vpl_slots;
    // Path slots exist for the tiles this context renders only: local tile l = slot / 256 is film tile l * tile_mul + tile_add
    // (every tile, mul 1 add 0, unless the tiles are split over GPUs: then the tiles t % world == rank, mul = world, add = rank)
    uint32_t tile_mul, tile_add, n_local_tiles;
    uint32_t* depth_stats;   // pinned host memory, [96]: queue positions per path depth of the last path-tracing pass that finished (0xFFFFFFFF: none yet)
};

struct WavefrontBuffers {
    DevBuf ray, hit, T, L, xy, key, dim, mis, flags, qa, qb, cnt, sseg, sval, sxy, spath, vpl, vpl_total, tile_off, ir_vis, ir_bounds;
    WfPtrs p{};
    uint32_t vpl_capacity = 0;
    bool cnt_clean = false;      // the last pass on these buffers zeroed the counters itself when it was done with them (path tracing, box filter)
};

struct PassParams {
    rtx_settings st; int spp; uint32_t sample_index; uint32_t depth;
};

// ---- small device helpers -------------------------------------------------------------------

// Append to a queue: one atomic per workgroup (returning atomics on one address complete at
// roughly one per 11 ns, so the number of them per launch matters); the lanes that push
// take consecutive places. Every thread of the workgroup must call this, from uniform
// control flow.
__device__ __forceinline__ uint32_t queue_slot(bool push, uint32_t* counter)
{
    __shared__ uint32_t s_count[kBlock / 64], s_base;
    const unsigned long long mask = __ballot(push);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) s_count[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int i = 0; i < kBlock / 64; i++) total += s_count[i];
        s_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t at = s_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    for (uint32_t i = 0; i < wave; i++) at += s_count[i];
    __syncthreads();                        // the next call may overwrite the counts
    return push ? at : 0xFFFFFFFFu;
}

// Streamed queues. A queue push that costs one atomic per WORKGROUP (queue_slot) makes the waves of a workgroup wait
// for each other: the wave whose paths all left the scene sits at the barrier until the wave that shades surface hits
// has walked its chain of dependent loads, and holds its registers meanwhile. One atomic per WAVE needs no barrier, but
// returning atomics on one address complete at ~90 M/s, far too few for 30 000 waves per launch. So a queue is
// kQueueStreams streams, each with its own length counter; stream c owns the blocks of 64 queue positions whose index is
// c modulo kQueueStreams, and a wave appends to one stream with one atomic. A consumer sees positions
// [0, kQueueStreams * 64 * blocks of the longest stream); position k holds an entry if its place within its stream lies
// below that stream's length (stream_valid), the rest are holes, skipped at fetch. The producers' choice of stream keeps
// the streams equally long but for chance, and bounds them: see k_pt_shade.
__device__ __forceinline__ uint32_t stream_position(uint32_t e, uint32_t c) { return (((e >> 6) * (uint32_t)kQueueStreams + c) << 6) | (e & 63u); }
struct StreamLens { uint32_t len[kQueueStreams]; };
// every thread of the workgroup calls this (there is a barrier inside): the lengths of queue q into `s`, returns the number of positions
__device__ __forceinline__ uint32_t stream_load(const uint32_t* cnt, uint32_t q, StreamLens& s)
{
    if (threadIdx.x < (uint32_t)kQueueStreams) s.len[threadIdx.x] = cnt[stream_word(q, threadIdx.x)];
    __syncthreads();
    uint32_t blocks = 0;
#pragma unroll
    for (int c = 0; c < kQueueStreams; c++) blocks = max(blocks, (s.len[c] + 63u) >> 6);
    return blocks * (uint32_t)(kQueueStreams * 64);
}
__device__ __forceinline__ bool stream_valid(const StreamLens& s, uint32_t k)
{
    const uint32_t g = k >> 6;
    return (((g / (uint32_t)kQueueStreams) << 6) | (k & 63u)) < s.len[g % (uint32_t)kQueueStreams];
}
// every lane of the wave calls this from uniform control flow: the pushing lanes take consecutive entries of the stream
// whose length is *counter; returns the queue position (0xFFFFFFFF for lanes that do not push)
__device__ __forceinline__ uint32_t stream_push(bool push, uint32_t* counter, uint32_t stream)
{
    const unsigned long long m = __ballot(push);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (m != 0ull) {
        if (lane == 0) base = atomicAdd(counter, (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    }
    return push ? stream_position(base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), stream) : 0xFFFFFFFFu;
}

__device__ __forceinline__ void slot_pixel(const WfPtrs& wf, uint32_t slot, uint32_t& tile, uint32_t& x, uint32_t& y)
{
    const uint32_t tiles_x = wf.tiles_x;
    tile = (slot >> 8) * wf.tile_mul + wf.tile_add;
    const uint32_t j = slot & 255u, w = j >> 6, l = j & 63u;
    x = (tile % tiles_x) * 16u + (((w & 1u) << 3) | (l & 7u));
    y = (tile / tiles_x) * 16u + (((w >> 1) << 3) | (l >> 3));
}

__device__ __forceinline__ bool tile_active(const WfPtrs& w, const PassParams& pp, uint32_t tile)
{
    const rtx_settings& s = pp.st;
    // processTile's skip, RTBase/Renderer.h:682-687; with the sample split across GPUs the
    // count that matters is the global sample index this pass renders
    const int cur = s.shard_mode == 2 ? (int)pp.sample_index : pp.spp;
    if (cur > (int)s.init_spp && cur > (int)w.tile_samples[tile] && s.algorithm != RTX_AL_LIGHT_TRACE) return false;
    if (s.shard_mode == 1 && s.shard_world > 1 && (int)(tile % (uint32_t)s.shard_world) != s.shard_rank) return false;
    return true;
}

// The camera sample path slot `slot` takes this pass, processTile's per-sample set-up (RTBase/Renderer.h:695-697,
// 730-731): false if the slot has none (outside the film, or its tile is skipped). The path tracer's kernels that need
// the sample (closest hit and shade kernel of depth 0, film accumulation) each derive it from the slot number: the
// generator is a function of (seed, pixel, sample index, dimension), so nothing has to be stored and read back.
__device__ __forceinline__ bool camera_sample(const WfPtrs& w, const PassParams& pp, uint32_t slot, float& px, float& py, Rng& rng)
{
    if (slot >= w.P) return false;
    uint32_t tile, x, y;
    slot_pixel(w, slot, tile, x, y);
    if (!(x < w.W && y < w.H && tile_active(w, pp, tile))) return false;
    rng.key = rtx_rng_key(pp.st.seed, RTX_STREAM_CAMERA, y * w.W + x, pp.sample_index); rng.dim = 0;
    px = x + rng.next(); py = y + rng.next();
    return true;
}

__device__ __forceinline__ void st3(float* p, C3 c) { p[0] = c.r; p[1] = c.g; p[2] = c.b; }
__device__ __forceinline__ void st3(float* p, V3 c) { p[0] = c.x; p[1] = c.y; p[2] = c.z; }

// Film::splat, RTBase/Imaging.h:318-340 (filters :146-222). Float atomics: with the box
// filter a pixel normally receives one sample per pass, but px = x + r can round up into
// the next pixel, and the wide filters and light tracing write anywhere.
__device__ __forceinline__ float filter_gauss(float d) { const float a = 2.5f; const float t2 = expf(-a * (2 * 2)); return expf(-a * (d * d)) - t2; }
__device__ __forceinline__ float filter_mitchell(float d)
{
    const float B = 1.0f / 3.0f, C = 1.0f / 3.0f;
    const float a1 = (1.0f / 6.0f) * (12 - 9 * B - 6 * C), a2 = (-18 + 12 * B + 6 * C), a3 = (6 - 2 * B);
    const float b1 = (1.0f / 6.0f) * (-B - 6 * C), b2 = (6 * B + 30 * C), b3 = (-12 * B - 48 * C), b4 = (8 * B + 24 * C);
    d = fabsf(d);
    if (d >= 2) return 0;
    float ds = d * d, dc = ds * d;
    if (d >= 0 && d < 1) return a1 * dc + a2 * ds + a3;
    return b1 * dc + b2 * ds + b3 * d + b4;
}
__device__ __forceinline__ float filter_weight(int filter, float x, float y)
{
    if (filter == RTX_FT_BOX) return (fabsf(x) <= 1.0f && fabsf(y) <= 1.0f) ? 1.0f : 0.0f;
    if (filter == RTX_FT_GAUSSIAN) return filter_gauss(x) * filter_gauss(y);
    return filter_mitchell(x) * filter_mitchell(y);
}
__device__ __forceinline__ void film_splat(float* film, int W, int H, int filter, float x, float y, C3 L)
{
    const int size = filter == RTX_FT_BOX ? 0 : 2;
    float total = 0;
    for (int i = -size; i <= size; i++) for (int j = -size; j <= size; j++) {
        int px = (int)x + j, py = (int)y + i;
        if (px >= 0 && px < W && py >= 0 && py < H) total += filter_weight(filter, px - x, py - y);
    }
    for (int i = -size; i <= size; i++) for (int j = -size; j <= size; j++) {
        int px = (int)x + j, py = (int)y + i;
        if (px >= 0 && px < W && py >= 0 && py < H) {
            float wgt = filter_weight(filter, px - x, py - y);
            C3 v = L * wgt / total;
            float* f = film + 3 * ((size_t)py * W + px);
            atomicAdd(f, v.r); atomicAdd(f + 1, v.g); atomicAdd(f + 2, v.b);
        }
    }
}

// Next-event estimation of RayTracer::computeDirect, RTBase/Renderer.h:417-474, up to the
// visibility test: returns true and fills the shadow segment + the value the path gains if
// the segment turns out visible.
__device__ __forceinline__ bool direct_request(const DeviceScene& sc, const Shading& sd, Rng& rng, bool useMis, float misPdf,
                                               V3& p2, C3& value)
{
    if (isPureSpecular(*sd.mat)) return false;
    if (sc.n_lights == 0) return false;
    float pmf; const LightDev& light = sample_light(sc, rng, pmf);
    float pdf; C3 emitted; V3 p = light_sample(sc, light, rng, emitted, pdf);
    pdf *= pmf;
    if (light.kind == 0) {
        V3 wi = p - sd.x; float lsq = lengthSq(wi); wi = normalize(wi);
        float g = (wmax(dot(wi, sd.sNormal), 0.0f) * wmax(-dot(wi, ld3(light.gnormal)), 0.0f)) / lsq;
        if (g > 0) { p2 = p; value = bsdf_evaluate(sc, sd, wi) * emitted * g / pdf; return true; }
    } else {
        V3 wi = normalize(p);
        float g = wmax(dot(wi, sd.sNormal), 0.0f);
        if (g > 0) {
            p2 = sd.x + (p * 1000.0f);
            float weight = useMis ? pdf / (pdf + misPdf) : 1.0f;
            value = bsdf_evaluate(sc, sd, wi) * emitted * g * weight / pdf;
            return true;
        }
    }
    return false;
}

// ---- path tracing ---------------------------------------------------------------------------

// Per-sample set-up of processTile, RTBase/Renderer.h:695-732 (and the pixel-centre variant
// for the draw modes, :699-703).
__global__ __launch_bounds__(kBlock) void k_raygen(WfPtrs w, CameraDev cam, PassParams pp)
{
    const uint32_t slot = global_thread();
    bool active = false; uint32_t tile = 0, x = 0, y = 0;
    if (slot < w.P) {
        slot_pixel(w, slot, tile, x, y);
        active = x < w.W && y < w.H && tile_active(w, pp, tile);
    }
    if (slot < w.P) w.flags[slot] = 0;
    if (active) {
        Rng rng; rng.key = rtx_rng_key(pp.st.seed, RTX_STREAM_CAMERA, y * w.W + x, pp.sample_index); rng.dim = 0;
        float px, py;
        if (pp.st.draw_mode != RTX_DM_ALGORITHM) { px = x + 0.5f; py = y + 0.5f; }
        else { px = x + rng.next(); py = y + rng.next(); }
        float dx, dy, dz; camera_ray(cam, px, py, dx, dy, dz);
        rtx_ray r; r.o[0] = cam.origin[0]; r.o[1] = cam.origin[1]; r.o[2] = cam.origin[2]; r.d[0] = dx; r.d[1] = dy; r.d[2] = dz;
        w.ray[slot] = r;
        w.xy[2 * slot] = px; w.xy[2 * slot + 1] = py;
        w.key[slot] = rng.key; w.dim[slot] = rng.dim; w.mis[slot] = 0.0f;
        w.T[3 * slot] = 1.0f; w.T[3 * slot + 1] = 1.0f; w.T[3 * slot + 2] = 1.0f;
        w.L[3 * slot] = 0.0f; w.L[3 * slot + 1] = 0.0f; w.L[3 * slot + 2] = 0.0f;
        w.flags[slot] = kFlagStarted | (pp.st.can_hit_light ? kFlagCanHitLight : 0u);
    }
    // place in the queue: the tile's offset (k_tile_offsets) + rank inside the tile, no atomics
    __shared__ uint32_t s_count[kBlock / 64];
    const unsigned long long mask = __ballot(active);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) s_count[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (active) {
        uint32_t q = w.tile_off[blockIdx.x] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        for (uint32_t i = 0; i < wave; i++) q += s_count[i];
        w.qa[q] = slot;
    }
}

// Queue offset of every tile's first camera sample (tiles are skipped by adaptive sampling
// and by the tile split across GPUs) and the queue length; one workgroup.
__global__ __launch_bounds__(1024) void k_tile_offsets(WfPtrs w, PassParams pp)
{
    __shared__ uint32_t s_part[1024];
    const uint32_t per = (w.n_local_tiles + 1023u) / 1024u, t0 = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t lt = t0 + k, tile = lt * w.tile_mul + w.tile_add;
        if (lt < w.n_local_tiles && tile_active(w, pp, tile)) {
            const uint32_t sx = (tile % w.tiles_x) * 16u, sy = (tile / w.tiles_x) * 16u;
            sum += (min(sx + 16u, w.W) - sx) * (min(sy + 16u, w.H) - sy);
        }
    }
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? s_part[threadIdx.x - o] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t lt = t0 + k, tile = lt * w.tile_mul + w.tile_add;
        if (lt < w.n_local_tiles) {
            w.tile_off[lt] = run;
            if (tile_active(w, pp, tile)) {
                const uint32_t sx = (tile % w.tiles_x) * 16u, sy = (tile / w.tiles_x) * 16u;
                run += (min(sx + 16u, w.W) - sx) * (min(sy + 16u, w.H) - sy);
            }
        }
    }
    if (threadIdx.x == 1023u) w.cnt[(0) * kQueueStride] = s_part[1023];
}

// The items of a shade kernel's workgroup regrouped by class (miss, emitter, specular surface, one class per BSDF that
// samples a light). After the first bounce the queue holds paths in arrival order, and a wave whose 64 paths are a few
// surface hits among misses walks the long chain of dependent loads of a surface hit (triangle record, material, texel,
// light, CDF rows and columns, environment texels) for a handful of lanes and the short one of a miss for the others.
// Sorted, the waves of a workgroup are pure but for the ones at class boundaries. A counting sort: ballots per class,
// per-wave counts and a prefix in LDS; the path slot and the hit record travel through LDS to the lane that shades them.
struct ShadeRegroup {
    uint32_t slot[kBlock], id[kBlock]; float t[kBlock], alpha[kBlock], beta[kBlock], gamma[kBlock];
    uint32_t count[kBlock / 64][8], base[kBlock / 64][8], total;
};
// every thread of the workgroup calls this from uniform control flow; returns whether the thread has an item afterwards
__device__ __forceinline__ bool regroup_by_class(ShadeRegroup& s, bool active, uint32_t cls, uint32_t& slot, rtx_hit& h)
{
    static_assert(kClassCount <= 8, "eight counters per wave");
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t rank = 0;
#pragma unroll
    for (uint32_t c = 0; c < 8u; c++) {
        const unsigned long long m = __ballot(active && cls == c);
        if (lane == c) s.count[wave][c] = (uint32_t)__popcll(m);
        if (cls == c) rank = (uint32_t)__popcll(m & below);
    }
    __syncthreads();
    if (threadIdx.x < (kBlock / 64) * 8) {
        const uint32_t wv = threadIdx.x >> 3, c = threadIdx.x & 7u;
        uint32_t b = 0;
        for (uint32_t cc = 0; cc < c; cc++) for (uint32_t ww = 0; ww < kBlock / 64; ww++) b += s.count[ww][cc];
        for (uint32_t ww = 0; ww < wv; ww++) b += s.count[ww][c];
        s.base[wv][c] = b;
        if (threadIdx.x == (kBlock / 64) * 8 - 1) s.total = b + s.count[wv][c];      // last wave, last class: everything before it + its own
    }
    __syncthreads();
    if (active) {
        const uint32_t p = s.base[wave][cls] + rank;
        s.slot[p] = slot; s.id[p] = h.id; s.t[p] = h.t; s.alpha[p] = h.alpha; s.beta[p] = h.beta; s.gamma[p] = h.gamma;
    }
    __syncthreads();
    const bool mine = threadIdx.x < s.total;
    if (mine) { const uint32_t p = threadIdx.x; slot = s.slot[p]; h.id = s.id[p]; h.t = s.t[p]; h.alpha = s.alpha[p]; h.beta = s.beta[p]; h.gamma = s.gamma[p]; }
    return mine;
}

// What shading one path leaves for the queues: whether it goes on (its new ray and state are in the slot) and whether
// it asks for a shadow ray (segment sx -> p2, value sval the path gains if the segment is free).
struct ShadeOut { bool cont, shadow; V3 sx, p2; C3 sval; };
// One depth of RayTracer::pathTrace, RTBase/Renderer.h:476-526, for the path in `slot` whose ray found hit `h`.
// FIRST: depth 0, the path's state is the camera sample of the slot (nothing stored yet).
template <bool FIRST>
__device__ __forceinline__ void pt_shade_item(const WfPtrs& w, const DeviceScene& sc, const CameraDev& cam, const PassParams& pp, uint32_t slot, const rtx_hit& h, ShadeOut& out)
{
    bool& cont = out.cont; bool& shadow = out.shadow; V3& p2 = out.p2; C3& sval = out.sval; V3& sx = out.sx;
    cont = false; shadow = false;
    V3 o, d; C3 T, L; uint32_t flags, rkey, rdim; float misPdf;
    if constexpr (FIRST) {
        // the state ray generation would have stored (Renderer.h:730-736): the camera ray of the slot's sample, unit
        // throughput, no radiance yet, the generator behind the two jitter draws
        float px, py; Rng crng;
        camera_sample(w, pp, slot, px, py, crng);
        float dx, dy, dz; camera_ray(cam, px, py, dx, dy, dz);
        o = ld3(cam.origin); d = mk(dx, dy, dz);
        T = mkc(1.0f, 1.0f, 1.0f); L = mkc(0.0f, 0.0f, 0.0f);
        flags = kFlagStarted | (pp.st.can_hit_light ? kFlagCanHitLight : 0u); misPdf = 0.0f; rkey = crng.key; rdim = crng.dim;
    } else {
        const rtx_ray r = w.ray[slot];
        T = ldc(w.T + 3 * slot); L = ldc(w.L + 3 * slot);
        flags = w.flags[slot]; misPdf = w.mis[slot];
        rkey = w.key[slot]; rdim = w.dim[slot];
        // the whole path state in one round trip: without this the compiler loads each piece where it is first
        // read, behind the triangle record, the material and the texture, one dependent round trip each
        asm volatile("" :: "v"(T.r), "v"(T.g), "v"(T.b), "v"(L.r), "v"(L.g), "v"(L.b), "v"(flags), "v"(misPdf), "v"(rkey), "v"(rdim));
        o = ld3(r.o); d = ld3(r.d);
    }
    if (h.t < FLT_MAX) {
        Shading sd; shading_data(sc, h, o, d, sd);
        if (sd.mat->is_light) {
            if (flags & kFlagCanHitLight) { L = L + T * ldc(sd.mat->emission); if constexpr (!FIRST) st3(w.L + 3 * slot, L); }
        } else {
            Rng rng; rng.key = rkey; rng.dim = rdim;
            C3 dv;
            if (direct_request(sc, sd, rng, pp.st.use_mis != 0, misPdf, p2, dv)) { shadow = true; sval = T * dv; sx = sd.x; }
            if (pp.depth < pp.st.max_bounces) {
                float rr = wmin(lum(T), 0.9f);
                if (!(rr < rng.next())) {
                    T = T / rr;
                    C3 bs; float pdf; V3 wi = bsdf_sample(sc, sd, rng, bs, pdf);
                    T = T * bs * fabsf(dot(wi, sd.sNormal)) / pdf;
                    V3 no = sd.x + (wi * kEps);
                    rtx_ray nr; nr.o[0] = no.x; nr.o[1] = no.y; nr.o[2] = no.z; nr.d[0] = wi.x; nr.d[1] = wi.y; nr.d[2] = wi.z;
                    w.ray[slot] = nr; st3(w.T + 3 * slot, T); w.mis[slot] = pdf;
                    w.flags[slot] = (flags & ~kFlagCanHitLight) | (isPureSpecular(*sd.mat) ? kFlagCanHitLight : 0u);
                    if constexpr (FIRST) w.key[slot] = rkey;
                    cont = true;
                }
            }
            w.dim[slot] = rng.dim;
        }
    } else {
        // Renderer.h:519-525
        if (pp.depth == 0) L = L + background_evaluate(sc, d);
        else {
            float pdf = background_pdf(sc, d);
            float weight = pp.st.use_mis ? pdf / (pdf + misPdf) : 1.0f;
            L = L + background_evaluate(sc, d) * T * weight / pdf;
        }
        if constexpr (!FIRST) st3(w.L + 3 * slot, L);
    }
    // depth 0 starts the sample's radiance sum (the shadow rays and the later depths add to it, the film kernel reads it)
    if constexpr (FIRST) st3(w.L + 3 * slot, L);
}

// One depth of RayTracer::pathTrace, RTBase/Renderer.h:476-526, for every live path: one queue position per thread.
// FIRST: the queue of camera samples (dense, its length in the queue word of depth 0); otherwise a streamed queue.
// REGROUP: the workgroup sorts its items by class first (camera rays arrive in pixel-block order and are coherent as
// they are). The launch covers every position the queue can have; workgroups beyond its end leave at once.
// Survivors and shadow rays go to the streamed queues of the next depth, one atomic per wave and no barrier. Wave w
// of workgroup b appends to stream 4 * (b % 4) + (w + b / 4) % 4: every stream gets exactly one wave of every fourth
// workgroup, so a stream never grows beyond 64 entries per 1024 input positions (the output needs no more positions
// than the input had, P + kQueueSlack bounds them all), and the wave that holds the sorted workgroup's surface hits
// moves round the streams of its group instead of filling one.
template <bool FIRST, bool REGROUP>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pt_shade(WfPtrs w, DeviceScene sc, CameraDev cam, PassParams pp, const uint32_t* __restrict__ qin, uint32_t* qout)
{
    __shared__ ShadeRegroup s_regroup;
    __shared__ StreamLens s_in;
    uint32_t extent;
    if constexpr (FIRST) extent = w.P; else extent = stream_load(w.cnt, pp.depth, s_in);
    if (blockIdx.x * (uint32_t)kBlock >= extent) return;
    {
        const uint32_t i = global_thread();
        bool cont = false, shadow = false; uint32_t slot = 0; V3 p2 = mk(0, 0, 0); C3 sval = mkc(0, 0, 0); V3 sx = mk(0, 0, 0);
        bool have = i < extent; rtx_hit h; h.id = 0; h.t = FLT_MAX; h.alpha = h.beta = h.gamma = 0.0f;
        if constexpr (FIRST) {
            // depth 0: position i is path slot i (no queue); whether it carries a sample is decided again below, with the sample itself
            float px, py; Rng crng;
            have = camera_sample(w, pp, i, px, py, crng);
            slot = i;
            if (have) h = w.hit[slot];
        } else {
            have = have && stream_valid(s_in, i);
            if (have) { slot = qin[i]; h = w.hit[slot]; }
        }
        if constexpr (REGROUP) {
            uint32_t cls = kClassMiss;
            if (have && h.t < FLT_MAX) cls = sc.shade[h.id].shade_class;
            have = regroup_by_class(s_regroup, have, cls, slot, h);
        }
        ShadeOut so; so.cont = so.shadow = false; so.sx = so.p2 = mk(0, 0, 0); so.sval = mkc(0, 0, 0);
        if (have) pt_shade_item<FIRST>(w, sc, cam, pp, slot, h, so);
        cont = so.cont; shadow = so.shadow; p2 = so.p2; sval = so.sval; sx = so.sx;
        const uint32_t stream = ((blockIdx.x & 3u) << 2) | (((threadIdx.x >> 6) + (blockIdx.x >> 2)) & 3u);
        const uint32_t qs = stream_push(shadow, w.cnt + stream_word(kShadowCounterBase + pp.depth, stream), stream);
        if (shadow) {
            float* s = w.sseg + 6 * (size_t)qs; s[0] = sx.x; s[1] = sx.y; s[2] = sx.z; s[3] = p2.x; s[4] = p2.y; s[5] = p2.z;
            st3(w.sval + 3 * (size_t)qs, sval); w.spath[qs] = slot;
        }
        const uint32_t qc = stream_push(cont, w.cnt + stream_word(pp.depth + 1, stream), stream);
        if (cont) qout[qc] = slot;
    }
}

// Shadow rays of one depth: a visible segment adds its value to the path's radiance
// (Renderer.h:448-452, 463-469).
struct ShadowAccumulatePolicy {
    const WfPtrs& w;
    __device__ __forceinline__ bool fetch(uint32_t k, float o[3], float d[3], float& maxT) const { segment_to_ray(w.sseg + 6 * (size_t)k, o, d, maxT); return true; }
    __device__ __forceinline__ void retire(uint32_t k, const HitRec& best, bool reached) const
    {
        if (reached && best.pos == 0xFFFFFFFFu) {
            const uint32_t slot = w.spath[k];
            C3 L = ldc(w.L + 3 * slot) + ldc(w.sval + 3 * (size_t)k);
            st3(w.L + 3 * slot, L);
        }
    }
    __device__ __forceinline__ void note(uint32_t, uint32_t, uint32_t) const {}
};
// COUNT (here and in the kernels below): the instrumented variant behind rtx_count_render_work — same results, plus
// the nodes visited / triangles tested by one lane per ray (no helpers) added to counters[0..1]
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_shadow_accumulate(WfPtrs w, DeviceScene sc, uint32_t counter_index, uint32_t work_index, uint2* spill, uint32_t spill_stride,
                                                              unsigned long long* counters)
{
    __shared__ unsigned long long lds[kBlock * kLdsStack];
    LaneStack<kBlock, kLdsStack> stk; stk.init(lds, spill, spill_stride);
    ShadowAccumulatePolicy pol{w};
    __shared__ TailShare<kBlock> ts;
    traverse_dyn<ShadowAccumulatePolicy, kModeAny, COUNT, kRefill, kBlock, kLdsStack, kTriUnroll, kLeafLanes, COUNT ? 0 : kShare>(sc, stk, w.cnt[(counter_index) * kQueueStride], w.cnt + work_index, pol, counters, &ts);
}
// closest hits of a queue with the work counters on (the plain policy's note() keeps per-ray counts in the hit record,
// which a render pass needs intact)
struct ClosestQuietPolicy : ClosestPolicy {
    __device__ __forceinline__ void note(uint32_t, uint32_t, uint32_t) const {}
};
__global__ __launch_bounds__(kBlock) void k_closest_counted(DeviceScene sc, const rtx_ray* __restrict__ rays, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ n_dev,
                                                            rtx_hit* __restrict__ hits, uint2* spill, uint32_t spill_stride, uint32_t* work_counter, unsigned long long* counters)
{
    __shared__ unsigned long long lds[kBlock * kLdsStack];
    LaneStack<kBlock, kLdsStack> stk; stk.init(lds, spill, spill_stride);
    ClosestQuietPolicy pol{{sc, rays, queue, hits}};
    __shared__ TailShare<kBlock> ts;
    traverse_dyn<ClosestQuietPolicy, kModeClosest, true, kRefillClosest, kBlock, kLdsStack, kTriUnroll, kLeafLanesClosest, 0>(sc, stk, *n_dev, work_counter, pol, counters, &ts);
}

// The shadow rays of depth d - 1 and the closest-hit rays of depth d in ONE launch: neither depends on the other
// (both come out of the shade kernel of depth d - 1), late depths hold few rays, and a traversal launch lasts at
// least as long as its slowest ray, so one launch for both halves the number of such floors per pass. Items
// [0, n_closest) are the live paths of `queue`, items from n_closest on the shadow segments.
struct PtMixedPolicy {
    const DeviceScene& sc; const WfPtrs& w; const uint32_t* queue; uint32_t n_closest; const StreamLens& live; const StreamLens& segs;
    __device__ __forceinline__ bool fetch(uint32_t k, float o[3], float d[3], float& maxT, bool& any) const
    {
        o[0] = o[1] = o[2] = 0.0f; d[0] = d[1] = d[2] = 0.0f; maxT = 0.0f;
        if (k < n_closest) {
            any = false;
            if (!stream_valid(live, k)) return false;              // a hole of the streamed queue
            const rtx_ray r = w.ray[queue[k]];
            o[0] = r.o[0]; o[1] = r.o[1]; o[2] = r.o[2]; d[0] = r.d[0]; d[1] = r.d[1]; d[2] = r.d[2];
        } else {
            any = true;
            if (!stream_valid(segs, k - n_closest)) return false;
            segment_to_ray(w.sseg + 6 * (size_t)(k - n_closest), o, d, maxT);
        }
        return true;
    }
    __device__ __forceinline__ void retire(uint32_t k, const HitRec& best, bool reached) const
    {
        if (k < n_closest) w.hit[queue[k]] = make_hit(sc, best);
        else if (reached && best.pos == 0xFFFFFFFFu) {
            const uint32_t j = k - n_closest, slot = w.spath[j];
            C3 L = ldc(w.L + 3 * slot) + ldc(w.sval + 3 * (size_t)j);
            st3(w.L + 3 * slot, L);
        }
    }
    __device__ __forceinline__ void note(uint32_t, uint32_t, uint32_t) const {}
};
// (the per-item flag costs registers: held to the 96 that keep five waves per SIMD like the other traversal kernels; +3 % on a 1080p pass)
// depth = max_bounces + 1: the live queue of that depth is empty, the launch walks the shadow rays of the last depth
template <bool COUNT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_pt_trace_mixed(WfPtrs w, DeviceScene sc, uint32_t depth, const uint32_t* __restrict__ queue, uint32_t work_index,
                                                           uint2* spill, uint32_t spill_stride, unsigned long long* counters)
{
    __shared__ unsigned long long lds[kBlock * kLdsStack];
    __shared__ StreamLens s_live, s_segs;
    LaneStack<kBlock, kLdsStack> stk; stk.init(lds, spill, spill_stride);
    const uint32_t n_closest = stream_load(w.cnt, depth, s_live), n_shadow = stream_load(w.cnt, kShadowCounterBase + depth - 1, s_segs);
    PtMixedPolicy pol{sc, w, queue, n_closest, s_live, s_segs};
    __shared__ TailShare<kBlock> ts;
    traverse_dyn<PtMixedPolicy, kModeMixed, COUNT, kRefill, kBlock, kLdsStack, kTriUnroll, kLeafLanes, COUNT ? 0 : kShare>(sc, stk, n_closest + n_shadow, w.cnt + work_index, pol, counters, &ts);
}

// The late depths of a path-tracing pass in ONE launch. Behind the second bounce a 1080p pass holds a few percent of
// its paths, and as a wavefront every depth still costs a traversal launch that lasts as long as its slowest ray plus a
// shade launch that lasts as long as its chain of dependent loads: ~350 us for 3 % of the rays. Here a WAVE takes a
// block of 64 positions of the live queue of depth pp.depth (their closest hits are known) and carries those paths to
// their ends by itself: shade, then one wave-local traversal for the shadow rays just asked for and the next rays of the
// survivors (traverse_dyn's LOCAL mode; idle lanes help the busy ones), shade again, ... No queue leaves the wave:
// the lists (live slots, their hits, pending shadow segments) sit in LDS, the path state stays in the slot arrays.
// Every sum is taken in the order the wavefront kernels take it (a path's shadow ray of depth d before its emission at
// depth d + 1), so the film is the same bit for bit whichever way the late depths run.
template <int N>
struct TailWaveLists {
    uint32_t live[N];                                                  // path slots that go on, in list order
    uint32_t hit_id[N]; float hit_t[N], hit_alpha[N], hit_beta[N], hit_gamma[N];   // closest hit of live[k]
    float seg[6][N], val[3][N]; uint32_t seg_slot[N];                  // shadow rays asked for by the last shade step
};
template <int N>
struct PtTailPolicy {
    const DeviceScene& sc; const WfPtrs& w; TailWaveLists<N>& t; uint32_t n_live;
    __device__ __forceinline__ bool fetch(uint32_t k, float o[3], float d[3], float& maxT, bool& any) const
    {
        if (k < n_live) {
            const rtx_ray r = w.ray[t.live[k]];
            o[0] = r.o[0]; o[1] = r.o[1]; o[2] = r.o[2]; d[0] = r.d[0]; d[1] = r.d[1]; d[2] = r.d[2]; maxT = 0.0f; any = false;
        } else {
            const uint32_t j = k - n_live;
            const float sg[6] = {t.seg[0][j], t.seg[1][j], t.seg[2][j], t.seg[3][j], t.seg[4][j], t.seg[5][j]};
            segment_to_ray(sg, o, d, maxT); any = true;
        }
        return true;
    }
    __device__ __forceinline__ void retire(uint32_t k, const HitRec& best, bool reached) const
    {
        if (k < n_live) {
            const rtx_hit h = make_hit(sc, best);
            t.hit_id[k] = h.id; t.hit_t[k] = h.t; t.hit_alpha[k] = h.alpha; t.hit_beta[k] = h.beta; t.hit_gamma[k] = h.gamma;
        } else if (reached && best.pos == 0xFFFFFFFFu) {
            const uint32_t j = k - n_live, slot = t.seg_slot[j];
            C3 L = ldc(w.L + 3 * slot) + mkc(t.val[0][j], t.val[1][j], t.val[2][j]);
            st3(w.L + 3 * slot, L);
        }
    }
    __device__ __forceinline__ void note(uint32_t, uint32_t, uint32_t) const {}
};
constexpr int kTailRefill = 8;       // idle lanes at which a wave takes more of its own items
// ROUNDS: a wave's task is ROUNDS blocks of 64 queue positions; it shades them 64 at a time and walks all their rays in
// one wave-local traversal (fuller lanes per trip, fewer waves resident: the lists take 4 KB of LDS per block).
template <int ROUNDS>
__global__ __launch_bounds__(kBlock) void k_pt_tail(WfPtrs w, DeviceScene sc, CameraDev cam, PassParams pp, const uint32_t* __restrict__ qin, uint32_t task_index,
                                                    uint2* spill, uint32_t spill_stride RTX_PACKET_STATS(, unsigned long long* stats))
{
    constexpr uint32_t kTask = 64u * ROUNDS;
    RTX_PACKET_STATS(unsigned long long t_shade = 0, t_trace = 0, t_begin = stats ? wall_clock64() : 0ull; uint32_t n_rounds = 0, n_tasks_done = 0;)
    __shared__ unsigned long long lds[kBlock * kLdsStack];
    __shared__ TailShare<kBlock> ts;
    __shared__ StreamLens s_in;
    __shared__ TailWaveLists<64 * ROUNDS> s_lists[kBlock / 64];
    LaneStack<kBlock, kLdsStack> stk; stk.init(lds, spill, spill_stride);
    const uint32_t first_depth = pp.depth;
    const uint32_t n_tasks = stream_load(w.cnt, first_depth, s_in) / kTask;    // (the only barrier: from here on the waves go their own ways)
    TailWaveLists<64 * ROUNDS>& t = s_lists[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    // A wave's first task is the one of its own number; further ones (a queue longer than one round of the grid's waves)
    // come from a counter. Thousands of waves asking one counter at the start of the launch would be served one after the
    // other, ~11 ns each: the last wave of the grid would start tens of microseconds late.
    const uint32_t n_waves = gridDim.x * (uint32_t)(kBlock / 64);
    uint32_t task = blockIdx.x * (uint32_t)(kBlock / 64) + (threadIdx.x >> 6);
    for (;; ) {
        if (task >= n_tasks) break;
        uint32_t n_in = kTask;                                          // items to shade at this depth: queue positions first, the live list afterwards
        for (pp.depth = first_depth; ; pp.depth++) {
            uint32_t n_live = 0, n_seg = 0;
            RTX_PACKET_STATS(const unsigned long long c0 = stats ? wall_clock64() : 0ull;)
            for (uint32_t base = 0; base < n_in; base += 64u) {
                bool have; uint32_t slot = 0; rtx_hit h; h.id = 0; h.t = FLT_MAX; h.alpha = h.beta = h.gamma = 0.0f;
                if (pp.depth == first_depth) {
                    const uint32_t i = task * kTask + base + lane;
                    have = stream_valid(s_in, i);
                    if (have) { slot = qin[i]; h = w.hit[slot]; }
                } else {
                    const uint32_t k = base + lane;
                    have = k < n_in;
                    if (have) { slot = t.live[k]; h.id = t.hit_id[k]; h.t = t.hit_t[k]; h.alpha = t.hit_alpha[k]; h.beta = t.hit_beta[k]; h.gamma = t.hit_gamma[k]; }
                }
                ShadeOut so; so.cont = so.shadow = false; so.sx = so.p2 = mk(0, 0, 0); so.sval = mkc(0, 0, 0);
                if (have) pt_shade_item<false>(w, sc, cam, pp, slot, h, so);
                // survivors are packed to the front of the live list: they never reach the entries a later round still has to read
                // (at most `base` of them so far), and this round has read its own
                const unsigned long long mc = __ballot(so.cont), ms = __ballot(so.shadow);
                if (so.cont) t.live[n_live + (uint32_t)__popcll(mc & below)] = slot;
                if (so.shadow) {
                    const uint32_t j = n_seg + (uint32_t)__popcll(ms & below);
                    t.seg[0][j] = so.sx.x; t.seg[1][j] = so.sx.y; t.seg[2][j] = so.sx.z; t.seg[3][j] = so.p2.x; t.seg[4][j] = so.p2.y; t.seg[5][j] = so.p2.z;
                    t.val[0][j] = so.sval.r; t.val[1][j] = so.sval.g; t.val[2][j] = so.sval.b; t.seg_slot[j] = slot;
                }
                n_live += (uint32_t)__popcll(mc); n_seg += (uint32_t)__popcll(ms);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            }
            RTX_PACKET_STATS(if (stats) { t_shade += wall_clock64() - c0; n_rounds++; })
            if (n_live + n_seg == 0u) break;
            // the lists and the slots' new state were written by other lanes than the ones that read them next
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            PtTailPolicy<64 * ROUNDS> pol{sc, w, t, n_live};
            RTX_PACKET_STATS(const unsigned long long c1 = stats ? wall_clock64() : 0ull;)
            traverse_dyn<PtTailPolicy<64 * ROUNDS>, kModeMixed, false, kTailRefill, kBlock, kLdsStack, kTriUnroll, kLeafLanes, kShare, true>(sc, stk, n_live + n_seg, nullptr, pol, nullptr, &ts);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            RTX_PACKET_STATS(if (stats) t_trace += wall_clock64() - c1;)
            if (n_live == 0u) break;                                   // the shadow rays of the paths' last vertices
            n_in = n_live;
        }
        RTX_PACKET_STATS(n_tasks_done++;)
        if (n_tasks <= n_waves) break;                                  // (every task had a wave of its own)
        uint32_t next = 0;
        if (lane == 0) next = atomicAdd(w.cnt + task_index, 1u);
        task = n_waves + (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
    }
    // (tuning build: clock ticks per wave in shade steps and in traversals, waves with a task, their tasks and rounds, the longest wave)
    RTX_PACKET_STATS(if (stats && lane == 0 && n_tasks_done) { atomicAdd(&stats[16], t_shade); atomicAdd(&stats[17], t_trace); atomicAdd(&stats[18], 1ull); atomicAdd(&stats[19], (unsigned long long)n_tasks_done);
                     atomicAdd(&stats[20], (unsigned long long)n_rounds); atomicMax(&stats[21], wall_clock64() - t_begin); atomicMax(&stats[22], t_shade); atomicMax(&stats[23], t_trace); })
}

// Closest hits of the camera rays of a path-tracing pass, RTBase/Renderer.h:730-736: item k is path slot k, and its ray
// is generated where it is fetched (Camera::generateRay through the slot's jittered sample), so a pass stores no camera
// rays and runs no ray-generation kernel. Slots without a sample this pass are done at fetch.
template <bool COUNT>
struct PtPrimaryPolicy {
    const DeviceScene& sc; const WfPtrs& w; const CameraDev& cam; const PassParams& pp;
    __device__ __forceinline__ bool fetch(uint32_t k, float o[3], float d[3], float& maxT) const
    {
        o[0] = cam.origin[0]; o[1] = cam.origin[1]; o[2] = cam.origin[2]; d[0] = d[1] = d[2] = 0.0f; maxT = 0.0f;
        float px, py; Rng rng;
        if (!camera_sample(w, pp, k, px, py, rng)) return false;
        camera_ray(cam, px, py, d[0], d[1], d[2]);
        // instrumented runs: the number of camera rays, kept where a dense queue of depth 0 would keep its length
        if constexpr (COUNT) atomicAdd(&w.cnt[(0) * kQueueStride], 1u);
        return true;
    }
    __device__ __forceinline__ void retire(uint32_t k, const HitRec& best, bool) const { w.hit[k] = make_hit(sc, best); }
    __device__ __forceinline__ void note(uint32_t, uint32_t, uint32_t) const {}
};
// (held to the registers that keep five waves per SIMD, like the other traversal kernels: the ray set-up at fetch would take 106)
template <bool COUNT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_pt_closest_primary(WfPtrs w, DeviceScene sc, CameraDev cam, PassParams pp, uint32_t work_index, uint2* spill, uint32_t spill_stride,
                                                               unsigned long long* counters)
{
    __shared__ unsigned long long lds[kBlock * kLdsStack];
    LaneStack<kBlock, kLdsStack> stk; stk.init(lds, spill, spill_stride);
    PtPrimaryPolicy<COUNT> pol{sc, w, cam, pp};
    __shared__ TailShare<kBlock> ts;
    traverse_dyn<PtPrimaryPolicy<COUNT>, kModeClosest, COUNT, kRefillClosest, kBlock, kLdsStack, kTriUnroll, kLeafLanesClosest, COUNT ? 0 : kShare>(sc, stk, w.P, w.cnt + work_index, pol, counters, &ts);
}

// film->splat(px, py, col) for every sample of the pass, Renderer.h:745. CAMERA: the sample is derived from the slot
// number (path tracing, see camera_sample); otherwise ray generation has stored it.
// What the host wants to know about a path-tracing pass before it enqueues a later one: how many queue positions every
// depth had (it starts the tail kernel at the first depth that fits one round of it). Written to pinned host memory by
// the last kernel of the pass; the host reads it whenever it gets there, no synchronisation (any depth is correct).
__device__ __forceinline__ void publish_depth_stats(const WfPtrs& w)
{
    if (blockIdx.x != 0 || threadIdx.x >= 96u || !w.depth_stats) return;
    uint32_t blocks = 0;
    for (int c = 0; c < kQueueStreams; c++) blocks = max(blocks, (w.cnt[stream_word(threadIdx.x, (uint32_t)c)] + 63u) >> 6);
    w.depth_stats[threadIdx.x] = blocks * (uint32_t)(kQueueStreams * 64);
}

template <bool CAMERA>
__global__ __launch_bounds__(kBlock) void k_film_accumulate(WfPtrs w, PassParams pp, int filter)
{
    if constexpr (CAMERA) publish_depth_stats(w);
    const uint32_t slot = global_thread();
    float fx, fy;
    if constexpr (CAMERA) { Rng rng; if (!camera_sample(w, pp, slot, fx, fy, rng)) return; }
    else { if (slot >= w.P || !(w.flags[slot] & kFlagStarted)) return; fx = w.xy[2 * slot]; fy = w.xy[2 * slot + 1]; }
    film_splat(w.film, (int)w.W, (int)w.H, filter, fx, fy, ldc(w.L + 3 * slot));
}

// Box filter (Film::splat with size 0: one pixel, weight 1, RTBase/Imaging.h:318-340): the sample of slot s lands in
// the pixel it was generated for, unless x + r rounded up to the next integer. Phase 0 adds the samples that stay in
// their own pixel with plain loads and stores (one sample per pixel per pass, nobody else writes it in this
// launch); phase 1, a second launch, adds the few strays with float atomics.
// clean (second launch only): the pass is done with its counters (queue lengths, work counters) and zeroes them for the
// next pass on the same buffers, which then needs no memset of its own
template <bool CAMERA>
__global__ __launch_bounds__(kBlock) void k_film_accumulate_box(WfPtrs w, PassParams pp, int phase, int clean)
{
    if constexpr (CAMERA) { if (phase == 0) publish_depth_stats(w); }
    if (phase == 1 && clean) for (uint32_t i = global_thread(); i < (uint32_t)kCounterWords; i += gridDim.x * (uint32_t)kBlock) w.cnt[i] = 0u;
    const uint32_t slot = global_thread();
    float fx, fy;
    if constexpr (CAMERA) { Rng rng; if (!camera_sample(w, pp, slot, fx, fy, rng)) return; }
    else { if (slot >= w.P || !(w.flags[slot] & kFlagStarted)) return; fx = w.xy[2 * slot]; fy = w.xy[2 * slot + 1]; }
    uint32_t tile, x, y;
    slot_pixel(w, slot, tile, x, y);
    const int px = (int)fx, py = (int)fy;
    const bool own = px == (int)x && py == (int)y;
    if (own != (phase == 0)) return;
    if (px < 0 || px >= (int)w.W || py < 0 || py >= (int)w.H) return;
    // total weight 1, so v = L * 1 / 1 as film_splat computes it
    const C3 v = ldc(w.L + 3 * slot) * 1.0f / 1.0f;
    float* f = w.film + 3 * ((size_t)py * w.W + px);
    if (phase == 0) { f[0] += v.r; f[1] += v.g; f[2] += v.b; }
    else { atomicAdd(f, v.r); atomicAdd(f + 1, v.g); atomicAdd(f + 2, v.b); }
}

// camera: the samples are derived from the slot numbers (path tracing) instead of read from what ray generation stored
inline void launch_film_accumulate(const WfPtrs& w, const PassParams& pp, bool camera, int filter, dim3 grid, dim3 blk, hipStream_t s, bool clean_counters = false)
{
    if (filter == (int)RTX_FT_BOX) {
        for (int phase = 0; phase < 2; phase++) {
            if (camera) hipLaunchKernelGGL(k_film_accumulate_box<true>, grid, blk, 0, s, w, pp, phase, clean_counters ? 1 : 0);
            else hipLaunchKernelGGL(k_film_accumulate_box<false>, grid, blk, 0, s, w, pp, phase, 0);
        }
    } else if (camera) hipLaunchKernelGGL(k_film_accumulate<true>, grid, blk, 0, s, w, pp, filter);
    else hipLaunchKernelGGL(k_film_accumulate<false>, grid, blk, 0, s, w, pp, filter);
}

// ---- draw modes: RayTracer::direct / albedo / viewNormals, RTBase/Renderer.h:536-575 -------
__global__ __launch_bounds__(kBlock) void k_dm_shade(WfPtrs w, DeviceScene sc, PassParams pp)
{
    const uint32_t n = w.cnt[(0) * kQueueStride];
    const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
    for (uint32_t it = 0; it < rounds; it++) {
        const uint32_t i = it * gridDim.x * kBlock + global_thread();
        bool shadow = false; uint32_t slot = 0; V3 p2 = mk(0, 0, 0), sx = mk(0, 0, 0); C3 sval = mkc(0, 0, 0);
        if (i < n) {
            slot = w.qa[i];
            const rtx_hit h = w.hit[slot]; const rtx_ray r = w.ray[slot];
            const V3 o = ld3(r.o), d = ld3(r.d);
            C3 L = mkc(0, 0, 0);
            if (h.t < FLT_MAX) {
                Shading sd; shading_data(sc, h, o, d, sd);
                if (pp.st.draw_mode == RTX_DM_NORMALS) L = mkc(fabsf(sd.sNormal.x), fabsf(sd.sNormal.y), fabsf(sd.sNormal.z));
                else if (sd.mat->is_light) L = ldc(sd.mat->emission);
                else if (pp.st.draw_mode == RTX_DM_ALBEDO) L = bsdf_evaluate(sc, sd, mk(0, 1, 0));
                else {
                    Rng rng; rng.key = w.key[slot]; rng.dim = w.dim[slot];
                    C3 dv;
                    if (direct_request(sc, sd, rng, pp.st.use_mis != 0, 0.0f, p2, dv)) { shadow = true; sval = dv; sx = sd.x; }
                    w.dim[slot] = rng.dim;
                }
            } else if (pp.st.draw_mode != RTX_DM_NORMALS) L = background_evaluate(sc, d);
            st3(w.L + 3 * slot, L);
        }
        uint32_t qs = queue_slot(shadow, &w.cnt[(kShadowCounterBase) * kQueueStride]);
        if (shadow) {
            float* s = w.sseg + 6 * (size_t)qs; s[0] = sx.x; s[1] = sx.y; s[2] = sx.z; s[3] = p2.x; s[4] = p2.y; s[5] = p2.z;
            st3(w.sval + 3 * (size_t)qs, sval); w.spath[qs] = slot;
        }
    }
}

// ---- light tracing: RTBase/Renderer.h:308-411 ----------------------------------------------

// Camera::projectOntoCamera, RTBase/Scene.h:61-75
__device__ __forceinline__ bool project_onto_camera(const CameraDev& c, V3 p, float& x, float& y)
{
    const float* m = c.camera_to_view;
    V3 pv = mk((p.x * m[0] + p.y * m[1] + p.z * m[2]) + m[3], (p.x * m[4] + p.y * m[5] + p.z * m[6]) + m[7], (p.x * m[8] + p.y * m[9] + p.z * m[10]) + m[11]);
    const float* q = c.proj;
    V3 v1 = mk((pv.x * q[0] + pv.y * q[1] + pv.z * q[2]) + q[3], (pv.x * q[4] + pv.y * q[5] + pv.z * q[6]) + q[7], (pv.x * q[8] + pv.y * q[9] + pv.z * q[10]) + q[11]);
    float ww = (q[12] * pv.x) + (q[13] * pv.y) + (q[14] * pv.z) + q[15];
    ww = 1.0f / ww;
    V3 pp = v1 * ww;
    x = (pp.x + 1.0f) * 0.5f;
    y = (pp.y + 1.0f) * 0.5f;
    if (x < 0 || x > 1.0f || y < 0 || y > 1.0f) return false;
    x = x * c.width;
    y = 1.0f - y;
    y = y * c.height;
    return true;
}

// RayTracer::connectToCamera, RTBase/Renderer.h:308-334, up to the visibility test
__device__ __forceinline__ bool connect_request(const CameraDev& cam, V3 p, V3 n, C3 col, float& x, float& y, C3& value)
{
    if (!project_onto_camera(cam, p, x, y)) return false;
    V3 toCam = ld3(cam.origin) - p;
    float lsq = lengthSq(toCam);
    toCam = normalize(toCam);
    float cosS = dot(toCam, n), cosC = dot(toCam, ld3(cam.view_dir));
    float g = wmax(cosS, 0.0f) * wmax(-cosC, 0.0f) / lsq;
    if (!(g > 0)) return false;
    float c2 = (cosC * cosC);
    float we = 1.0f / (c2 * c2 * cam.afilm);
    value = col * we * g;
    return true;
}

// the stream the calling wave appends to (wave w of workgroup b: see k_pt_shade)
__device__ __forceinline__ uint32_t wave_stream() { return ((blockIdx.x & 3u) << 2) | (((threadIdx.x >> 6) + (blockIdx.x >> 2)) & 3u); }

// every lane of the wave calls this from uniform control flow (a streamed queue: one atomic per wave, no barrier)
__device__ __forceinline__ void push_connection(const WfPtrs& w, bool want, uint32_t queue, V3 p, const CameraDev& cam, float x, float y, C3 value)
{
    const uint32_t stream = wave_stream();
    const uint32_t qs = stream_push(want, w.cnt + stream_word(queue, stream), stream);
    if (want) {
        float* s = w.sseg + 6 * (size_t)qs; s[0] = p.x; s[1] = p.y; s[2] = p.z; s[3] = cam.origin[0]; s[4] = cam.origin[1]; s[5] = cam.origin[2];
        st3(w.sval + 3 * (size_t)qs, value); w.sxy[2 * (size_t)qs] = x; w.sxy[2 * (size_t)qs + 1] = y;
    }
}

// RayTracer::lightTrace, RTBase/Renderer.h:379-411: one light path per pixel slot
__global__ __launch_bounds__(kBlock) void k_lt_emit(WfPtrs w, DeviceScene sc, CameraDev cam, PassParams pp)
{
    const uint32_t slot = global_thread();
    bool active = false, conn = false; uint32_t tile = 0, x = 0, y = 0; V3 p = mk(0, 0, 0); float fx = 0, fy = 0; C3 cval = mkc(0, 0, 0);
    if (slot < w.P) {
        slot_pixel(w, slot, tile, x, y);
        active = x < w.W && y < w.H && tile_active(w, pp, tile) && sc.n_lights > 0;
        w.flags[slot] = 0;
    }
    if (active) {
        Rng rng; rng.key = rtx_rng_key(pp.st.seed, RTX_STREAM_LIGHT, y * w.W + x, pp.sample_index); rng.dim = 0;
        float pmf; const LightDev& light = sample_light(sc, rng, pmf);
        float pdfPos, pdfDir;
        p = light_sample_position(sc, light, rng, pdfPos);
        V3 wi = light_sample_direction(light, rng, pdfDir);
        V3 nLight = light.kind == 0 ? ld3(light.gnormal) : -wi;                     // Light::normal, Lights.h:60,113,373
        float cosTheta = dot(wi, nLight);
        C3 Le = light_evaluate(sc, light, -wi) / (pmf * pdfPos);
        if (pp.st.can_hit_light) conn = connect_request(cam, p, nLight, Le, fx, fy, cval);
        if (light.kind == 0) Le = Le * cosTheta;
        rtx_ray r; r.o[0] = p.x; r.o[1] = p.y; r.o[2] = p.z; r.d[0] = wi.x; r.d[1] = wi.y; r.d[2] = wi.z;
        w.ray[slot] = r; st3(w.L + 3 * slot, Le);
        w.T[3 * slot] = 1.0f; w.T[3 * slot + 1] = 1.0f; w.T[3 * slot + 2] = 1.0f;
        w.key[slot] = rng.key; w.dim[slot] = rng.dim;
    }
    push_connection(w, conn, kShadowCounterBase + 95, p, cam, fx, fy, cval);
    // lightTracePath returns at once when maxBounces == 0 (Renderer.h:339)
    bool live = active && pp.st.max_bounces > 0;
    const uint32_t stream = wave_stream();
    const uint32_t q = stream_push(live, w.cnt + stream_word(0u, stream), stream);
    if (live) w.qa[q] = slot;
}

// One depth of RayTracer::lightTracePath, RTBase/Renderer.h:336-377
// (one queue position per thread; the launch covers every position the queue can have, workgroups beyond its end leave at once)
// REGROUP: the workgroup sorts its items by class first, as the path tracer's shade kernel does (light paths land all
// over the scene from their first vertex on, so every depth is a mix of materials)
template <bool REGROUP>
__global__ __launch_bounds__(kBlock) void k_lt_shade(WfPtrs w, DeviceScene sc, CameraDev cam, PassParams pp, const uint32_t* __restrict__ qin, uint32_t* qout)
{
    __shared__ ShadeRegroup s_regroup;
    __shared__ StreamLens s_in;
    const uint32_t extent = stream_load(w.cnt, pp.depth, s_in);
    if (blockIdx.x * (uint32_t)kBlock >= extent) return;
    {
        const uint32_t i = global_thread();
        bool cont = false, conn = false; uint32_t slot = 0; V3 hp = mk(0, 0, 0); float fx = 0, fy = 0; C3 cval = mkc(0, 0, 0);
        bool have = i < extent && stream_valid(s_in, i);
        rtx_hit h; h.id = 0; h.t = FLT_MAX; h.alpha = h.beta = h.gamma = 0.0f;
        if (have) { slot = qin[i]; h = w.hit[slot]; }
        if constexpr (REGROUP) {
            uint32_t cls = kClassMiss;
            if (have && h.t < FLT_MAX) cls = sc.shade[h.id].shade_class;
            have = regroup_by_class(s_regroup, have, cls, slot, h);
        }
        if (have) {
            const rtx_ray r = w.ray[slot];
            if (h.t < FLT_MAX) {
                Shading sd; shading_data(sc, h, ld3(r.o), ld3(r.d), sd);
                if (!(sd.mat->is_light || isPureSpecular(*sd.mat))) {
                    C3 T = ldc(w.T + 3 * slot); const C3 Le = ldc(w.L + 3 * slot);
                    V3 wi = normalize(ld3(cam.origin) - sd.x);
                    C3 col = T * bsdf_evaluate(sc, sd, wi) * Le;
                    conn = connect_request(cam, sd.x, sd.sNormal, col, fx, fy, cval); hp = sd.x;
                    Rng rng; rng.key = w.key[slot]; rng.dim = w.dim[slot];
                    float rr = wmin(lum(T), 0.9f);
                    if (!(rr < rng.next())) {
                        T = T / rr;
                        C3 bs; float pdf; wi = bsdf_sample(sc, sd, rng, bs, pdf);
                        T = T * bs * fabsf(dot(wi, sd.sNormal)) / pdf;
                        V3 no = sd.x + (wi * kEps);
                        rtx_ray nr; nr.o[0] = no.x; nr.o[1] = no.y; nr.o[2] = no.z; nr.d[0] = wi.x; nr.d[1] = wi.y; nr.d[2] = wi.z;
                        w.ray[slot] = nr; st3(w.T + 3 * slot, T);
                        cont = pp.depth + 1 < pp.st.max_bounces;
                    }
                    w.dim[slot] = rng.dim;
                }
            }
        }
        push_connection(w, conn, kShadowCounterBase + pp.depth, hp, cam, fx, fy, cval);
        const uint32_t stream = wave_stream();
        const uint32_t qc = stream_push(cont, w.cnt + stream_word(pp.depth + 1, stream), stream);
        if (cont) qout[qc] = slot;
    }
}

// Camera connections of one stage: visible ones splat onto the film (Renderer.h:325-331)
struct SplatPolicy {
    const WfPtrs& w; int filter; const StreamLens& segs;
    __device__ __forceinline__ bool fetch(uint32_t k, float o[3], float d[3], float& maxT) const
    {
        if (!stream_valid(segs, k)) { o[0] = o[1] = o[2] = 0.0f; d[0] = d[1] = d[2] = 0.0f; maxT = 0.0f; return false; }    // a hole of the streamed queue
        segment_to_ray(w.sseg + 6 * (size_t)k, o, d, maxT); return true;
    }
    __device__ __forceinline__ void retire(uint32_t k, const HitRec& best, bool reached) const
    {
        if (reached && best.pos == 0xFFFFFFFFu)
            film_splat(w.film, (int)w.W, (int)w.H, filter, w.sxy[2 * (size_t)k], w.sxy[2 * (size_t)k + 1], ldc(w.sval + 3 * (size_t)k));
    }
};
__global__ __launch_bounds__(kBlock) void k_lt_splat(WfPtrs w, DeviceScene sc, uint32_t queue_index, uint32_t work_index, int filter, uint2* spill, uint32_t spill_stride)
{
    __shared__ unsigned long long lds[kBlock * kLdsStack];
    __shared__ StreamLens s_segs;
    LaneStack<kBlock, kLdsStack> stk; stk.init(lds, spill, spill_stride);
    const uint32_t n = stream_load(w.cnt, queue_index, s_segs);
    SplatPolicy pol{w, filter, s_segs};
    __shared__ TailShare<kBlock> ts;
    traverse_dyn<SplatPolicy, kModeAny, false, kRefill, kBlock, kLdsStack, kTriUnroll, kLeafLanes, kShare>(sc, stk, n, w.cnt + work_index, pol, nullptr, &ts);
}

// The camera connections of one stage and the closest hits of the next in one launch (as k_pt_trace_mixed does for the
// path tracer's shadow rays): items [0, n_closest) are the live light paths of `queue`, the others connection segments.
struct LtMixedPolicy {
    const DeviceScene& sc; const WfPtrs& w; const uint32_t* queue; uint32_t n_closest; int filter; const StreamLens& live; const StreamLens& segs;
    __device__ __forceinline__ bool fetch(uint32_t k, float o[3], float d[3], float& maxT, bool& any) const
    {
        o[0] = o[1] = o[2] = 0.0f; d[0] = d[1] = d[2] = 0.0f; maxT = 0.0f;
        if (k < n_closest) {
            any = false;
            if (!stream_valid(live, k)) return false;              // a hole of the streamed queue
            const rtx_ray r = w.ray[queue[k]];
            o[0] = r.o[0]; o[1] = r.o[1]; o[2] = r.o[2]; d[0] = r.d[0]; d[1] = r.d[1]; d[2] = r.d[2];
        } else {
            any = true;
            if (!stream_valid(segs, k - n_closest)) return false;
            segment_to_ray(w.sseg + 6 * (size_t)(k - n_closest), o, d, maxT);
        }
        return true;
    }
    __device__ __forceinline__ void retire(uint32_t k, const HitRec& best, bool reached) const
    {
        if (k < n_closest) w.hit[queue[k]] = make_hit(sc, best);
        else if (reached && best.pos == 0xFFFFFFFFu) {
            const size_t j = k - n_closest;
            film_splat(w.film, (int)w.W, (int)w.H, filter, w.sxy[2 * j], w.sxy[2 * j + 1], ldc(w.sval + 3 * j));
        }
    }
    __device__ __forceinline__ void note(uint32_t, uint32_t, uint32_t) const {}
};
template <bool COUNT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_lt_trace_mixed(WfPtrs w, DeviceScene sc, uint32_t live_queue, uint32_t segment_queue,
                                                           const uint32_t* __restrict__ queue, uint32_t work_index, int filter, uint2* spill, uint32_t spill_stride,
                                                           unsigned long long* counters)
{
    __shared__ unsigned long long lds[kBlock * kLdsStack];
    __shared__ StreamLens s_live, s_segs;
    LaneStack<kBlock, kLdsStack> stk; stk.init(lds, spill, spill_stride);
    const uint32_t n_closest = stream_load(w.cnt, live_queue, s_live), n_seg = stream_load(w.cnt, segment_queue, s_segs);
    LtMixedPolicy pol{sc, w, queue, n_closest, filter, s_live, s_segs};
    __shared__ TailShare<kBlock> ts;
    traverse_dyn<LtMixedPolicy, kModeMixed, COUNT, kRefill, kBlock, kLdsStack, kTriUnroll, kLeafLanes, COUNT ? 0 : kShare>(sc, stk, n_closest + n_seg, w.cnt + work_index, pol, counters, &ts);
}

// ---- instant radiosity: RTBase/Renderer.h:124-302 -------------------------------------------

__device__ __forceinline__ void vpl_store(const WfPtrs& w, uint32_t idx, V3 x, V3 n, C3 Le)
{
    float* v = w.vpl + 10 * (size_t)idx;
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = n.x; v[4] = n.y; v[5] = n.z; v[6] = Le.r; v[7] = Le.g; v[8] = Le.b; v[9] = 1.0f;
}

// Loop body of RayTracer::traceVPLs, RTBase/Renderer.h:176-207; VPL slot of path j at depth d
// is j*(maxBounces+1)+d, which keeps the reference's list order (path by path, along the path)
__global__ __launch_bounds__(kBlock) void k_vpl_emit(WfPtrs w, DeviceScene sc, PassParams pp, uint32_t n_paths)
{
    const uint32_t j = global_thread();
    bool live = false;
    if (j < n_paths && sc.n_lights > 0) {
        Rng rng; rng.key = rtx_rng_key(pp.st.seed, RTX_STREAM_VPL, j, pp.sample_index); rng.dim = 0;
        float pmf; const LightDev& light = sample_light(sc, rng, pmf);
        float pdfPos, pdfDir;
        V3 p = light_sample_position(sc, light, rng, pdfPos);
        V3 wi = light_sample_direction(light, rng, pdfDir);
        V3 nLight = light.kind == 0 ? ld3(light.gnormal) : -wi;
        float cosTheta = dot(wi, nLight);
        C3 Le = light_evaluate(sc, light, -wi) * cosTheta / (pmf * pdfPos);
        vpl_store(w, j * (pp.st.max_bounces + 1), p, nLight, Le);
        rtx_ray r; r.o[0] = p.x; r.o[1] = p.y; r.o[2] = p.z; r.d[0] = wi.x; r.d[1] = wi.y; r.d[2] = wi.z;
        w.ray[j] = r; st3(w.L + 3 * j, Le);
        w.T[3 * j] = 1.0f; w.T[3 * j + 1] = 1.0f; w.T[3 * j + 2] = 1.0f;
        w.key[j] = rng.key; w.dim[j] = rng.dim;
        live = pp.st.max_bounces > 0;
    }
    uint32_t q = queue_slot(live, &w.cnt[(0) * kQueueStride]);
    if (live) w.qa[q] = j;
}

// One depth of RayTracer::VPLTracePath, RTBase/Renderer.h:124-168
__global__ __launch_bounds__(kBlock) void k_vpl_shade(WfPtrs w, DeviceScene sc, PassParams pp, const uint32_t* __restrict__ qin, uint32_t* qout)
{
    const uint32_t n = w.cnt[(pp.depth) * kQueueStride];
    const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
    for (uint32_t it = 0; it < rounds; it++) {
        const uint32_t i = it * gridDim.x * kBlock + global_thread();
        bool cont = false; uint32_t j = 0;
        if (i < n) {
            j = qin[i];
            const rtx_hit h = w.hit[j]; const rtx_ray r = w.ray[j];
            if (h.t < FLT_MAX) {
                Shading sd; shading_data(sc, h, ld3(r.o), ld3(r.d), sd);
                if (!(sd.mat->is_light || isPureSpecular(*sd.mat))) {
                    C3 T = ldc(w.T + 3 * j); const C3 Le = ldc(w.L + 3 * j);
                    Rng rng; rng.key = w.key[j]; rng.dim = w.dim[j];
                    C3 bs; float pdf; V3 wi = bsdf_sample(sc, sd, rng, bs, pdf);
                    T = T * bs * fabsf(dot(wi, sd.sNormal)) / pdf;
                    vpl_store(w, j * (pp.st.max_bounces + 1) + pp.depth + 1, sd.x, sd.sNormal, T * Le);
                    float rr = wmin(lum(T), 0.9f);
                    if (!(rr < rng.next())) {
                        T = T / rr;
                        V3 no = sd.x + (wi * kEps);
                        rtx_ray nr; nr.o[0] = no.x; nr.o[1] = no.y; nr.o[2] = no.z; nr.d[0] = wi.x; nr.d[1] = wi.y; nr.d[2] = wi.z;
                        w.ray[j] = nr; st3(w.T + 3 * j, T);
                        cont = pp.depth + 1 < pp.st.max_bounces;
                    }
                    w.dim[j] = rng.dim;
                }
            }
        }
        uint32_t qc = queue_slot(cont, &w.cnt[(pp.depth + 1) * kQueueStride]);
        if (cont) qout[qc] = j;
    }
}

// The VPLs the walk left, in list order (vpls in RayTracer::radiosityLightPass, RTBase/Renderer.h:262-278: path by
// path, along each path), packed into w.vplc; their number goes to *w.vpl_total. One workgroup.
__global__ __launch_bounds__(1024) void k_vpl_compact(WfPtrs w)
{
    __shared__ uint32_t s_wave[1024 / 64], s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i0 = 0; i0 < w.vpl_slots; i0 += 1024u) {
        const uint32_t i = i0 + threadIdx.x;
        const bool valid = i < w.vpl_slots && w.vpl[10 * (size_t)i + 9] != 0.0f;
        const unsigned long long mask = __ballot(valid);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        if (valid) {
            uint32_t at = s_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            for (uint32_t j = 0; j < wave; j++) at += s_wave[j];
            for (int q = 0; q < 10; q++) w.vplc[10 * (size_t)at + q] = w.vpl[10 * (size_t)i + q];
        }
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = 0; for (int j = 0; j < 1024 / 64; j++) t += s_wave[j]; s_base += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *w.vpl_total = s_base;
}

// RayTracer::radiosityLightPass + radiosityComputeDirect, RTBase/Renderer.h:210-278: every pixel gathers from every
// VPL, in list order. Three kernels instead of one loop per pixel:
//   k_ir_prepare  shading data of every primary hit; pixels that gather (a hit on a surface that is neither a light
//                 nor purely specular) get a place g in the gather queue and a record of what BSDF::evaluate needs
//   k_ir_trace    the pixels x VPLs visibility segments as items of the dynamic traversal; item k of a launch is
//                 (VPL v0 + k / ng, pixel g = k % ng), so the rays in flight together leave neighbouring pixels for
//                 the same VPL. An item whose geometry term is not positive is done at fetch, without a walk. Result:
//                 one byte per item, 1 = the VPL contributes to the pixel
//   k_ir_shade    per pixel, the contributions of the VPLs marked 1, summed in list order like the reference's loop
// A launch covers the VPLs [v0, v0 + vb); the host runs as many launches as the upper bound of the VPL count needs.
// The gather record of place g: x and shading normal in sseg[g], wo in T[g], albedo in sval[g], material in spath[g].
__global__ __launch_bounds__(kBlock) void k_ir_prepare(WfPtrs w, DeviceScene sc, PassParams pp)
{
    const uint32_t n = w.cnt[(0) * kQueueStride];
    const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
    for (uint32_t it = 0; it < rounds; it++) {
        const uint32_t i = it * gridDim.x * kBlock + global_thread();
        bool gather = false; uint32_t slot = 0; Shading sd; sd.mat = nullptr;
        if (i < n) {
            slot = w.qa[i];
            const rtx_hit h = w.hit[slot]; const rtx_ray r = w.ray[slot];
            C3 L = mkc(0, 0, 0);
            if (h.t < FLT_MAX) {
                shading_data(sc, h, ld3(r.o), ld3(r.d), sd);
                if (isPureSpecular(*sd.mat)) L = mkc(0, 0, 0);
                else if (sd.mat->is_light) L = pp.st.can_hit_light ? ldc(sd.mat->emission) : mkc(0, 0, 0);
                else gather = true;
            } else L = background_evaluate(sc, ld3(r.d));
            if (!gather) st3(w.L + 3 * slot, L);
        }
        const uint32_t g = queue_slot(gather, &w.cnt[(1) * kQueueStride]);
        if (gather) {
            w.qb[g] = slot;
            st3(w.sseg + 6 * (size_t)g, sd.x); st3(w.sseg + 6 * (size_t)g + 3, sd.sNormal);
            st3(w.T + 3 * (size_t)g, sd.wo); st3(w.sval + 3 * (size_t)g, sd.alb);
            w.spath[g] = (uint32_t)(sd.mat - sc.materials);
        }
    }
}

// Box of the points and box of the shading normals of the places [256 b, 256 b + 256) of the gather queue (neighbouring
// pixels mostly: a piece of one surface), for ir_block_culled. One workgroup per block; the launch covers every block
// the film could have.
__global__ __launch_bounds__(kBlock) void k_ir_bounds(WfPtrs w, int everything)
{
    static_assert(kBlock == 256, "one thread per place of a block");
    const uint32_t ng = w.cnt[(1) * kQueueStride];
    if (blockIdx.x * 256u >= ng) return;
    const uint32_t g = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float lo[6], hi[6];
#pragma unroll
    for (int q = 0; q < 6; q++) { lo[q] = FLT_MAX; hi[q] = -FLT_MAX; }
    if (g < ng) {
#pragma unroll
        for (int q = 0; q < 6; q++) lo[q] = hi[q] = w.sseg[6 * (size_t)g + q];
    }
#pragma unroll
    for (int q = 0; q < 6; q++)
        for (int o = 32; o > 0; o >>= 1) { lo[q] = fminf(lo[q], __shfl_xor(lo[q], o, 64)); hi[q] = fmaxf(hi[q], __shfl_xor(hi[q], o, 64)); }
    __shared__ float s_lo[kBlock / 64][6], s_hi[kBlock / 64][6];
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 6; q++) { s_lo[wave][q] = lo[q]; s_hi[wave][q] = hi[q]; }
    }
    __syncthreads();
    if (threadIdx.x < 6u) {
        const uint32_t q = threadIdx.x;
        float l = s_lo[0][q], h = s_hi[0][q];
        for (uint32_t wv = 1; wv < kBlock / 64; wv++) { l = fminf(l, s_lo[wv][q]); h = fmaxf(h, s_hi[wv][q]); }
        if (everything) { l = -FLT_MAX; h = FLT_MAX; }      // (tuning switch: boxes that cull nothing)
        // layout: point low, point high, normal low, normal high
        float* out = w.ir_bounds + 12 * (size_t)blockIdx.x + (q < 3u ? q : q + 3u);
        out[0] = l; out[3] = h;
    }
}

// True when no place of a block can have a positive geometry term with the VPL at vx with normal vn: either the VPL
// lies behind every point's surface or every point behind the VPL's. d = vx - x of every pair lies between vx - xhi
// and vx - xlo component by component, as computed (rounding is monotone); interval bounds of the two dot products
// follow. The margins are those of the pair test in k_ir_trace_packet with room for the roundings of the bounds
// themselves: a pair of a block this returns true for fails that test too, so skipping the block changes no result.
__device__ __forceinline__ bool ir_block_culled(const float* bb, V3 vx, V3 vn)
{
    const float vxc[3] = {vx.x, vx.y, vx.z}, vnc[3] = {vn.x, vn.y, vn.z};
    float a_max = 0.0f, a_mag = 0.0f, b_min = 0.0f, b_mag = 0.0f;
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const float dlo = vxc[q] - bb[3 + q], dhi = vxc[q] - bb[q], nlo = bb[6 + q], nhi = bb[9 + q];
        const float dm = fmaxf(fabsf(dlo), fabsf(dhi));
        a_max += fmaxf(fmaxf(dlo * nlo, dlo * nhi), fmaxf(dhi * nlo, dhi * nhi));
        a_mag += dm * fmaxf(fabsf(nlo), fabsf(nhi));
        b_min += fminf(dlo * vnc[q], dhi * vnc[q]);
        b_mag += dm * fabsf(vnc[q]);
    }
    return a_max < -1e-5f * a_mag || b_min > 1e-5f * b_mag;
}

// geometry term between a gather point and a VPL, Renderer.h:229-236; wi comes back normalised
__device__ __forceinline__ float ir_geometry(V3 x, V3 ns, V3 vx, V3 vn, V3& wi)
{
    wi = vx - x; const float lsq = lengthSq(wi); wi = normalize(wi);
    return (wmax(dot(wi, ns), 0.0f) * wmax(-dot(wi, vn), 0.0f)) / lsq;
}

struct IrGatherPolicy {
    const WfPtrs& w; uint32_t ng, ngr, v0; uint8_t* vis; unsigned long long* counters;
    __device__ __forceinline__ bool fetch(uint32_t k, float o[3], float d[3], float& maxT) const
    {
        const uint32_t vl = k / ngr, g = k - vl * ngr;
        if (g >= ng) { o[0] = o[1] = o[2] = 0.0f; d[0] = d[1] = d[2] = 0.0f; maxT = 0.0f; return false; }    // (padding of the last block)
        const float* v = w.vplc + 10 * (size_t)(v0 + vl); const float* r = w.sseg + 6 * (size_t)g;
        const V3 x = ld3(r), vx = ld3(v);
        V3 wi;
        if (!(ir_geometry(x, ld3(r + 3), vx, ld3(v + 3), wi) > 0)) {
            vis[k] = 0; o[0] = o[1] = o[2] = 0.0f; d[0] = d[1] = d[2] = 0.0f; maxT = 0.0f;
            return false;
        }
        const float s[6] = {x.x, x.y, x.z, vx.x, vx.y, vx.z};
        segment_to_ray(s, o, d, maxT);
        return true;
    }
    __device__ __forceinline__ void retire(uint32_t k, const HitRec& best, bool reached) const { vis[k] = (reached && best.pos == 0xFFFFFFFFu) ? 1 : 0; }
    // instrumented runs: one more segment walked
    __device__ __forceinline__ void note(uint32_t, uint32_t, uint32_t) const { atomicAdd(&counters[6], 1ull); }
};
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_ir_trace(WfPtrs w, DeviceScene sc, uint32_t v0, uint32_t vb, uint8_t* vis, uint32_t work_index, uint2* spill, uint32_t spill_stride,
                                                     unsigned long long* counters)
{
    __shared__ unsigned long long lds[kBlock * kLdsStack];
    LaneStack<kBlock, kLdsStack> stk; stk.init(lds, spill, spill_stride);
    const uint32_t ng = w.cnt[(1) * kQueueStride], total = *w.vpl_total;
    const uint32_t nv = total > v0 ? min(vb, total - v0) : 0u;
    const uint32_t ngr = (ng + 255u) & ~255u;
    IrGatherPolicy pol{w, ng, ngr, v0, vis, counters};
    __shared__ TailShare<kBlock> ts;
    traverse_dyn<IrGatherPolicy, kModeAny, COUNT, kRefill, kBlock, kLdsStack, kTriUnroll, kLeafLanes, COUNT ? 0 : kShare>(sc, stk, ngr * nv, w.cnt + work_index, pol, counters, &ts);
}

// k_ir_trace as one traversal per wave (traverse_packet.h): consecutive items are neighbouring pixels gathering from the
// same VPL, so their segments pass the same nodes. A wave draws super-chunks of 256 (pixel, VPL) pairs from 64
// interleaved counters (as traverse_dyn does, so that the atomics do not queue on one address and the whole grid
// sweeps the pairs on one front), evaluates the geometry term of 64 pairs at a time, marks the culled ones done and
// collects the others in a small ring in LDS; whenever the ring holds a full wave of segments (or the pairs have
// run out) it walks them as one packet. Same result bytes as k_ir_trace.
__global__ __launch_bounds__(kBlock) void k_ir_trace_packet(WfPtrs w, DeviceScene sc, uint32_t v0, uint32_t vb, uint8_t* vis, uint32_t work_index, uint2* spill, uint32_t spill_stride
                                                            RTX_PACKET_STATS(, unsigned long long* stats))
{
    RTX_PACKET_STATS(uint32_t ws[8] = {0, 0, 0, 0, 0, 0, 0, 0};)    // (tuning build: [0..3] and [7] packet_occluded's, [4] packets, [5] their segments, [6] blocks culled whole)
    __shared__ PacketStacks<kBlock> ps;
    __shared__ uint32_t s_ring[kBlock / 64][128];
    LaneStack<kBlock, 0> exact_stack; exact_stack.init(nullptr, spill, spill_stride);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t ng = w.cnt[(1) * kQueueStride], total = *w.vpl_total;
    const uint32_t nv = total > v0 ? min(vb, total - v0) : 0u;
    const uint32_t ngr = (ng + 255u) & ~255u;       // a super-chunk is one block of 256 places for one VPL
    const uint32_t n = ngr * nv, n_super = n / 256u;
    uint32_t* work = w.cnt + work_index;
    uint32_t seg = blockIdx.x % (uint32_t)kWorkSegments;
    unsigned long long dry = 0ull;                  // counters this wave has found empty
    bool exhausted = n == 0u;
    uint32_t pair_next = 0, pair_end = 0, ring_head = 0, ring_count = 0, vl0 = 0;
    ps.hit[wv][lane] = 0u;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (;;) {
        while (ring_count < 64u && !exhausted) {
            if (pair_next == pair_end) {
                uint32_t j = 0;
                if (lane == 0) j = atomicAdd(work + seg * (uint32_t)kWorkStride, 1u);
                j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
                const uint32_t g = work_chunk(j, seg);
                if (j >= (1u << 24) || g >= n_super) {
                    // all counters advance on one front: when this one is empty the others have at most a few chunks left
                    dry |= 1ull << seg;
                    if (dry == ~0ull) { exhausted = true; break; }
                    do { seg = (seg + 1u) & (uint32_t)(kWorkSegments - 1); } while ((dry >> seg) & 1ull);
                    continue;
                }
                pair_next = g * 256u; pair_end = pair_next + 256u;
                vl0 = pair_next / ngr;
                // the whole block at once where its bounds allow it (nothing is written for its pairs: k_ir_shade decides the same)
                const float* v = w.vplc + 10 * (size_t)(v0 + vl0);
                if (ir_block_culled(w.ir_bounds + 12 * (size_t)((pair_next - vl0 * ngr) >> 8), ld3(v), ld3(v + 3))) { pair_next = pair_end; RTX_PACKET_STATS(ws[6]++;) continue; }
            }
            const uint32_t k = pair_next + lane;
            bool keep = false;
            if (k < pair_end && k - vl0 * ngr < ng) {
                const uint32_t vl = vl0, g = k - vl0 * ngr;
                const float* v = w.vplc + 10 * (size_t)(v0 + vl); const float* r = w.sseg + 6 * (size_t)g;
                // A cheap test first: the geometry term (ir_geometry) is zero when the VPL lies behind the pixel's surface or
                // the pixel behind the VPL's. The exact decision takes the NORMALISED direction (a square root and two
                // divisions); its two cosines differ from the plain dot products below by a positive factor and by rounding
                // errors of at most ~1e-6 of the sum of the absolute products, so a pair whose plain dot product is more
                // negative than 4e-6 of that sum is culled for certain. Everything else goes on and is decided exactly when
                // its segment is set up.
                const V3 d = ld3(v) - ld3(r), ns = ld3(r + 3), vn = ld3(v + 3);
                const float ax = d.x * ns.x, ay = d.y * ns.y, az = d.z * ns.z, bx = d.x * vn.x, by = d.y * vn.y, bz = d.z * vn.z;
                const float a = (ax + ay) + az, b = (bx + by) + bz;
                const float am = (fabsf(ax) + fabsf(ay)) + fabsf(az), bm = (fabsf(bx) + fabsf(by)) + fabsf(bz);
                keep = !(a < -4e-6f * am) && !(b > 4e-6f * bm);
                if (!keep) vis[k] = 0;
            }
            const unsigned long long km = __ballot(keep);
            if (keep) s_ring[wv][(ring_head + ring_count + (uint32_t)__popcll(km & below)) & 127u] = k;
            ring_count += (uint32_t)__popcll(km);
            pair_next = min(pair_next + 64u, pair_end);
        }
        if (ring_count == 0u) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // the ring entries other lanes wrote
        const uint32_t take = min(64u, ring_count);
        bool has = lane < take;
        const uint32_t k = has ? s_ring[wv][(ring_head + lane) & 127u] : 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        ring_head = (ring_head + take) & 127u; ring_count -= take;
        float o[3] = {0, 0, 0}, d[3] = {1, 1, 1}, maxT = 0.0f;
        if (has) {
            const uint32_t vl = k / ngr, g = k - vl * ngr;
            const float* v = w.vplc + 10 * (size_t)(v0 + vl); const float* r = w.sseg + 6 * (size_t)g;
            const V3 x = ld3(r), vx = ld3(v);
            V3 wi;
            if (ir_geometry(x, ld3(r + 3), vx, ld3(v + 3), wi) > 0) {
                const float s[6] = {x.x, x.y, x.z, vx.x, vx.y, vx.z};
                segment_to_ray(s, o, d, maxT);
            } else { vis[k] = 0; has = false; }
        }
        RTX_PACKET_STATS(ws[4]++; ws[5] += take;)
        const bool blocked = packet_occluded<kBlock, false>(sc, ps, exact_stack, has, o[0], o[1], o[2], d[0], d[1], d[2], maxT, nullptr, nullptr RTX_PACKET_STATS(, stats ? ws : nullptr));
        if (has) vis[k] = blocked ? 0 : 1;
    }
    RTX_PACKET_STATS(if (stats && lane == 0) for (int q = 0; q < 8; q++) atomicAdd(&stats[8 + q], (unsigned long long)ws[q]);)
}

// first: the sum starts at zero (otherwise it carries on from the launch before); last: divide by vpls.size(), Renderer.h:244
__global__ __launch_bounds__(kBlock) void k_ir_shade(WfPtrs w, DeviceScene sc, uint32_t v0, uint32_t vb, const uint8_t* __restrict__ vis, int first, int last)
{
    const uint32_t ng = w.cnt[(1) * kQueueStride], total = *w.vpl_total;
    const uint32_t nv = total > v0 ? min(vb, total - v0) : 0u;
    const uint32_t ngr = (ng + 255u) & ~255u;
    const uint32_t lane = threadIdx.x & 63u;
    // a wave takes 64 consecutive places: they lie in one block of 256, and what k_ir_trace_packet skipped for the block
    // as a whole (it wrote no result bytes there) is skipped here by the same test on the same numbers
    for (uint32_t base = blockIdx.x * kBlock + (threadIdx.x & ~63u); base < ng; base += gridDim.x * kBlock) {
        const uint32_t g = base + lane;
        const bool act = g < ng;
        const float* bb = w.ir_bounds + 12 * (size_t)(base >> 8);
        uint32_t slot = 0; Shading sd; sd.mat = &sc.materials[0];
        sd.x = sd.sNormal = sd.wo = mk(0, 0, 0); sd.alb = mkc(0, 0, 0);
        if (act) {
            slot = w.qb[g];
            sd.x = ld3(w.sseg + 6 * (size_t)g); sd.sNormal = ld3(w.sseg + 6 * (size_t)g + 3); sd.wo = ld3(w.T + 3 * (size_t)g);
            sd.alb = ldc(w.sval + 3 * (size_t)g); sd.mat = &sc.materials[w.spath[g]];
        }
        sd.gNormal = sd.sNormal; sd.tu = sd.tv = 0.0f; sd.t = 0.0f;
        sd.frame.fromVector(sd.sNormal);
        C3 acc = (first || !act) ? mkc(0, 0, 0) : ldc(w.L + 3 * slot);
        for (uint32_t v64 = 0; v64 < nv; v64 += 64u) {
            // lane l decides for VPL v64 + l; the wave then walks the VPLs left, in list order
            bool look = false;
            if (v64 + lane < nv) { const float* v = w.vplc + 10 * (size_t)(v0 + v64 + lane); look = !ir_block_culled(bb, ld3(v), ld3(v + 3)); }
            unsigned long long todo = __ballot(look);
            while (todo != 0ull) {
                // eight result bytes at a time, so that their loads are in flight together (one dependent load per VPL otherwise)
                uint32_t idx[8]; uint8_t flag[8];
#pragma unroll
                for (uint32_t q = 0; q < 8u; q++) {
                    idx[q] = 0xFFFFFFFFu;
                    if (todo != 0ull) { idx[q] = v64 + (uint32_t)__builtin_ctzll(todo); todo &= todo - 1ull; }
                }
#pragma unroll
                for (uint32_t q = 0; q < 8u; q++) flag[q] = (act && idx[q] != 0xFFFFFFFFu) ? vis[(size_t)idx[q] * ngr + g] : (uint8_t)0;
#pragma unroll
                for (uint32_t q = 0; q < 8u; q++) {
                    if (!flag[q]) continue;
                    const float* v = w.vplc + 10 * (size_t)(v0 + idx[q]);
                    V3 wi; const float gt = ir_geometry(sd.x, sd.sNormal, ld3(v), ld3(v + 3), wi);
                    acc = acc + bsdf_evaluate(sc, sd, wi) * ldc(v + 6) * gt;
                }
            }
        }
        if (act) {
            if (last) acc = acc / (float)total;
            st3(w.L + 3 * slot, acc);
        }
    }
}

// ---- adaptive sampling: RayTracer::calculateTileSamples, RTBase/Renderer.h:643-671 ----------
// One lane per tile, summing in the reference's order (Film::getLums walks x outside, y
// inside, RTBase/Imaging.h:376-385) so the truncated sample targets agree exactly.
__global__ __launch_bounds__(kBlock) void k_tile_samples(WfPtrs w, int spp, uint32_t total_spp)
{
    const uint32_t i = global_thread();
    if (i >= w.n_tiles) return;
    const uint32_t sx = (i % w.tiles_x) * 16u, sy = (i / w.tiles_x) * 16u;
    const uint32_t ex = min(sx + 16u, w.W), ey = min(sy + 16u, w.H);
    const uint32_t count = (ex - sx) * (ey - sy);
    float total = 0.0f;
    for (uint32_t x = sx; x < ex; x++) for (uint32_t y = sy; y < ey; y++) total += lum(ldc(w.film + 3 * ((size_t)y * w.W + x)));
    float mean = count == 0 ? 0.0f : total / (float)count;
    float variance = 0.0f;
    for (uint32_t x = sx; x < ex; x++) for (uint32_t y = sy; y < ey; y++) { float l = lum(ldc(w.film + 3 * ((size_t)y * w.W + x))); variance += (l - mean) * (l - mean); }
    variance = count == 0 ? 0.0f : variance / (float)count;
    float q = variance / (variance + mean * mean + kEps);
    float weight = (q < kEps) ? kEps : (q > 1.0f) ? 1.0f : q;
    w.tile_samples[i] = (uint32_t)((float)spp + (float)(total_spp - (uint32_t)spp) * weight);
}

// Colour plane of RayTracer::createAOV, RTBase/Renderer.h:798-805: film / spp, with the tile's
// own sample count once adaptive sampling has taken over
__global__ __launch_bounds__(kBlock) void k_aov_color(WfPtrs w, float* __restrict__ color, int film_spp, int adaptive, int init_spp)
{
    const uint32_t i = global_thread();
    if (i >= w.W * w.H) return;
    const uint32_t x = i % w.W, y = i / w.W;
    const uint32_t tile = (y / 16u) * w.tiles_x + x / 16u;
    const int spp = (adaptive && init_spp < film_spp) ? (int)min(w.tile_samples[tile], (uint32_t)film_spp) : film_spp;
    const C3 c = ldc(w.film + 3 * (size_t)i) / (float)spp;
    st3(color + 3 * (size_t)i, c);
}

// ---- host side --------------------------------------------------------------------------------

inline void wavefront_release(rtx_ctx* c) { for (auto& wf : c->wf) { delete wf; wf = nullptr; } }

// Path state of pipeline lane `lane` for the current camera and the tiles `st` makes this context render: every tile of
// the film, or with the tiles split over GPUs (shard_mode 1) only the context's own, so that the per-pass sweeps over the
// slots (ray generation, film accumulation) and the memory shrink with the number of ranks. Buffers only ever grow.
inline int wavefront_alloc(rtx_ctx* c, int lane, const rtx_settings& st)
{
    if (!c->wf[lane]) c->wf[lane] = new WavefrontBuffers;
    WavefrontBuffers& b = *c->wf[lane];
    const uint32_t n_tiles = c->tiles_x * c->tiles_y;
    const bool split = st.shard_mode == 1 && st.shard_world > 1;
    const uint32_t mul = split ? (uint32_t)st.shard_world : 1u, add = split ? (uint32_t)st.shard_rank : 0u;
    const uint32_t n_local = n_tiles > add ? (n_tiles - add + mul - 1u) / mul : 0u;
    // (the light paths of instant radiosity use slots of their own: numThreads x vplRaysPerTile of them, Renderer.h:280-302)
    const size_t vpl_paths = st.algorithm == RTX_AL_INSTANT_RADIOSITY ? (size_t)st.vpl_rays_per_tile * st.num_threads : 0;
    const size_t P = std::max<size_t>(std::max<size_t>((size_t)n_local * 256, (vpl_paths + 255) / 256 * 256), 256);
#define RTX_ENS(buf, bytes) do { if ((buf).ensure(bytes) != hipSuccess) return RTX_ERR_NOMEM; } while (0)
    RTX_ENS(b.ray, P * sizeof(rtx_ray)); RTX_ENS(b.hit, P * sizeof(rtx_hit)); RTX_ENS(b.T, P * 12); RTX_ENS(b.L, P * 12);
    RTX_ENS(b.xy, P * 8); RTX_ENS(b.key, P * 4); RTX_ENS(b.dim, P * 4); RTX_ENS(b.mis, P * 4); RTX_ENS(b.flags, P * 4);
    const size_t Q = P + kQueueSlack;      // what is indexed by queue position
    RTX_ENS(b.qa, Q * 4); RTX_ENS(b.qb, Q * 4); RTX_ENS(b.cnt, kCounterWords * 4);
    RTX_ENS(b.sseg, Q * 24); RTX_ENS(b.sval, Q * 12); RTX_ENS(b.sxy, Q * 8); RTX_ENS(b.spath, Q * 4); RTX_ENS(b.vpl_total, 16); RTX_ENS(b.tile_off, P / 256 * 4);
    if (st.algorithm == RTX_AL_INSTANT_RADIOSITY) RTX_ENS(b.ir_bounds, P / 256 * 48);
#undef RTX_ENS
    WfPtrs& p = b.p;
    p.ray = b.ray.as<rtx_ray>(); p.hit = b.hit.as<rtx_hit>(); p.T = b.T.as<float>(); p.L = b.L.as<float>(); p.xy = b.xy.as<float>();
    p.key = b.key.as<uint32_t>(); p.dim = b.dim.as<uint32_t>(); p.mis = b.mis.as<float>(); p.flags = b.flags.as<uint32_t>();
    p.qa = b.qa.as<uint32_t>(); p.qb = b.qb.as<uint32_t>(); p.cnt = b.cnt.as<uint32_t>();
    p.sseg = b.sseg.as<float>(); p.sval = b.sval.as<float>(); p.sxy = b.sxy.as<float>(); p.spath = b.spath.as<uint32_t>();
    p.vpl_total = b.vpl_total.as<uint32_t>(); p.tile_off = b.tile_off.as<uint32_t>(); p.ir_bounds = b.ir_bounds.as<float>();
    p.film = c->b_film.as<float>(); p.tile_samples = c->b_tile_samples.as<uint32_t>();
    p.P = (uint32_t)n_local * 256u; p.W = (uint32_t)c->cam.width; p.H = (uint32_t)c->cam.height; p.tiles_x = c->tiles_x; p.n_tiles = n_tiles;
    p.tile_mul = mul; p.tile_add = add; p.n_local_tiles = n_local;
    p.depth_stats = c->depth_stats;
    return RTX_OK;
}

// New camera / film size: the path state of the old size is dropped; rendering calls size it again for what they need
inline int wavefront_resize(rtx_ctx* c)
{
    for (int i = 0; i < kPipeline; i++) { delete c->wf[i]; c->wf[i] = nullptr; }
    // tileSamples.resize(totalTiles, 1), RTBase/Renderer.h:112
    std::vector<uint32_t> ones((size_t)c->tiles_x * c->tiles_y, 1u);
    if (hipMemcpy(c->b_tile_samples.p, ones.data(), ones.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return RTX_ERR_HIP;
    return RTX_OK;
}

inline int wavefront_tile_samples(rtx_ctx* c, hipStream_t stream)
{
    WfPtrs w{};         // the kernel reads the film geometry only
    w.film = c->b_film.as<float>(); w.tile_samples = c->b_tile_samples.as<uint32_t>();
    w.W = (uint32_t)c->cam.width; w.H = (uint32_t)c->cam.height; w.tiles_x = c->tiles_x; w.n_tiles = c->tiles_x * c->tiles_y;
    hipLaunchKernelGGL(k_tile_samples, dim3((w.n_tiles + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, w, c->spp, c->settings.total_spp);
    return hipGetLastError() == hipSuccess ? RTX_OK : RTX_ERR_HIP;
}

// RayTracer::createAOV, RTBase/Renderer.h:786-820, into device planes of W*H*3 floats (any may be
// null): colour = film / spp; albedo and normals = the draw-mode integrators through the pixel
// centres (:699-720 with DM_ALBEDO / DM_NORMALS), rendered into the plane instead of the film.
inline int wavefront_aov(rtx_ctx* c, float* color, float* albedo, float* normal)
{
    {   // every tile of the film, whatever share of it this context renders
        rtx_settings whole = c->settings; whole.algorithm = RTX_AL_PATH_TRACE; whole.shard_mode = 0; whole.shard_world = 1; whole.shard_rank = 0;
        int rc = wavefront_alloc(c, 0, whole); if (rc) return rc;
    }
    WfPtrs w = c->wf[0]->p; hipStream_t s = c->stream;
    c->wf[0]->cnt_clean = false;
    const dim3 blk(kBlock), gridP((w.P + kBlock - 1) / kBlock), gridQ(std::min<uint32_t>((w.P + kBlock - 1) / kBlock, kMaxGridBlocks)),
               gridT(std::min<uint32_t>((w.P + kBlock - 1) / kBlock, (uint32_t)c->resident_blocks));
    const size_t plane = (size_t)w.W * w.H * 3 * sizeof(float);
    if (color) hipLaunchKernelGGL(k_aov_color, dim3((w.W * w.H + kBlock - 1) / kBlock), blk, 0, s, w, color, c->spp, (int)c->settings.adaptive_sampling, (int)c->settings.init_spp);
    for (int which = 0; which < 2; which++) {
        float* target = which == 0 ? albedo : normal;
        if (!target) continue;
        PassParams pp; pp.st = c->settings; pp.spp = 1; pp.sample_index = 1; pp.depth = 0;
        pp.st.draw_mode = which == 0 ? RTX_DM_ALBEDO : RTX_DM_NORMALS;
        pp.st.algorithm = RTX_AL_PATH_TRACE; pp.st.filter = RTX_FT_BOX; pp.st.init_spp = 0x7FFFFFFF; pp.st.shard_mode = 0; pp.st.shard_world = 1;   // every tile, whole film
        w.film = target;
        if (hipMemsetAsync(target, 0, plane, s) != hipSuccess || hipMemsetAsync(w.cnt, 0, kCounterWords * 4, s) != hipSuccess) return RTX_ERR_HIP;
        hipLaunchKernelGGL(k_tile_offsets, dim3(1), dim3(1024), 0, s, w, pp);
        hipLaunchKernelGGL(k_raygen, gridP, blk, 0, s, w, c->dcam, pp);
        hipLaunchKernelGGL(k_closest<false>, gridT, blk, 0, s, c->dscene, (const rtx_ray*)w.ray, (const uint32_t*)w.qa, 0u, (const uint32_t*)w.cnt, w.hit,
                           c->b_spill.as<uint2>(), (uint32_t)(kMaxGridBlocks * kBlock), w.cnt + kClosestWorkBase, (unsigned long long*)nullptr);
        hipLaunchKernelGGL(k_dm_shade, gridQ, blk, 0, s, w, c->dscene, pp);
        launch_film_accumulate(w, pp, false, (int)RTX_FT_BOX, gridP, blk, s);
    }
    return hipGetLastError() == hipSuccess ? RTX_OK : RTX_ERR_HIP;
}

// One pass on pipeline lane `lane`, enqueued on stream s. Film sums of consecutive passes are
// kept in pass order (float addition does not commute bit for bit): the pass waits for
// film_wait before it touches the film and records film_signal when it has.
inline int wavefront_render_pass(rtx_ctx* c, int lane, hipStream_t s, hipEvent_t film_wait, hipEvent_t film_signal)
{
    if (!c->wf[lane]) return RTX_ERR_STATE;
    WavefrontBuffers& b = *c->wf[lane]; WfPtrs& w = b.p; const rtx_settings& st = c->settings;
    const dim3 blk(kBlock), gridP((w.P + kBlock - 1) / kBlock), gridQ(std::min<uint32_t>((w.P + kBlock - 1) / kBlock, kMaxGridBlocks)),
               gridT(std::min<uint32_t>((w.P + kBlock - 1) / kBlock, (uint32_t)c->resident_blocks));   // traversal kernels fetch work dynamically
    const uint32_t sstride = kMaxGridBlocks * kBlock;
    uint2* spill = c->b_spill.as<uint2>() + (size_t)lane * c->spill_entries * sstride;

    // several passes in flight fill each other's idle lanes: spreading a small launch thin (helpers walking speculative
    // subtrees) then only takes throughput away from the other passes, so it is kept for launches that are nearly empty
    DeviceScene dscene = c->dscene;
    if (c->passes_in_flight > 1) dscene.spread_below = std::min<uint32_t>(dscene.spread_below, 2u);

    c->spp++;                                            // Film::incrementSPP, RTBase/Imaging.h:403
    PassParams pp; pp.st = st; pp.spp = c->spp; pp.depth = 0;
    pp.sample_index = (st.shard_mode == 2 && st.shard_world > 1) ? (uint32_t)(c->spp - 1) * (uint32_t)st.shard_world + (uint32_t)st.shard_rank + 1u : (uint32_t)c->spp;
    const bool counters_zero = b.cnt_clean;
    b.cnt_clean = false;
    if (!counters_zero && hipMemsetAsync(w.cnt, 0, kCounterWords * 4, s) != hipSuccess) return RTX_ERR_HIP;

    const bool cw = c->count_work;                       // instrumented kernels (rtx_count_render_work)
    unsigned long long* ctr = cw ? c->b_counters.as<unsigned long long>() : nullptr;
    auto closest = [&](const uint32_t* queue, uint32_t counter) {
        if (cw) hipLaunchKernelGGL(k_closest_counted, gridT, blk, 0, s, dscene, (const rtx_ray*)w.ray, queue, (const uint32_t*)(w.cnt + counter * kQueueStride), w.hit, spill, sstride,
                                   w.cnt + kClosestWorkBase + counter * kWorkSlotWords, ctr);
        else hipLaunchKernelGGL(k_closest<false>, gridT, blk, 0, s, dscene, (const rtx_ray*)w.ray, queue, 0u, (const uint32_t*)(w.cnt + counter * kQueueStride), w.hit, spill, sstride,
                           w.cnt + kClosestWorkBase + counter * kWorkSlotWords, (unsigned long long*)nullptr);
    };
    // camera rays (queue 0 of a pass, in pixel-block order): packets of 64 first, what does not fit a packet per lane.
    // The rest queue is w.qb (free until the first shade kernel of the pass writes it), its length queue word 95.
    auto closest_primary = [&]() {
#ifndef RTX_TUNING
        closest(w.qa, 0);
#else
        if (cw || !c->primary_packet || c->stack_need > (uint32_t)kPacketStack) { closest(w.qa, 0); return; }
        uint32_t* rest_count = w.cnt + 95 * kQueueStride;
        hipLaunchKernelGGL(k_closest_packet, dim3(std::min<uint32_t>((w.P / 64 + 3) / 4, (uint32_t)c->resident_packet_blocks)), blk, 0, s, dscene, (const rtx_ray*)w.ray, (const uint32_t*)w.qa, 0u,
                           (const uint32_t*)w.cnt, w.hit, spill, sstride, w.cnt + kShadowWorkBase + 94 * kWorkSlotWords, w.qb, rest_count);
        hipLaunchKernelGGL(k_closest<false>, gridT, blk, 0, s, dscene, (const rtx_ray*)w.ray, (const uint32_t*)w.qb, 0u, (const uint32_t*)rest_count, w.hit, spill, sstride,
                           w.cnt + kClosestWorkBase, (unsigned long long*)nullptr);
#endif
    };
    auto shadow = [&](hipStream_t on, uint32_t counter_index, uint32_t work_index, uint2* stacks) {
        if (cw) hipLaunchKernelGGL(k_shadow_accumulate<true>, gridT, blk, 0, on, w, dscene, counter_index, work_index, stacks, sstride, ctr);
        else hipLaunchKernelGGL(k_shadow_accumulate<false>, gridT, blk, 0, on, w, dscene, counter_index, work_index, stacks, sstride, (unsigned long long*)nullptr);
    };
    const bool algo = st.draw_mode == RTX_DM_ALGORITHM;

    if (!algo) {
        // RTBase/Renderer.h:699-720
        hipLaunchKernelGGL(k_tile_offsets, dim3(1), dim3(1024), 0, s, w, pp);
        hipLaunchKernelGGL(k_raygen, gridP, blk, 0, s, w, c->dcam, pp);
        closest_primary();
        hipLaunchKernelGGL(k_dm_shade, gridQ, blk, 0, s, w, dscene, pp);
        shadow(s, (uint32_t)kShadowCounterBase, (uint32_t)kShadowWorkBase, spill);
        if (film_wait && hipStreamWaitEvent(s, film_wait, 0) != hipSuccess) return RTX_ERR_HIP;
        launch_film_accumulate(w, pp, false, (int)st.filter, gridP, blk, s);
        c->stats.samples += (uint64_t)w.W * w.H;
    } else if (st.algorithm == RTX_AL_PATH_TRACE) {
        // no ray generation: the kernels of depth 0 derive the camera samples from the slot numbers (camera_sample)
        // closest(0), shade(0), then per depth ONE traversal launch for the shadow rays of depth d - 1 and the closest
        // hits of depth d (neither depends on the other, and a traversal launch lasts at least as long as its slowest
        // ray), shade(d); the launch of depth max_bounces + 1 has the shadow rays of the last depth only.
        // A shade launch has a thread for every position a queue can have (streamed queues, see stream_push).
        const dim3 gridS((w.P + 1023u) / 1024u * 4u);
        auto mixed = [&](uint32_t d, const uint32_t* queue) {
            if (cw) hipLaunchKernelGGL(k_pt_trace_mixed<true>, gridT, blk, 0, s, w, dscene, d, queue, (uint32_t)(kClosestWorkBase + d * kWorkSlotWords), spill, sstride, ctr);
            else hipLaunchKernelGGL(k_pt_trace_mixed<false>, gridT, blk, 0, s, w, dscene, d, queue, (uint32_t)(kClosestWorkBase + d * kWorkSlotWords), spill, sstride, (unsigned long long*)nullptr);
        };
        uint32_t* qin = w.qa; uint32_t* qout = w.qb;
        // from depth c->tail_from_depth on (few paths left) one launch walks the rest of every path, see k_pt_tail
        // (the first depth, not before c->tail_from_depth, whose queue fit one round of the tail kernel's waves in the last pass
        // the device has reported; the report lags by the passes in flight, and any depth gives the same film)
        // (a wave takes two blocks of 64 positions at a time when passes overlap, one when this pass has the device to itself:
        //  fuller lanes per instruction help the former, more waves the latter)
        const int tail_rounds = c->tail_rounds ? c->tail_rounds : (c->passes_in_flight > 1 ? 2 : 1);
        const int tail_blocks_fit = c->resident_tail_blocks[tail_rounds - 1];
        uint32_t tail_from = 0xFFFFFFFFu;
        if (!cw && c->tail_from_depth != 0u && c->depth_stats) {
            const volatile uint32_t* ds = c->depth_stats;
            const uint32_t fits = (uint32_t)tail_blocks_fit * (uint32_t)(kBlock / 64) * 64u * (uint32_t)tail_rounds;
            for (uint32_t d = std::max(1u, c->tail_from_depth); d <= st.max_bounces && d < 96u; d++) if (ds[d] <= fits) { tail_from = d; break; }
        }
        bool tail_done = false;
        for (uint32_t d = 0; d <= st.max_bounces; d++) {
            pp.depth = d;
            if (d >= tail_from) {
                mixed(d, qin);
                RTX_PACKET_STATS(unsigned long long* stats = c->tail_stats ? c->b_counters.as<unsigned long long>() : nullptr;)
                // (RTX_TAIL_GRID, tuning build: fewer workgroups than the queue was sized for, so that waves take several tasks)
                const int tail_blocks = c->tail_grid_limit > 0 ? std::min(tail_blocks_fit, c->tail_grid_limit) : tail_blocks_fit;
                if (tail_rounds == 2) hipLaunchKernelGGL(k_pt_tail<2>, dim3(tail_blocks), blk, 0, s, w, dscene, c->dcam, pp, (const uint32_t*)qin, (uint32_t)kShadowWorkBase, spill, sstride RTX_PACKET_STATS(, stats));
                else hipLaunchKernelGGL(k_pt_tail<1>, dim3(tail_blocks), blk, 0, s, w, dscene, c->dcam, pp, (const uint32_t*)qin, (uint32_t)kShadowWorkBase, spill, sstride RTX_PACKET_STATS(, stats));
                tail_done = true;
                break;
            }
            if (d > 0) mixed(d, qin);
            else if (cw) hipLaunchKernelGGL(k_pt_closest_primary<true>, gridT, blk, 0, s, w, dscene, c->dcam, pp, (uint32_t)kClosestWorkBase, spill, sstride, ctr);
            else hipLaunchKernelGGL(k_pt_closest_primary<false>, gridT, blk, 0, s, w, dscene, c->dcam, pp, (uint32_t)kClosestWorkBase, spill, sstride, (unsigned long long*)nullptr);
            const bool regroup = d >= c->regroup_from_depth;
            if (d == 0) { if (regroup) hipLaunchKernelGGL((k_pt_shade<true, true>), gridS, blk, 0, s, w, dscene, c->dcam, pp, (const uint32_t*)qin, qout);
                          else hipLaunchKernelGGL((k_pt_shade<true, false>), gridS, blk, 0, s, w, dscene, c->dcam, pp, (const uint32_t*)qin, qout); }
            else if (regroup) hipLaunchKernelGGL((k_pt_shade<false, true>), gridS, blk, 0, s, w, dscene, c->dcam, pp, (const uint32_t*)qin, qout);
            else hipLaunchKernelGGL((k_pt_shade<false, false>), gridS, blk, 0, s, w, dscene, c->dcam, pp, (const uint32_t*)qin, qout);
            std::swap(qin, qout);
        }
        if (!tail_done) mixed(st.max_bounces + 1u, qin);
        if (film_wait && hipStreamWaitEvent(s, film_wait, 0) != hipSuccess) return RTX_ERR_HIP;
        pp.depth = 0;
        // (the instrumented passes leave their counters for the caller to read)
        const bool clean_after = !cw && st.filter == RTX_FT_BOX;
        launch_film_accumulate(w, pp, true, (int)st.filter, gridP, blk, s, clean_after);
        b.cnt_clean = clean_after;
        c->stats.samples += (uint64_t)w.W * w.H;
    } else if (st.algorithm == RTX_AL_LIGHT_TRACE) {
        hipLaunchKernelGGL(k_lt_emit, gridP, blk, 0, s, w, dscene, c->dcam, pp);
        uint32_t* qin = w.qa; uint32_t* qout = w.qb;
        const dim3 gridS((w.P + 1023u) / 1024u * 4u);       // a thread for every position a streamed queue can have
        // the connections of the light vertices (queue 95) ride with the closest hits of depth 0, those of depth d with the
        // closest hits of depth d + 1; the last ones go alone
        for (uint32_t d = 0; d < st.max_bounces; d++) {
            pp.depth = d;
            const uint32_t seg_q = (uint32_t)kShadowCounterBase + (d == 0 ? 95u : d - 1u), work = (uint32_t)(kClosestWorkBase + d * kWorkSlotWords);
            if (cw) hipLaunchKernelGGL(k_lt_trace_mixed<true>, gridT, blk, 0, s, w, dscene, d, seg_q, (const uint32_t*)qin, work, (int)st.filter, spill, sstride, ctr);
            else hipLaunchKernelGGL(k_lt_trace_mixed<false>, gridT, blk, 0, s, w, dscene, d, seg_q, (const uint32_t*)qin, work, (int)st.filter, spill, sstride, (unsigned long long*)nullptr);
            if (c->lt_regroup) hipLaunchKernelGGL(k_lt_shade<true>, gridS, blk, 0, s, w, dscene, c->dcam, pp, (const uint32_t*)qin, qout);
            else hipLaunchKernelGGL(k_lt_shade<false>, gridS, blk, 0, s, w, dscene, c->dcam, pp, (const uint32_t*)qin, qout);
            std::swap(qin, qout);
        }
        const uint32_t last = st.max_bounces > 0 ? st.max_bounces - 1u : 95u;
        hipLaunchKernelGGL(k_lt_splat, gridT, blk, 0, s, w, dscene, (uint32_t)(kShadowCounterBase + last), (uint32_t)(kShadowWorkBase + last * kWorkSlotWords), (int)st.filter, spill, sstride);
        c->stats.samples += (uint64_t)w.W * w.H;
    } else {
        // radiosityVplPass, RTBase/Renderer.h:280-302: numThreads x vplRaysPerTile light paths
        const uint32_t n_paths = st.vpl_rays_per_tile * st.num_threads;
        const uint32_t slots = n_paths * (st.max_bounces + 1);
        if (b.vpl.ensure(std::max<size_t>((size_t)slots * 80, 16)) != hipSuccess) return RTX_ERR_NOMEM;
        w.vpl = b.vpl.as<float>(); w.vplc = w.vpl + 10 * (size_t)slots; w.vpl_slots = slots;
        if (slots && hipMemsetAsync(w.vpl, 0, (size_t)slots * 40, s) != hipSuccess) return RTX_ERR_HIP;
        if (n_paths) {
            hipLaunchKernelGGL(k_vpl_emit, dim3((n_paths + kBlock - 1) / kBlock), blk, 0, s, w, dscene, pp, n_paths);
            uint32_t* qin = w.qa; uint32_t* qout = w.qb;
            for (uint32_t d = 0; d < st.max_bounces; d++) {
                pp.depth = d;
                closest(qin, d);
                hipLaunchKernelGGL(k_vpl_shade, gridQ, blk, 0, s, w, dscene, pp, (const uint32_t*)qin, qout);
                std::swap(qin, qout);
            }
        }
        hipLaunchKernelGGL(k_vpl_compact, dim3(1), dim3(1024), 0, s, w);
        // per-pixel gather, RTBase/Renderer.h:730-745. The device knows how many pixels gather and how many VPLs are
        // valid; the host sizes the launches for the upper bounds (every pixel, every slot): vb VPLs per launch, so
        // that a launch has fewer than 2^32 items and its result bytes fit the buffer.
        uint32_t vb = slots, batches = slots ? 1u : 0u;
        if (slots) {
            const uint64_t cap = 1ull << 30;
            vb = (uint32_t)std::max<uint64_t>(std::max<uint64_t>(cap / w.P, 1), (slots + 63u) / 64u);
            vb = std::min(vb, slots); batches = (slots + vb - 1u) / vb;
            if ((uint64_t)vb * w.P > 0xF0000000ull) return RTX_ERR_INVALID;      // (a film and a VPL count far beyond any use)
            if (b.ir_vis.ensure((size_t)vb * w.P) != hipSuccess) return RTX_ERR_NOMEM;
        }
        pp.depth = 0;
        if (hipMemsetAsync(w.cnt, 0, kCounterWords * 4, s) != hipSuccess) return RTX_ERR_HIP;
        hipLaunchKernelGGL(k_tile_offsets, dim3(1), dim3(1024), 0, s, w, pp);
        hipLaunchKernelGGL(k_raygen, gridP, blk, 0, s, w, c->dcam, pp);
        closest_primary();
        hipLaunchKernelGGL(k_ir_prepare, gridQ, blk, 0, s, w, dscene, pp);
        hipLaunchKernelGGL(k_ir_bounds, dim3(w.P / 256u), blk, 0, s, w, c->ir_block_cull ? 0 : 1);
        for (uint32_t bi = 0; bi < std::max(batches, 1u); bi++) {
            const uint32_t v0 = bi * vb, work = (uint32_t)(kShadowWorkBase + bi * kWorkSlotWords);
            uint8_t* vis = b.ir_vis.as<uint8_t>();
            if (batches) {
                if (cw) hipLaunchKernelGGL(k_ir_trace<true>, gridT, blk, 0, s, w, dscene, v0, vb, vis, work, spill, sstride, ctr);
                else if (c->ir_packet && c->stack_need <= (uint32_t)kPacketStack) hipLaunchKernelGGL(k_ir_trace_packet, dim3(c->resident_packet_blocks), blk, 0, s, w, dscene, v0, vb, vis, work, spill, sstride RTX_PACKET_STATS(, c->tail_stats ? c->b_counters.as<unsigned long long>() : nullptr));
                else hipLaunchKernelGGL(k_ir_trace<false>, gridT, blk, 0, s, w, dscene, v0, vb, vis, work, spill, sstride, (unsigned long long*)nullptr);
            }
            hipLaunchKernelGGL(k_ir_shade, gridQ, blk, 0, s, w, dscene, v0, vb, (const uint8_t*)vis, bi == 0 ? 1 : 0, bi + 1u >= batches ? 1 : 0);
        }
        if (film_wait && hipStreamWaitEvent(s, film_wait, 0) != hipSuccess) return RTX_ERR_HIP;
        launch_film_accumulate(w, pp, false, (int)st.filter, gridP, blk, s);
        c->stats.samples += (uint64_t)w.W * w.H;
    }
    if (hipGetLastError() != hipSuccess) return RTX_ERR_HIP;
    // RTBase/Renderer.h:780-781
    if (c->spp == (int)st.init_spp && st.algorithm != RTX_AL_LIGHT_TRACE && st.shard_mode != 2) {
        int rc = wavefront_tile_samples(c, s);
        if (rc) return rc;
    }
    if (film_signal && hipEventRecord(film_signal, s) != hipSuccess) return RTX_ERR_HIP;
    return RTX_OK;
}

} // namespace rtx
