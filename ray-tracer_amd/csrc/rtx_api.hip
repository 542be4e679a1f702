// rtx_api.hip — implementation of the C ABI declared in include/rtx.h.
#include <hip/hip_runtime.h>

#include <functional>
#include <thread>
#include <condition_variable>

#include <algorithm>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "context.h"
#include "trace_kernels.h"
#include "bvh_device.h"
#include "wavefront.h"
#include "debug_tables.h"

using namespace rtx;

namespace rtx {
thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}
}

namespace {

// Switches that exist for measurements only (schedules under test, grid sizes, counters on stderr) are read in the
// tuning build of the library (-DRTX_TUNING, librtx_tuning.so, used by tools/ and the A/B legs of bench.py); the
// shipped library has one schedule and reads none of them.
#ifdef RTX_TUNING
const char* tuning_env(const char* name) { return getenv(name); }
#else
const char* tuning_env(const char*) { return nullptr; }
#endif

template <class T>
int upload(DevBuf& b, const std::vector<T>& v, hipStream_t s)
{
    // slack behind the payload: kernels may load (never use) up to three triangle records past the end
    size_t bytes = v.size() * sizeof(T) + 128;
    HIP_TRY(b.ensure(bytes));
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return RTX_OK;
}

int grid_for(size_t n) { return (int)std::min<size_t>((n + kBlock - 1) / kBlock, kMaxGridBlocks); }
// kernels that fetch their work dynamically only need the workgroups that are resident at once; a launch too small to
// fill them gets up to 16 lanes per ray, which the traversal spends on walking a ray's pending subtrees side by side
int dyn_grid_for(const rtx_ctx* c, size_t n) { const size_t per = (size_t)c->rays_per_block; return (int)std::min<size_t>((n + per - 1) / per, (size_t)c->resident_blocks); }

// ---- scratch of the entry point that allocates nothing (rtx_trace_closest_device_noalloc) -------------------
// The closest-hit kernel needs two pieces of device memory beyond rays and hits: the work counters its waves draw
// ray chunks from, and the part of the per-lane traversal stacks that does not fit in LDS. The other entry points
// keep both in buffers that grow on demand (hipMalloc); this one keeps them in a file-scope device array of fixed
// size, so that it can be called where allocation is not allowed. Every GPU the library runs on holds its own copy
// (kNoAllocScratchBytes, ~60 MiB) until the process ends; all calls on a GPU share it, so calls that overlap on one
// GPU (two streams, two contexts) race.
constexpr int kNaBlocks = 1024;            // workgroups of a launch on the static scratch (4 per CU)
constexpr int kNaSpillEntries = 30;        // stack entries per lane beyond the kLdsStack entries in LDS
struct NoAllocScratch {
    uint32_t work[kWorkSlotWords];                                     // zeroed by every call
    uint2 spill[(size_t)kNaSpillEntries * kNaBlocks * kBlock];         // written before read, never cleared
};
constexpr size_t kNoAllocScratchBytes = sizeof(NoAllocScratch);
static_assert(kLdsStack + kNaSpillEntries == RTX_NOALLOC_MAX_STACK, "include/rtx.h states the stack limit of the no-allocation entry point");
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

// `scratch` names the array for the compiler only; its address on a GPU comes from hipGetSymbolAddress with that
// GPU current (each GPU has its own copy at its own address). Resolved once per device by rtx_create.
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mutex;

int resolve_scratch(int device, void** out)
{
    if (device < 0 || device >= kMaxDevices) return fail(RTX_ERR_INVALID, "device %d beyond the %d the library keeps scratch for", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));       // the caller has made `device` current
        if (!p) return fail(RTX_ERR_HIP, "hipGetSymbolAddress returned a null address for the static scratch");
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return RTX_OK;
}

int ensure_spill(rtx_ctx* c)
{
    size_t bytes = std::max<size_t>((size_t)c->spill_lanes * c->spill_entries * kMaxGridBlocks * kBlock * sizeof(uint2), 16);
    HIP_TRY(c->b_spill.ensure(bytes));
    return RTX_OK;
}

} // namespace

// Host-side copies between the caller's buffers and the pinned slots run on threads that stay around: starting sixteen
// threads per 8 MB slot (what parallel_ranges does, fine for a scene upload) cost more than the copy itself.
namespace {
struct CopyPool {
    std::mutex callers, m;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> workers;
    const std::function<void(uint64_t, uint64_t)>* fn = nullptr;
    uint64_t n = 0, per = 0, generation = 0;
    int pending = 0;
    bool stop = false;
    explicit CopyPool(int threads)
    {
        for (int id = 0; id < threads; id++) workers.emplace_back([this, id]() {
            uint64_t seen = 0;
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv_work.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                const uint64_t b = (uint64_t)id * per, e = std::min(n, b + per);
                const auto* f = fn;
                lk.unlock();
                if (b < e) (*f)(b, e);
                lk.lock();
                if (--pending == 0) cv_done.notify_one();
            }
        });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
    }
    void run(uint64_t count, const std::function<void(uint64_t, uint64_t)>& f)
    {
        if (count < (1u << 20)) { f(0, count); return; }
        std::lock_guard<std::mutex> one_caller(callers);
        std::unique_lock<std::mutex> lk(m);
        const uint64_t parts = workers.size();
        fn = &f; n = count; per = ((count + parts - 1) / parts + 4095) & ~(uint64_t)4095; pending = (int)parts; generation++;
        cv_work.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};
// The pool belongs to the contexts: the first host-buffer call makes it, the last rtx_destroy joins its threads.
std::mutex g_pool_mutex;
CopyPool* g_pool = nullptr;
int g_contexts = 0;
CopyPool& copy_pool()
{
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    if (!g_pool) g_pool = new CopyPool((int)std::min(16u, std::max(2u, std::thread::hardware_concurrency())));
    return *g_pool;
}
void context_born() { std::lock_guard<std::mutex> lk(g_pool_mutex); g_contexts++; }
void context_gone()
{
    CopyPool* last = nullptr;
    { std::lock_guard<std::mutex> lk(g_pool_mutex); if (--g_contexts == 0) { last = g_pool; g_pool = nullptr; } }
    delete last;            // joins the threads (outside the lock: nothing else can reach the pool any more)
}
} // namespace

extern "C" {

const char* rtx_last_error(void) { return g_err; }

static int create_resources(rtx_ctx* c, int device);

int rtx_create(rtx_ctx** out, int device)
{
    if (!out) return fail(RTX_ERR_INVALID, "out is null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(RTX_ERR_INVALID, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    rtx_ctx* c = new (std::nothrow) rtx_ctx;
    if (!c) return fail(RTX_ERR_NOMEM, "out of host memory");
    context_born();
    const int rc = create_resources(c, device);
    if (rc) { rtx_destroy(c); return rc; }      // streams, events and buffers made so far are released (g_err keeps the cause)
    *out = c;
    return RTX_OK;
}

static int create_resources(rtx_ctx* c, int device)
{
    c->device = device;
    c->variant = 1;
    if (const char* v = tuning_env("RTX_VARIANT")) c->variant = atoi(v);
    { int rc = resolve_scratch(device, &c->na_scratch); if (rc) return rc; }
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    {
        // workgroups of the traversal kernels that fit on the device at once (LDS-bound)
        int per_cu = 0; hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_closest<false>, kBlock, 0));
        c->resident_blocks = std::max(1, std::min(kMaxGridBlocks, per_cu * prop.multiProcessorCount));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_ir_trace_packet, kBlock, 0));
#ifdef RTX_TUNING
        { int pc = 0; HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, k_closest_packet, kBlock, 0)); per_cu = std::min(per_cu, pc); }
#endif
        c->resident_packet_blocks = std::max(1, std::min(kMaxGridBlocks, per_cu * prop.multiProcessorCount));
        if (const char* v = tuning_env("RTX_TAIL_ROUNDS")) c->tail_rounds = std::max(0, std::min(2, atoi(v)));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pt_tail<1>, kBlock, 0));
        c->resident_tail_blocks[0] = std::max(1, std::min(kMaxGridBlocks, per_cu * prop.multiProcessorCount));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pt_tail<2>, kBlock, 0));
        c->resident_tail_blocks[1] = std::max(1, std::min(kMaxGridBlocks, per_cu * prop.multiProcessorCount));
        if (const char* v = tuning_env("RTX_IR_PACKET")) c->ir_packet = atoi(v);
        if (const char* v = tuning_env("RTX_IR_BLOCK_CULL")) c->ir_block_cull = atoi(v) != 0;
        if (const char* v = tuning_env("RTX_PRIMARY_PACKET")) c->primary_packet = atoi(v);
        if (const char* v = tuning_env("RTX_QUERY_PACKET")) c->query_packet = atoi(v);
        if (const char* g = tuning_env("RTX_GRID")) c->resident_blocks = std::max(1, std::min(kMaxGridBlocks, atoi(g)));   // tuning override
        if (tuning_env("RTX_DEBUG_COUNTERS")) fprintf(stderr, "[rtx] %d CUs x %d resident traversal workgroups\n", prop.multiProcessorCount, per_cu);
    }
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_tiles, hipEventDisableTiming));
    for (int i = 0; i < kPipeline; i++) {
        // (lane 0 runs on the caller's stream: no stream of its own; lanes beyond the third get theirs when they are first used)
        if (i == 1 || i == 2) HIP_TRY(hipStreamCreateWithFlags(&c->pipe_stream[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->pipe_film[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->pipe_done[i], hipEventDisableTiming));
    }
    if (const char* v = tuning_env("RTX_RAYS_PER_BLOCK")) c->rays_per_block = std::max(1, std::min(kBlock, atoi(v)));
    if (const char* v = tuning_env("RTX_PIPELINE")) c->pipeline = std::max(0, std::min(kPipeline, atoi(v)));
    c->tail_stats = tuning_env("RTX_DEBUG_COUNTERS") != nullptr;
    if (const char* v = tuning_env("RTX_LT_REGROUP")) c->lt_regroup = atoi(v) != 0;
    if (const char* v = tuning_env("RTX_TAIL_GRID")) c->tail_grid_limit = std::max(0, atoi(v));
    if (const char* v = tuning_env("RTX_TAIL_FROM")) c->tail_from_depth = (uint32_t)std::max(0, atoi(v));
    if (const char* v = tuning_env("RTX_REGROUP_FROM")) c->regroup_from_depth = (uint32_t)std::max(0, atoi(v));
    HIP_TRY(hipHostMalloc((void**)&c->depth_stats, 96 * sizeof(uint32_t), hipHostMallocDefault));
    for (int i = 0; i < 96; i++) c->depth_stats[i] = 0xFFFFFFFFu;
    HIP_TRY(c->b_counters.ensure(4 * 5 * kWorkSlotWords));   // instrumentation counters, then the work-counter slots of the closest-hit, visibility and packet launches (+ the length of the packet kernel's rest queue)
    // RayTracer's defaults, RTBase/Settings.h:40-61
    rtx_settings& s = c->settings;
    s.algorithm = RTX_AL_PATH_TRACE; s.draw_mode = RTX_DM_ALGORITHM; s.filter = RTX_FT_BOX;
    s.use_mis = 0; s.can_hit_light = 0; s.adaptive_sampling = 0;
    s.max_bounces = 5; s.init_spp = 10; s.total_spp = 8192; s.vpl_rays_per_tile = 1; s.num_threads = 3;
    s.seed = 1; s.shard_mode = 0; s.shard_rank = 0; s.shard_world = 1;
    return RTX_OK;
}

void rtx_destroy(rtx_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    wavefront_release(c);
    for (int i = 0; i < kPipeline; i++) {
        if (c->pipe_stream[i]) (void)hipStreamDestroy(c->pipe_stream[i]);
        if (c->pipe_film[i]) (void)hipEventDestroy(c->pipe_film[i]);
        if (c->pipe_done[i]) (void)hipEventDestroy(c->pipe_done[i]);
    }
    for (auto& e : c->io_kernel_done) if (e) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_tiles) (void)hipEventDestroy(c->ev_tiles);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->depth_stats) (void)hipHostFree(c->depth_stats);
    delete c;
    context_gone();
}

// new scene, camera or settings: what the last passes reported about their queue sizes says nothing about the next ones
static void forget_depth_stats(rtx_ctx* c) { if (c->depth_stats) for (int i = 0; i < 96; i++) c->depth_stats[i] = 0xFFFFFFFFu; }

int rtx_set_stream(rtx_ctx* c, void* s)
{
    CTX_TRY(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = s == RTX_STREAM_OWN ? c->own_stream : (hipStream_t)s;      // NULL is HIP's null stream, as for any hipStream_t
    return RTX_OK;
}

int rtx_synchronize(rtx_ctx* c) { CTX_TRY(c); HIP_TRY(hipStreamSynchronize(c->stream)); return RTX_OK; }

int rtx_upload_scene(rtx_ctx* c, const rtx_scene_desc* d)
{
    CTX_TRY(c);
    if (!d) return fail(RTX_ERR_INVALID, "scene is null");
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_scene = false;
    forget_depth_stats(c);
    // the BVH is built on the device unless RTX_BVH_BUILD=host asks for the host build (same tree)
    const char* where = getenv("RTX_BVH_BUILD");
    Bvh2BuildFn device_build = [c](const std::vector<TriPrep>& t, const float* mn, const float* mx, Bvh2& out) {
        hipError_t e = build_bvh2_device(t, mn, mx, out, c->stream);
        if (e != hipSuccess) fail(RTX_ERR_HIP, "device BVH build: %s", hipGetErrorString(e));
        return e == hipSuccess;
    };
    std::string err = prepare_scene(*d, c->host, (where && std::string(where) == "host") ? Bvh2BuildFn() : device_build);
    if (!err.empty()) return fail(RTX_ERR_INVALID, "scene rejected: %s", err.c_str());
    const HostScene& h = c->host;
    hipStream_t s = c->stream;
    int rc;
    if ((rc = upload(c->b_nodes4, h.bvh4.nodes, s))) return rc;
    if ((rc = upload(c->b_nodes2, h.bvh4.nodes2, s))) return rc;
    if ((rc = upload(c->b_tris, h.tris, s))) return rc;
    if ((rc = upload(c->b_tri_ids, h.bvh4.order, s))) return rc;
    if ((rc = upload(c->b_shade, h.shade, s))) return rc;
    if ((rc = upload(c->b_materials, h.materials, s))) return rc;
    if ((rc = upload(c->b_texels, h.texels, s))) return rc;
    if ((rc = upload(c->b_lights, h.lights, s))) return rc;
    if ((rc = upload(c->b_env_texels, h.env_texels, s))) return rc;
    if ((rc = upload(c->b_env_rows, h.env_cdf_rows, s))) return rc;
    if ((rc = upload(c->b_env_cols, h.env_cdf_cols, s))) return rc;
    if ((rc = upload(c->b_env_guide_rows, h.env_guide_rows, s))) return rc;
    if ((rc = upload(c->b_env_guide_cols, h.env_guide_cols, s))) return rc;
    HIP_TRY(hipStreamSynchronize(s));

    DeviceScene& ds = c->dscene;
    ds.nodes4 = c->b_nodes4.as<float4>(); ds.nodes2 = c->b_nodes2.as<float4>(); ds.tris = c->b_tris.as<TriIsect>(); ds.tri_ids = c->b_tri_ids.as<uint32_t>();
    ds.shade = c->b_shade.as<TriShade>(); ds.materials = c->b_materials.as<MaterialDev>(); ds.texels = c->b_texels.as<float>();
    ds.lights = c->b_lights.as<LightDev>(); ds.n_lights = (uint32_t)h.lights.size(); ds.n_tris = (uint32_t)h.tris.size();
    for (int a = 0; a < 3; a++) { ds.root_min[a] = h.bvh4.root_min[a]; ds.root_max[a] = h.bvh4.root_max[a]; ds.bg_emission[a] = h.bg_emission[a]; ds.scene_centre[a] = h.scene_centre[a]; }
    ds.bg_kind = h.bg_kind; ds.env_w = h.env_w; ds.env_h = h.env_h;
    ds.env_texels = c->b_env_texels.as<float>(); ds.env_cdf_rows = c->b_env_rows.as<float>(); ds.env_cdf_cols = c->b_env_cols.as<float>();
    ds.env_guide_rows = c->b_env_guide_rows.as<uint32_t>(); ds.env_guide_cols = c->b_env_guide_cols.as<uint32_t>();
    ds.env_inv_w = h.env_w ? 1.0f / (float)h.env_w : 0.f; ds.env_inv_h = h.env_h ? 1.0f / (float)h.env_h : 0.f;
    ds.scene_radius = h.scene_radius;
    ds.spread_below = 32;
    ds.fine_per_wave = 192;
    if (const char* v = tuning_env("RTX_FINE")) ds.fine_per_wave = (uint32_t)std::max(0, std::min(4096, atoi(v)));   // tuning override
    if (const char* v = tuning_env("RTX_SPREAD")) ds.spread_below = (uint32_t)std::max(0, std::min(32, atoi(v)));   // tuning override

    // traversal stack: at most three pushes per wide level, or one pending node per binary
    // level on the exact path; entries beyond the LDS part go to global memory
    uint32_t need = std::max(3u * h.bvh4.depth + 2u, h.bvh2.depth + 3u);
    // (sized for the shallowest LDS part any kernel variant uses)
    c->stack_need = need;
    c->spill_entries = need > (uint32_t)kMinLdsStack ? need - kMinLdsStack : 0;
    if ((rc = ensure_spill(c))) return rc;

    c->stats.bvh2_nodes = (uint32_t)h.bvh2.nodes.size(); c->stats.bvh2_depth = h.bvh2.depth;
    c->stats.bvh4_nodes = (uint32_t)h.bvh4.nodes.size(); c->stats.bvh4_depth = h.bvh4.depth;
    c->stats.n_lights = ds.n_lights; c->stats.build_ms = h.build_ms;
    c->have_scene = true;
    return RTX_OK;
}

int rtx_camera_look_at(rtx_camera* out, const float from[3], const float to[3], const float up[3],
                       float fov_deg, int width, int height, int flip_x)
{
    if (!out || !from || !to || !up) return fail(RTX_ERR_INVALID, "null argument");
    if (width < 1 || height < 1) return fail(RTX_ERR_INVALID, "bad film size %dx%d", width, height);
    camera_look_at(*out, from, to, up, fov_deg, width, height, flip_x);
    return RTX_OK;
}

int rtx_host_transform_instance(const float world[16], size_t n, float* positions, float* normals)
{
    if (!world || (n && (!positions || !normals))) return fail(RTX_ERR_INVALID, "null argument");
    transform_instance(world, n, positions, normals);
    return RTX_OK;
}

int rtx_set_camera(rtx_ctx* c, const rtx_camera* cam)
{
    CTX_TRY(c);
    if (!cam) return fail(RTX_ERR_INVALID, "camera is null");
    if (cam->width < 1 || cam->height < 1 || (uint64_t)cam->width * cam->height > (1ull << 28)) return fail(RTX_ERR_INVALID, "bad film size %dx%d", cam->width, cam->height);
    HIP_TRY(hipStreamSynchronize(c->stream));
    forget_depth_stats(c);
    bool resized = !c->have_camera || cam->width != c->cam.width || cam->height != c->cam.height;
    c->cam = *cam;
    CameraDev& d = c->dcam;
    std::memcpy(d.proj, cam->proj, 64); std::memcpy(d.inv_proj, cam->inv_proj, 64);
    std::memcpy(d.camera, cam->camera, 64); std::memcpy(d.camera_to_view, cam->camera_to_view, 64);
    for (int a = 0; a < 3; a++) { d.origin[a] = cam->origin[a]; d.view_dir[a] = cam->view_dir[a]; }
    d.afilm = cam->afilm; d.width = (float)cam->width; d.height = (float)cam->height; d.w = cam->width; d.h = cam->height;
    c->have_camera = true;
    if (resized) {
        // Film::init, RTBase/Imaging.h:388-395; tile grid, RTBase/Renderer.h:107-112
        c->tiles_x = (cam->width + 15) / 16; c->tiles_y = (cam->height + 15) / 16;
        HIP_TRY(c->b_film.ensure((size_t)cam->width * cam->height * 3 * sizeof(float)));
        HIP_TRY(c->b_tile_samples.ensure((size_t)c->tiles_x * c->tiles_y * sizeof(uint32_t)));
        int rc = wavefront_resize(c); if (rc) return fail(rc, "wavefront buffers: allocation failed");
        return rtx_clear(c);
    }
    return RTX_OK;
}

int rtx_set_settings(rtx_ctx* c, const rtx_settings* s)
{
    CTX_TRY(c);
    if (!s) return fail(RTX_ERR_INVALID, "settings is null");
    if (s->algorithm < 0 || s->algorithm > 2 || s->draw_mode < 0 || s->draw_mode > 3 || s->filter < 0 || s->filter > 2)
        return fail(RTX_ERR_INVALID, "enum out of range");
    if (s->shard_mode < 0 || s->shard_mode > 2 || s->shard_world < 1 || s->shard_rank < 0 || s->shard_rank >= s->shard_world)
        return fail(RTX_ERR_INVALID, "bad shard (mode %d rank %d world %d)", s->shard_mode, s->shard_rank, s->shard_world);
    if (s->max_bounces > 64) return fail(RTX_ERR_INVALID, "max_bounces %u too large", s->max_bounces);
    c->settings = *s;
    forget_depth_stats(c);
    // RayTracer::init, RTBase/Renderer.h:81-84
    c->settings.adaptive_sampling = s->adaptive_sampling && s->algorithm == RTX_AL_PATH_TRACE;
    if (!c->settings.adaptive_sampling) c->settings.init_spp = s->total_spp;
    return RTX_OK;
}

int rtx_clear(rtx_ctx* c)
{
    CTX_TRY(c);
    if (!c->have_camera) return fail(RTX_ERR_STATE, "no camera set");
    HIP_TRY(hipMemsetAsync(c->b_film.p, 0, (size_t)c->cam.width * c->cam.height * 3 * sizeof(float), c->stream));
    c->spp = 0;
    return RTX_OK;
}

int rtx_get_spp(rtx_ctx* c, int* spp) { if (!c || !spp) return fail(RTX_ERR_INVALID, "null argument"); *spp = c->spp; return RTX_OK; }
int rtx_set_spp(rtx_ctx* c, int spp) { if (!c || spp < 0) return fail(RTX_ERR_INVALID, "bad argument"); c->spp = spp; return RTX_OK; }

int rtx_render_passes(rtx_ctx* c, int n)
{
    CTX_TRY(c);
    if (!c->have_scene || !c->have_camera) return fail(RTX_ERR_STATE, "scene and camera must be set before rendering");
    if (n < 0) return fail(RTX_ERR_INVALID, "negative pass count");
    // Up to c->pipeline passes are in flight, each on a stream and a set of path state of its
    // own: late bounces hold few rays and a launch then lasts as long as its slowest ray, so
    // the passes fill each other's idle lanes. The film is still summed in pass order.
    {
        // adaptive sampling (RTBase/Renderer.h:776-781): from pass initSPP on only the tiles that still want samples are
        // rendered, a different workload; a call that crosses that pass is rendered as two, each with its own number of lanes
        const rtx_settings& as = c->settings;
        const bool adaptive = as.adaptive_sampling && as.algorithm != RTX_AL_LIGHT_TRACE && as.shard_mode != 2;
        if (adaptive && c->spp < (int)as.init_spp && (long long)c->spp + n > (long long)as.init_spp) {
            const int first = (int)as.init_spp - c->spp;
            int rc = rtx_render_passes(c, first);
            return rc ? rc : rtx_render_passes(c, n - first);
        }
    }
    { int rc = wavefront_alloc(c, 0, c->settings); if (rc) return fail(rc, "path state: allocation failed"); }
    // (three as a rule; a small film, or a small share of the tiles of one, leaves more of the device idle at the end of
    // every launch and takes a fourth: 640x360 0.194 -> 0.163 ms per pass, a quarter of 1080p 0.287 -> 0.245, an eighth
    // 0.198 -> 0.170, but half of 1080p 0.412 -> 0.422 and 1080p itself 0.70 -> 0.72)
    // (and the sparse passes of adaptive sampling: 2 080 -> 2 155 spp/s on the mixed scene, 10 -> 256 samples)
    const bool sparse_passes = c->settings.adaptive_sampling && c->settings.algorithm != RTX_AL_LIGHT_TRACE && c->settings.shard_mode != 2 && c->spp >= (int)c->settings.init_spp;
    const int in_flight = c->pipeline ? c->pipeline : ((c->wf[0]->p.P < 800000u || sparse_passes) ? 4 : 3);
    const int lanes = std::max(1, std::min(n, in_flight));
    c->passes_in_flight = lanes;
    for (int l = 1; l < lanes; l++) { int rc = wavefront_alloc(c, l, c->settings); if (rc) return fail(rc, "path state: allocation failed"); }
    if (lanes == 1) {
        for (int i = 0; i < n; i++) {
            int rc = wavefront_render_pass(c, 0, c->stream, nullptr, nullptr);
            if (rc) return fail(rc, "render pass failed: %s", hipGetErrorString(hipGetLastError()));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        return RTX_OK;
    }
    if (c->spill_lanes < lanes) { c->spill_lanes = lanes; int rc = ensure_spill(c); if (rc) return rc; }
    // lane 0 runs on the caller's stream itself, the others on streams of their own
    hipStream_t ls[kPipeline];
    for (int l = 1; l < lanes; l++) if (!c->pipe_stream[l]) HIP_TRY(hipStreamCreateWithFlags(&c->pipe_stream[l], hipStreamNonBlocking));
    for (int l = 0; l < lanes; l++) ls[l] = l == 0 ? c->stream : c->pipe_stream[l];
    HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
    for (int l = 1; l < lanes; l++) HIP_TRY(hipStreamWaitEvent(ls[l], c->ev_fork, 0));
    const rtx_settings& st = c->settings;
    // an error in the middle leaves passes enqueued on the other lanes: nothing returns before they have drained
    auto drain = [&]() { for (int l = 0; l < lanes; l++) (void)hipStreamSynchronize(ls[l]); };
#define RTX_PIPE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { drain(); return fail(RTX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
    for (int i = 0; i < n; i++) {
        const int l = i % lanes;
        int rc = wavefront_render_pass(c, l, ls[l], i > 0 ? c->pipe_film[(i - 1) % lanes] : nullptr, c->pipe_film[l]);
        if (rc) { const hipError_t e = hipGetLastError(); drain(); return fail(rc, "render pass failed: %s", hipGetErrorString(e)); }
        if (c->spp == (int)st.init_spp && st.algorithm != RTX_AL_LIGHT_TRACE && st.shard_mode != 2) {
            // the adaptive sample targets were just computed on this lane: later passes read them
            RTX_PIPE_TRY(hipEventRecord(c->ev_tiles, ls[l]));
            for (int k = 0; k < lanes; k++) if (k != l) RTX_PIPE_TRY(hipStreamWaitEvent(ls[k], c->ev_tiles, 0));
        }
    }
    for (int l = 1; l < lanes; l++) {
        RTX_PIPE_TRY(hipEventRecord(c->pipe_done[l], ls[l]));
        RTX_PIPE_TRY(hipStreamWaitEvent(c->stream, c->pipe_done[l], 0));
    }
#undef RTX_PIPE_TRY
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTX_OK;
}

static int stage_out(rtx_ctx* c, const void* dev, size_t bytes, void* host);   // (below, with the other host-buffer plumbing)

int rtx_film_read(rtx_ctx* c, float* rgb, int* spp, uint32_t* tiles)
{
    CTX_TRY(c);
    if (!c->have_camera) return fail(RTX_ERR_STATE, "no camera set");
    if (tiles) HIP_TRY(hipMemcpyAsync(tiles, c->b_tile_samples.p, (size_t)c->tiles_x * c->tiles_y * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (rgb) { int rc = stage_out(c, c->b_film.p, (size_t)c->cam.width * c->cam.height * 3 * sizeof(float), rgb); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (spp) *spp = c->spp;
    return RTX_OK;
}

int rtx_film_write(rtx_ctx* c, const float* rgb, int spp)
{
    CTX_TRY(c);
    if (!c->have_camera) return fail(RTX_ERR_STATE, "no camera set");
    if (!rgb || spp < 0) return fail(RTX_ERR_INVALID, "bad argument");
    HIP_TRY(hipMemcpyAsync(c->b_film.p, rgb, (size_t)c->cam.width * c->cam.height * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->spp = spp;
    return RTX_OK;
}

int rtx_film_device(rtx_ctx* c, void** p, size_t* n)
{
    if (!c || !p) return fail(RTX_ERR_INVALID, "null argument");
    if (!c->have_camera) return fail(RTX_ERR_STATE, "no camera set");
    *p = c->b_film.p; if (n) *n = (size_t)c->cam.width * c->cam.height * 3;
    return RTX_OK;
}

int rtx_tile_samples_device(rtx_ctx* c, void** p, size_t* n)
{
    if (!c || !p) return fail(RTX_ERR_INVALID, "null argument");
    if (!c->have_camera) return fail(RTX_ERR_STATE, "no camera set");
    *p = c->b_tile_samples.p; if (n) *n = (size_t)c->tiles_x * c->tiles_y;
    return RTX_OK;
}

int rtx_update_tile_samples(rtx_ctx* c)
{
    CTX_TRY(c);
    if (!c->have_camera) return fail(RTX_ERR_STATE, "no camera set");
    int rc = wavefront_tile_samples(c, c->stream);
    if (rc) return fail(rc, "tile statistics kernel failed");
    return RTX_OK;
}

// ---- denoiser inputs -----------------------------------------------------------------------

int rtx_create_aov_device(rtx_ctx* c, float* color, float* albedo, float* normal)
{
    CTX_TRY(c);
    if (!c->have_scene || !c->have_camera) return fail(RTX_ERR_STATE, "scene and camera must be set first");
    int rc = wavefront_aov(c, color, albedo, normal);
    if (rc) return fail(rc, "AOV pass failed");
    return RTX_OK;
}

int rtx_create_aov(rtx_ctx* c, float* color, float* albedo, float* normal)
{
    CTX_TRY(c);
    if (!c->have_scene || !c->have_camera) return fail(RTX_ERR_STATE, "scene and camera must be set first");
    const size_t plane = (size_t)c->cam.width * c->cam.height * 3 * sizeof(float);
    HIP_TRY(c->b_io_a.ensure(plane * 3));
    float* d = c->b_io_a.as<float>(); const size_t n = plane / sizeof(float);
    int rc = rtx_create_aov_device(c, color ? d : nullptr, albedo ? d + n : nullptr, normal ? d + 2 * n : nullptr);
    if (rc) return rc;
    if (color) HIP_TRY(hipMemcpyAsync(color, d, plane, hipMemcpyDeviceToHost, c->stream));
    if (albedo) HIP_TRY(hipMemcpyAsync(albedo, d + n, plane, hipMemcpyDeviceToHost, c->stream));
    if (normal) HIP_TRY(hipMemcpyAsync(normal, d + 2 * n, plane, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTX_OK;
}

// ---- ray queries -------------------------------------------------------------------------

int rtx_trace_closest_device(rtx_ctx* c, const rtx_ray* rays, size_t n, rtx_hit* hits)
{
    CTX_TRY(c);
    if (!c->have_scene) return fail(RTX_ERR_STATE, "no scene uploaded");
    if (n == 0) return RTX_OK;
    if (!rays || !hits) return fail(RTX_ERR_INVALID, "null buffer");
    if (n > 0xF0000000ull) return fail(RTX_ERR_INVALID, "too many rays in one call");
#ifdef RTX_TUNING
    if (c->variant == 0) {
        // A/B baseline: static assignment of rays to lanes (RTX_VARIANT=0)
        hipLaunchKernelGGL(k_closest_static, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, c->dscene, rays, (uint32_t)n, hits,
                           c->b_spill.as<uint2>(), (uint32_t)(kMaxGridBlocks * kBlock));
    } else if (c->variant >= 2) {
        uint32_t* wc = c->b_counters.as<uint32_t>() + kWorkSlotWords;
        HIP_TRY(hipMemsetAsync(wc, 0, kWorkSlotWords * 4, c->stream));
        const uint32_t gb = (uint32_t)dyn_grid_for(c, n);
#define RTX_AB(V, R, U, S, L, W, SH) if (c->variant == V) hipLaunchKernelGGL((k_closest_ab<R, U, S, L, W, SH>), dim3(gb), dim3(kBlock), 0, c->stream, c->dscene, rays, \
                           (uint32_t)n, hits, c->b_spill.as<uint2>(), (uint32_t)(kMaxGridBlocks * kBlock), wc);
        RTX_AB(2, 20, 4, 28, 12, 4, 0) RTX_AB(3, 16, 4, 24, 12, 4, 1) RTX_AB(4, 24, 4, 32, 12, 4, 1)  // no tail sharing; neighbours of the default (20 idle lanes to refill, 28 leaf lanes to stop descending)
#undef RTX_AB
    } else
#endif
    {
        uint32_t* wc = c->b_counters.as<uint32_t>() + kWorkSlotWords;
        const uint32_t* queue = nullptr; const uint32_t* n_dev = nullptr;
#ifdef RTX_TUNING
        if (c->query_packet && c->stack_need <= (uint32_t)kPacketStack && n >= 4096) {
            // coherent runs of 64 rays (camera rays in pixel-block order) are walked as packets, one traversal per wave;
            // what does not fit goes to the per-lane kernel below through a queue
            HIP_TRY(c->b_rest.ensure(n * sizeof(uint32_t)));
            uint32_t* pw = c->b_counters.as<uint32_t>() + 3 * kWorkSlotWords;
            uint32_t* rest_count = pw + kWorkSlotWords;
            HIP_TRY(hipMemsetAsync(pw, 0, (kWorkSlotWords + 1) * 4, c->stream));
            const size_t packets = (n + 63) / 64;
            hipLaunchKernelGGL(k_closest_packet, dim3((unsigned)std::min<size_t>((packets + 3) / 4, (size_t)c->resident_packet_blocks)), dim3(kBlock), 0, c->stream, c->dscene, rays,
                               (const uint32_t*)nullptr, (uint32_t)n, (const uint32_t*)nullptr, hits, c->b_spill.as<uint2>(), (uint32_t)(kMaxGridBlocks * kBlock), pw, c->b_rest.as<uint32_t>(), rest_count);
            queue = c->b_rest.as<uint32_t>(); n_dev = rest_count;
        }
#endif
        HIP_TRY(hipMemsetAsync(wc, 0, kWorkSlotWords * 4, c->stream));
        hipLaunchKernelGGL(k_closest<false>, dim3(dyn_grid_for(c, n)), dim3(kBlock), 0, c->stream, c->dscene, rays, queue,
                           (uint32_t)n, n_dev, hits, c->b_spill.as<uint2>(), (uint32_t)(kMaxGridBlocks * kBlock), wc,
                           (unsigned long long*)nullptr);
    }
    HIP_TRY(hipGetLastError());
    c->stats.closest_rays += n;
    return RTX_OK;
}

int rtx_trace_closest_device_noalloc(rtx_ctx* c, const rtx_ray* rays, size_t n, rtx_hit* hits)
{
    CTX_TRY(c);
    if (!c->have_scene) return fail(RTX_ERR_STATE, "no scene uploaded");
    if (n == 0) return RTX_OK;
    if (!rays || !hits) return fail(RTX_ERR_INVALID, "null buffer");
    if (n > RTX_NOALLOC_MAX_RAYS) return fail(RTX_ERR_INVALID, "%zu rays in one call, the no-allocation entry point takes at most %llu", n, (unsigned long long)RTX_NOALLOC_MAX_RAYS);
    if (c->stack_need > (uint32_t)(kLdsStack + kNaSpillEntries))
        return fail(RTX_ERR_INVALID, "the scene's tree needs %u traversal-stack entries per ray, the no-allocation entry point holds %d (RTX_NOALLOC_MAX_STACK)",
                    c->stack_need, kLdsStack + kNaSpillEntries);
    // from here on: the device address resolved at rtx_create, never the array's name
    NoAllocScratch* s = (NoAllocScratch*)c->na_scratch;
    HIP_TRY(hipMemsetAsync(s->work, 0, sizeof s->work, c->stream));
    const int grid = std::min(dyn_grid_for(c, n), kNaBlocks);
    hipLaunchKernelGGL(k_closest<false>, dim3(grid), dim3(kBlock), 0, c->stream, c->dscene, rays, (const uint32_t*)nullptr, (uint32_t)n,
                       (const uint32_t*)nullptr, hits, s->spill, (uint32_t)(kNaBlocks * kBlock), s->work, (unsigned long long*)nullptr);
    HIP_TRY(hipGetLastError());
    c->stats.closest_rays += n;
    return RTX_OK;
}

// Ray queries on host buffers, in chunks of one pinned slot: while the host threads copy chunk k into its slot, the DMA
// engines move chunk k - 1 in and the results of chunk k - 2 out (a stream each way, the kernels on the context's stream
// between them), so that the call costs about what the host-side copies cost instead of copies + transfers + kernel
// one after the other (a plain hipMemcpy from pageable memory moves ~10 GB/s). `launch` starts the query of `count`
// items at device addresses `in` / `out` on c->stream. Returns when `host_out` is complete.
static int staged_query(rtx_ctx* c, const void* host_in, size_t in_size, void* host_out, size_t out_size, size_t n,
                        const std::function<int(void*, size_t, void*)>& launch)
{
    HIP_TRY(c->b_io_a.ensure(n * in_size));
    HIP_TRY(c->b_io_b.ensure(n * out_size));
    HIP_TRY(c->pin_in.ensure());
    HIP_TRY(c->pin_out.ensure());
    for (auto& e : c->io_kernel_done) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipStream_t s_in = c->pipe_stream[1], s_out = c->pipe_stream[2];
    const size_t per = (PinBuf::kSlot / std::max(in_size, out_size)) & ~(size_t)63;
    const size_t n_chunks = (n + per - 1) / per;
    auto count_of = [&](size_t k) { return std::min(per, n - k * per); };
    auto drain = [&](size_t k) -> int {          // results of chunk k: pinned slot -> caller's buffer
        const size_t len = count_of(k) * out_size;
        const char* slot = c->pin_out.p + (k & 1) * PinBuf::kSlot;
        HIP_TRY(hipEventSynchronize(c->pin_out.ev[k & 1]));
        copy_pool().run(len, [&](uint64_t a, uint64_t b) { std::memcpy((char*)host_out + k * per * out_size + a, slot + a, b - a); });
        return RTX_OK;
    };
    auto body = [&]() -> int {
        for (size_t k = 0; k < n_chunks; k++) {
            const size_t cnt = count_of(k), p = k & 1;
            char* slot = c->pin_in.p + p * PinBuf::kSlot;
            if (k >= 2) HIP_TRY(hipEventSynchronize(c->pin_in.ev[p]));        // the transfer that last read this slot is done
            copy_pool().run(cnt * in_size, [&](uint64_t a, uint64_t b) { std::memcpy(slot + a, (const char*)host_in + k * per * in_size + a, b - a); });
            char* din = (char*)c->b_io_a.p + k * per * in_size; char* dout = (char*)c->b_io_b.p + k * per * out_size;
            HIP_TRY(hipMemcpyAsync(din, slot, cnt * in_size, hipMemcpyHostToDevice, s_in));
            HIP_TRY(hipEventRecord(c->pin_in.ev[p], s_in));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->pin_in.ev[p], 0));
            int rc = launch(din, cnt, dout); if (rc) return rc;
            HIP_TRY(hipEventRecord(c->io_kernel_done[p], c->stream));
            // (the pinned output slot of this parity was emptied by drain(k - 2), one trip ago)
            HIP_TRY(hipStreamWaitEvent(s_out, c->io_kernel_done[p], 0));
            HIP_TRY(hipMemcpyAsync(c->pin_out.p + p * PinBuf::kSlot, dout, cnt * out_size, hipMemcpyDeviceToHost, s_out));
            HIP_TRY(hipEventRecord(c->pin_out.ev[p], s_out));
            if (k >= 1) { rc = drain(k - 1); if (rc) return rc; }
        }
        return drain(n_chunks - 1);
    };
    const int rc = body();
    if (rc) { (void)hipStreamSynchronize(s_in); (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(s_out); }   // nothing may touch the pinned slots after the call
    return rc;
}

// Device buffer -> host buffer through the two pinned slots (transfer of one slot beside the host-side copy of the
// other); returns when `host` is complete. A plain hipMemcpy into pageable memory moves ~10 GB/s.
static int stage_out(rtx_ctx* c, const void* dev, size_t bytes, void* host)
{
    HIP_TRY(c->pin_out.ensure());
    const size_t n_slots = (bytes + PinBuf::kSlot - 1) / PinBuf::kSlot;
    auto issue = [&](size_t k) -> hipError_t {
        const size_t off = k * PinBuf::kSlot, len = std::min(PinBuf::kSlot, bytes - off);
        hipError_t e = hipMemcpyAsync(c->pin_out.p + (k & 1) * PinBuf::kSlot, (const char*)dev + off, len, hipMemcpyDeviceToHost, c->stream);
        return e == hipSuccess ? hipEventRecord(c->pin_out.ev[k & 1], c->stream) : e;
    };
    auto body = [&]() -> int {
        for (size_t k = 0; k < std::min<size_t>(2, n_slots); k++) HIP_TRY(issue(k));
        for (size_t k = 0; k < n_slots; k++) {
            const size_t off = k * PinBuf::kSlot, len = std::min(PinBuf::kSlot, bytes - off);
            const char* slot = c->pin_out.p + (k & 1) * PinBuf::kSlot;
            HIP_TRY(hipEventSynchronize(c->pin_out.ev[k & 1]));
            copy_pool().run(len, [&](uint64_t a, uint64_t b) { std::memcpy((char*)host + off + a, slot + a, b - a); });
            if (k + 2 < n_slots) HIP_TRY(issue(k + 2));
        }
        return RTX_OK;
    };
    const int rc = body();
    if (rc) (void)hipStreamSynchronize(c->stream);
    return rc;
}

int rtx_trace_closest(rtx_ctx* c, const rtx_ray* rays, size_t n, rtx_hit* hits)
{
    CTX_TRY(c);
    if (n == 0) return RTX_OK;
    if (!rays || !hits) return fail(RTX_ERR_INVALID, "null buffer");
    if (!c->have_scene) return fail(RTX_ERR_STATE, "no scene uploaded");
    if (n > 0xF0000000ull) return fail(RTX_ERR_INVALID, "too many rays in one call");
    return staged_query(c, rays, sizeof(rtx_ray), hits, sizeof(rtx_hit), n,
                        [&](void* in, size_t count, void* out) { return rtx_trace_closest_device(c, (const rtx_ray*)in, count, (rtx_hit*)out); });
}

int rtx_trace_visible_device(rtx_ctx* c, const float* p1p2, size_t n, uint8_t* vis)
{
    CTX_TRY(c);
    if (!c->have_scene) return fail(RTX_ERR_STATE, "no scene uploaded");
    if (n == 0) return RTX_OK;
    if (!p1p2 || !vis) return fail(RTX_ERR_INVALID, "null buffer");
    if (n > 0xF0000000ull) return fail(RTX_ERR_INVALID, "too many segments in one call");
    uint32_t* wc = c->b_counters.as<uint32_t>() + 2 * kWorkSlotWords;
    HIP_TRY(hipMemsetAsync(wc, 0, kWorkSlotWords * 4, c->stream));
#ifdef RTX_TUNING
    if (c->variant >= 2) {
#define RTX_ABV(V, R, U, S, SH) if (c->variant == V) hipLaunchKernelGGL((k_visible_ab<R, U, S, SH>), dim3(dyn_grid_for(c, n)), dim3(kBlock), 0, c->stream, c->dscene, p1p2, \
                           (uint32_t)n, vis, c->b_spill.as<uint2>(), (uint32_t)(kMaxGridBlocks * kBlock), wc);
        RTX_ABV(2, 20, 4, 28, 0) RTX_ABV(3, 16, 4, 24, 1) RTX_ABV(4, 24, 4, 32, 1)   // no tail sharing; neighbours of the default thresholds
#undef RTX_ABV
    } else
#endif
    hipLaunchKernelGGL(k_visible, dim3(dyn_grid_for(c, n)), dim3(kBlock), 0, c->stream, c->dscene, p1p2, (uint32_t)n, vis,
                       c->b_spill.as<uint2>(), (uint32_t)(kMaxGridBlocks * kBlock), wc);
    HIP_TRY(hipGetLastError());
    c->stats.shadow_rays += n;
    return RTX_OK;
}

int rtx_trace_visible(rtx_ctx* c, const float* p1p2, size_t n, uint8_t* vis)
{
    CTX_TRY(c);
    if (n == 0) return RTX_OK;
    if (!p1p2 || !vis) return fail(RTX_ERR_INVALID, "null buffer");
    if (!c->have_scene) return fail(RTX_ERR_STATE, "no scene uploaded");
    if (n > 0xF0000000ull) return fail(RTX_ERR_INVALID, "too many segments in one call");
    return staged_query(c, p1p2, 6 * sizeof(float), vis, 1, n,
                        [&](void* in, size_t count, void* out) { return rtx_trace_visible_device(c, (const float*)in, count, (uint8_t*)out); });
}

int rtx_generate_primary_rays_device(rtx_ctx* c, rtx_ray* rays, uint32_t* pixel, size_t cap)
{
    CTX_TRY(c);
    if (!c->have_camera) return fail(RTX_ERR_STATE, "no camera set");
    size_t n = (size_t)c->cam.width * c->cam.height;
    if (!rays) return fail(RTX_ERR_INVALID, "null buffer");
    if (cap < n) return fail(RTX_ERR_INVALID, "ray buffer holds %zu rays, film needs %zu", cap, n);
    hipLaunchKernelGGL(k_primary_rays, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, c->dcam, rays, pixel);
    HIP_TRY(hipGetLastError());
    return RTX_OK;
}

int rtx_count_closest_work(rtx_ctx* c, const rtx_ray* rays, size_t n, uint64_t* nodes, uint64_t* tris, uint64_t* tri_groups)
{
    CTX_TRY(c);
    if (!c->have_scene) return fail(RTX_ERR_STATE, "no scene uploaded");
    if (!rays || !nodes || !tris || n == 0 || n > 0xF0000000ull) return fail(RTX_ERR_INVALID, "bad argument");
    HIP_TRY(c->b_io_b.ensure(n * sizeof(rtx_hit)));
    HIP_TRY(hipMemsetAsync(c->b_counters.p, 0, 4 * 2 * kWorkSlotWords, c->stream));
    hipLaunchKernelGGL(k_closest<true>, dim3(dyn_grid_for(c, n)), dim3(kBlock), 0, c->stream, c->dscene, rays, (const uint32_t*)nullptr,
                       (uint32_t)n, (const uint32_t*)nullptr, c->b_io_b.as<rtx_hit>(), c->b_spill.as<uint2>(), (uint32_t)(kMaxGridBlocks * kBlock),
                       c->b_counters.as<uint32_t>() + kWorkSlotWords, c->b_counters.as<unsigned long long>());
    HIP_TRY(hipGetLastError());
    unsigned long long h[6];
    HIP_TRY(hipMemcpyAsync(h, c->b_counters.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (tuning_env("RTX_DEBUG_COUNTERS")) {
        // distribution of the dependent steps of a ray (node visits + groups of four triangle tests): the slowest
        // ray bounds how long a launch lasts after its queue has run dry
        std::vector<rtx_hit> hh(n);
        HIP_TRY(hipMemcpy(hh.data(), c->b_io_b.p, n * sizeof(rtx_hit), hipMemcpyDeviceToHost));
        std::vector<uint32_t> steps(n);
        for (size_t i = 0; i < n; i++) steps[i] = hh[i].t < FLT_MAX || hh[i].alpha > 0 ? (uint32_t)hh[i].alpha + ((uint32_t)hh[i].beta + 3) / 4 : 0;
        std::sort(steps.begin(), steps.end());
        double sum = 0; for (uint32_t v : steps) sum += v;
        fprintf(stderr, "[rtx] dependent steps per ray: mean %.1f, median %u, p90 %u, p99 %u, p99.9 %u, max %u\n", sum / (double)n, steps[n / 2], steps[n * 9 / 10],
                steps[n * 99 / 100], steps[(size_t)(n * 0.999)], steps[n - 1]);
    }
    if (tuning_env("RTX_DEBUG_COUNTERS"))
        fprintf(stderr, "[rtx] lane node visits %llu, lane tri tests %llu; wave inner trips %llu, wave tri trips %llu, wave outer trips %llu\n", h[0], h[1], h[2], h[3], h[4]);
    *nodes = h[0]; *tris = h[1];
    if (tri_groups) *tri_groups = h[5];
    return RTX_OK;
}

int rtx_count_render_work(rtx_ctx* c, int n_passes, uint64_t out[8])
{
    CTX_TRY(c);
    if (!c->have_scene || !c->have_camera) return fail(RTX_ERR_STATE, "scene and camera must be set before rendering");
    if (!out || n_passes < 1) return fail(RTX_ERR_INVALID, "bad argument");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemsetAsync(c->b_counters.p, 0, 4 * kWorkSlotWords, c->stream));
    for (int i = 0; i < 8; i++) out[i] = 0;
    std::vector<uint32_t> q(kCounterWords);
    const bool radiosity = c->settings.algorithm == RTX_AL_INSTANT_RADIOSITY && c->settings.draw_mode == RTX_DM_ALGORITHM;
    c->count_work = true; c->passes_in_flight = 1;
    { int rc = wavefront_alloc(c, 0, c->settings); if (rc) { c->count_work = false; return fail(rc, "path state: allocation failed"); } }
    for (int i = 0; i < n_passes; i++) {
        const uint64_t before = c->stats.samples;
        int rc = wavefront_render_pass(c, 0, c->stream, nullptr, nullptr);
        hipError_t e = rc ? hipSuccess : hipMemcpyAsync(q.data(), c->wf[0]->p.cnt, q.size() * 4, hipMemcpyDeviceToHost, c->stream);
        uint32_t vpls = 0;
        if (!rc && e == hipSuccess && radiosity) e = hipMemcpyAsync(&vpls, c->wf[0]->p.vpl_total, 4, hipMemcpyDeviceToHost, c->stream);
        if (!rc && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (rc || e != hipSuccess) { c->count_work = false; return fail(rc ? rc : RTX_ERR_HIP, "instrumented render pass failed: %s", hipGetErrorString(e)); }
        // queue lengths of the pass: live paths per depth (one closest-hit ray each), shadow / connection segments per depth
        // (instant radiosity: the counters are cleared between the VPL walk and the gather, so its light paths' few rays
        // are not in the sums; queue 1 of the gather holds the gathering pixels, which are no rays)
        // (dense queues keep their length in one word; the path tracer's queues behind depth 0 in kQueueStreams stream lengths)
        auto queue_length = [&](uint32_t qi) { uint32_t nq = q[qi * kQueueStride]; for (uint32_t sc = 0; sc < (uint32_t)kQueueStreams; sc++) nq += q[stream_word(qi, sc)]; return nq; };
        for (uint32_t d = 0; d < 96; d++) { if (!(radiosity && d == 1)) out[0] += queue_length(d); out[1] += queue_length(96 + d); }
        if (i == 0 && tuning_env("RTX_DEBUG_COUNTERS"))
            for (uint32_t d = 0; d <= c->settings.max_bounces + 1u && d < 95u; d++)
                fprintf(stderr, "[rtx] depth %u: %u live paths, %u shadow / connection segments\n", d, queue_length(d), queue_length(96 + d));
        out[4] += c->stats.samples - before; out[7] += vpls;
    }
    c->count_work = false;
    unsigned long long h[8];
    HIP_TRY(hipMemcpy(h, c->b_counters.p, sizeof h, hipMemcpyDeviceToHost));
    out[2] = h[0]; out[3] = h[1]; out[5] = h[5]; out[6] = h[6]; out[1] += h[6];
    if (tuning_env("RTX_DEBUG_COUNTERS"))
        fprintf(stderr, "[rtx] traversal trips of the instrumented passes, summed over waves: %llu node steps, %llu triangle steps, %llu outer trips\n", h[2], h[3], h[4]);
#ifdef RTX_TUNING
    {
        // (the instrumented passes above ran the per-lane gather; one more plain pass for the packet walk's statistics)
        HIP_TRY(hipMemset(c->b_counters.as<unsigned long long>() + 8, 0, 16 * 8));
        int rc = wavefront_render_pass(c, 0, c->stream, nullptr, nullptr);
        if (rc) return fail(rc, "render pass failed");
        HIP_TRY(hipStreamSynchronize(c->stream));
        unsigned long long p[8];
        HIP_TRY(hipMemcpy(p, c->b_counters.as<unsigned long long>() + 8, sizeof p, hipMemcpyDeviceToHost));
        unsigned long long ts[8];
        HIP_TRY(hipMemcpy(ts, c->b_counters.as<unsigned long long>() + 16, sizeof ts, hipMemcpyDeviceToHost));
        if (ts[2]) fprintf(stderr, "[rtx] tail kernel (100 MHz ticks): %llu waves with work, %.2f tasks and %.2f rounds each; per wave %.1f us in shade steps, %.1f us in traversals; longest wave %.1f us (shade %.1f, traversals %.1f)\n",
                           ts[2], (double)ts[3] / ts[2], (double)ts[4] / ts[2], ts[0] / 100.0 / ts[2], ts[1] / 100.0 / ts[2], ts[5] / 100.0, ts[6] / 100.0, ts[7] / 100.0);
        if (p[4]) fprintf(stderr, "[rtx] packet walk: %llu packets of %.1f segments; per packet %.2f node steps with %.1f lanes, %.2f triangle steps with %.1f lanes\n", p[4], (double)p[5] / p[4],
                          (double)p[0] / p[4], p[0] ? (double)p[1] / p[0] : 0.0, (double)p[2] / p[4], p[2] ? (double)p[3] / p[2] : 0.0);
        if (p[4]) fprintf(stderr, "[rtx] gather: %llu (block of 256 places, VPL) pairs culled whole; %llu of the node steps with at most 16 lanes\n", p[6], p[7]);
    }
#endif
    return RTX_OK;
}

// ---- function-level parity tables (test support) ------------------------------------------

int rtx_debug_shading_data(rtx_ctx* c, const rtx_ray* rays, const rtx_hit* hits, size_t n, float* out)
{
    CTX_TRY(c);
    if (!c->have_scene) return fail(RTX_ERR_STATE, "no scene uploaded");
    if (n == 0) return RTX_OK;
    if (!rays || !hits || !out || n > (1u << 26)) return fail(RTX_ERR_INVALID, "bad argument");
    const uint32_t n_tris = c->dscene.n_tris;
    for (size_t i = 0; i < n; i++) if (hits[i].t < FLT_MAX && hits[i].id >= n_tris) return fail(RTX_ERR_INVALID, "hit %zu names triangle %u of %u", i, hits[i].id, n_tris);
    return debug_table(c, rays, n * sizeof(rtx_ray), hits, n * sizeof(rtx_hit), out, n * 24 * sizeof(float), [&](void* a, void* b, void* o) {
        hipLaunchKernelGGL(k_debug_shading_data, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->dscene, (const rtx_ray*)a, (const rtx_hit*)b, (uint32_t)n, (float*)o); });
}

int rtx_debug_bsdf_table(rtx_ctx* c, uint32_t material, const float* in, size_t n, float* out)
{
    CTX_TRY(c);
    if (!c->have_scene) return fail(RTX_ERR_STATE, "no scene uploaded");
    if (n == 0) return RTX_OK;
    if (!in || !out || n > (1u << 26)) return fail(RTX_ERR_INVALID, "bad argument");
    if (material >= c->host.materials.size()) return fail(RTX_ERR_INVALID, "material %u of %zu", material, c->host.materials.size());
    return debug_table(c, in, n * 14 * sizeof(float), nullptr, 0, out, n * 11 * sizeof(float), [&](void* a, void*, void* o) {
        hipLaunchKernelGGL(k_debug_bsdf_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->dscene, material, (const float*)a, (uint32_t)n, (float*)o); });
}

int rtx_debug_env_sample(rtx_ctx* c, const float* u01, size_t n, float* out)
{
    CTX_TRY(c);
    if (!c->have_scene) return fail(RTX_ERR_STATE, "no scene uploaded");
    if (c->dscene.bg_kind != 2) return fail(RTX_ERR_STATE, "the scene has no environment map");
    if (n == 0) return RTX_OK;
    if (!u01 || !out || n > (1u << 26)) return fail(RTX_ERR_INVALID, "bad argument");
    return debug_table(c, u01, n * 2 * sizeof(float), nullptr, 0, out, n * 7 * sizeof(float), [&](void* a, void*, void* o) {
        hipLaunchKernelGGL(k_debug_env_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->dscene, (const float*)a, (uint32_t)n, (float*)o); });
}

int rtx_debug_env_eval(rtx_ctx* c, const float* dirs, size_t n, float* out)
{
    CTX_TRY(c);
    if (!c->have_scene) return fail(RTX_ERR_STATE, "no scene uploaded");
    if (n == 0) return RTX_OK;
    if (!dirs || !out || n > (1u << 26)) return fail(RTX_ERR_INVALID, "bad argument");
    return debug_table(c, dirs, n * 3 * sizeof(float), nullptr, 0, out, n * 4 * sizeof(float), [&](void* a, void*, void* o) {
        hipLaunchKernelGGL(k_debug_env_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->dscene, (const float*)a, (uint32_t)n, (float*)o); });
}

// ---- introspection -----------------------------------------------------------------------

int rtx_get_bvh(rtx_ctx* c, uint32_t* n_nodes, float* bounds, uint32_t* topo, uint32_t* indices)
{
    if (!c) return fail(RTX_ERR_INVALID, "null context");
    if (!c->have_scene) return fail(RTX_ERR_STATE, "no scene uploaded");
    const Bvh2& b = c->host.bvh2;
    if (n_nodes) *n_nodes = (uint32_t)b.nodes.size();
    for (size_t i = 0; i < b.nodes.size(); i++) {
        const Bvh2Node& n = b.nodes[i];
        if (bounds) { for (int a = 0; a < 3; a++) { bounds[i * 6 + a] = n.bmin[a]; bounds[i * 6 + 3 + a] = n.bmax[a]; } }
        if (topo) { topo[i * 4] = n.start; topo[i * 4 + 1] = n.end; topo[i * 4 + 2] = n.child_l; topo[i * 4 + 3] = n.child_r; }
    }
    if (indices) std::memcpy(indices, b.indices.data(), b.indices.size() * sizeof(uint32_t));
    return RTX_OK;
}

struct rtx_host_bvh { HostScene hs; };

int rtx_host_bvh_build(const rtx_scene_desc* d, rtx_host_bvh** out)
{
    if (!d || !out) return fail(RTX_ERR_INVALID, "null argument");
    rtx_host_bvh* h = new (std::nothrow) rtx_host_bvh;
    if (!h) return fail(RTX_ERR_NOMEM, "out of host memory");
    std::string err = prepare_scene(*d, h->hs);
    if (!err.empty()) { delete h; return fail(RTX_ERR_INVALID, "scene rejected: %s", err.c_str()); }
    *out = h;
    return RTX_OK;
}

int rtx_host_bvh_get(const rtx_host_bvh* h, uint32_t* n_nodes, uint32_t* depth, float* bounds, uint32_t* topo,
                     uint32_t* indices, uint32_t* wide_nodes, uint32_t* wide_depth)
{
    if (!h) return fail(RTX_ERR_INVALID, "null handle");
    const Bvh2& b = h->hs.bvh2;
    if (n_nodes) *n_nodes = (uint32_t)b.nodes.size();
    if (depth) *depth = b.depth;
    if (wide_nodes) *wide_nodes = (uint32_t)h->hs.bvh4.nodes.size();
    if (wide_depth) *wide_depth = h->hs.bvh4.depth;
    for (size_t i = 0; i < b.nodes.size(); i++) {
        const Bvh2Node& n = b.nodes[i];
        if (bounds) { for (int a = 0; a < 3; a++) { bounds[i * 6 + a] = n.bmin[a]; bounds[i * 6 + 3 + a] = n.bmax[a]; } }
        if (topo) { topo[i * 4] = n.start; topo[i * 4 + 1] = n.end; topo[i * 4 + 2] = n.child_l; topo[i * 4 + 3] = n.child_r; }
    }
    if (indices) std::memcpy(indices, b.indices.data(), b.indices.size() * sizeof(uint32_t));
    return RTX_OK;
}

void rtx_host_bvh_free(rtx_host_bvh* h) { delete h; }

int rtx_get_stats(rtx_ctx* c, rtx_stats* out) { if (!c || !out) return fail(RTX_ERR_INVALID, "null argument"); *out = c->stats; return RTX_OK; }
int rtx_reset_stats(rtx_ctx* c) { if (!c) return fail(RTX_ERR_INVALID, "null context"); c->stats.closest_rays = c->stats.shadow_rays = c->stats.samples = 0; return RTX_OK; }

int rtx_timer_begin(rtx_ctx* c) { CTX_TRY(c); HIP_TRY(hipEventRecord(c->ev0, c->stream)); return RTX_OK; }
int rtx_timer_end(rtx_ctx* c, float* ms)
{
    CTX_TRY(c);
    if (!ms) return fail(RTX_ERR_INVALID, "null argument");
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return RTX_OK;
}

} // extern "C"
