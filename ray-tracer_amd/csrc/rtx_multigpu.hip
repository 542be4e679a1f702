// rtx_multigpu.hip — one render partitioned over several GPUs, behind the C ABI.
//
// The reference renders on one device (its only parallelism is the std::thread tile queue of
// RTBase/Renderer.h:677-782). Here every GPU has a context in a process of its own with the scene replicated;
// rtx_settings.shard_* says which part of the work the context renders, and the contexts meet in at most two
// places: when the GLOBAL sample count reaches initSPP (adaptive sampling needs the complete film of the first
// initSPP samples, calculateTileSamples, RTBase/Renderer.h:643-671, :780-781) and when the render is done (the
// films are summed). Those meetings go through an rtx_collective, a pair of in-place all-reduce functions;
// rtx_collective_rccl() fills one in over an RCCL communicator, loaded at run time so that single-GPU users
// need no RCCL at all.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>

#include "context.h"

using namespace rtx;

namespace {

// ---- RCCL, loaded on first use ---------------------------------------------------------------------------
struct UniqueId { char bytes[128]; };                  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclSum = 0, kNcclUint32 = 3, kNcclFloat32 = 7;   // ncclRedOp_t / ncclDataType_t values of rccl.h

Rccl g_rccl;
std::mutex g_rccl_mutex;

int load_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return RTX_OK;
    void* h = nullptr;
    // a library of that name the process has already mapped (a host application linked against RCCL, PyTorch's copy)
    // is the one to use: two RCCL instances in one process would not see each other's communicators
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return fail(RTX_ERR_STATE, "RCCL is not available: %s", dlerror());
    Rccl r; r.lib = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) return fail(RTX_ERR_STATE, "RCCL library lacks an expected symbol");
    g_rccl = r;
    return RTX_OK;
}
#define NCCL_TRY(expr) do { int r_ = (expr); if (r_ != 0) return fail(RTX_ERR_HIP, "%s: %s", #expr, g_rccl.GetErrorString(r_)); } while (0)

int rccl_sum_f32(void* comm, float* dev, size_t n, void* stream) { NCCL_TRY(g_rccl.AllReduce(dev, dev, n, kNcclFloat32, kNcclSum, comm, (hipStream_t)stream)); return RTX_OK; }
int rccl_sum_u32(void* comm, uint32_t* dev, size_t n, void* stream) { NCCL_TRY(g_rccl.AllReduce(dev, dev, n, kNcclUint32, kNcclSum, comm, (hipStream_t)stream)); return RTX_OK; }

// tile targets of tiles this rank does not own are meaningless (their pixels are empty here): zero them before the sum
__global__ void k_keep_own_tiles(uint32_t* tiles, uint32_t n, uint32_t rank, uint32_t world)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && i % world != rank) tiles[i] = 0u;
}

uint32_t passes_of_rank(uint32_t spp, const rtx_settings& s)
{
    if (s.shard_mode != 2 || s.shard_world <= 1) return spp;
    return spp / (uint32_t)s.shard_world + ((uint32_t)s.shard_rank < spp % (uint32_t)s.shard_world ? 1u : 0u);
}

} // namespace

extern "C" {

int rtx_rccl_unique_id(char id[128])
{
    if (!id) return fail(RTX_ERR_INVALID, "null argument");
    int rc = load_rccl(); if (rc) return rc;
    UniqueId u; NCCL_TRY(g_rccl.GetUniqueId(&u));
    std::memcpy(id, u.bytes, sizeof u.bytes);
    return RTX_OK;
}

int rtx_rccl_comm_create(rtx_ctx* c, void** comm, int world, int rank, const char id[128])
{
    CTX_TRY(c);                                  // the communicator binds to the context's GPU
    if (!comm || !id || world < 1 || rank < 0 || rank >= world) return fail(RTX_ERR_INVALID, "bad argument");
    int rc = load_rccl(); if (rc) return rc;
    UniqueId u; std::memcpy(u.bytes, id, sizeof u.bytes);
    NCCL_TRY(g_rccl.CommInitRank(comm, world, u, rank));
    return RTX_OK;
}

int rtx_rccl_comm_destroy(void* comm)
{
    if (!comm) return RTX_OK;
    int rc = load_rccl(); if (rc) return rc;
    NCCL_TRY(g_rccl.CommDestroy(comm));
    return RTX_OK;
}

int rtx_collective_rccl(rtx_collective* out, void* comm)
{
    if (!out || !comm) return fail(RTX_ERR_INVALID, "null argument");
    int rc = load_rccl(); if (rc) return rc;
    out->user = comm; out->allreduce_sum_f32 = rccl_sum_f32; out->allreduce_sum_u32 = rccl_sum_u32;
    return RTX_OK;
}

int rtx_film_allreduce(rtx_ctx* c, const rtx_collective* coll)
{
    CTX_TRY(c);
    if (!coll || !coll->allreduce_sum_f32) return fail(RTX_ERR_INVALID, "null collective");
    if (!c->have_camera) return fail(RTX_ERR_STATE, "no camera set");
    int rc = coll->allreduce_sum_f32(coll->user, c->b_film.as<float>(), (size_t)c->cam.width * c->cam.height * 3, c->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTX_OK;
}

int rtx_render_sharded(rtx_ctx* c, const rtx_collective* coll)
{
    CTX_TRY(c);
    if (!c->have_scene || !c->have_camera) return fail(RTX_ERR_STATE, "scene and camera must be set before rendering");
    // The passes below are counted from zero: with samples already in the film the adaptive meeting would come at the
    // wrong pass (or never) and the film would hold more samples than SPP says afterwards.
    if (c->spp != 0) return fail(RTX_ERR_STATE, "rtx_render_sharded starts from an empty film: SPP is %d, call rtx_clear first", c->spp);
    const rtx_settings st = c->settings;                 // adaptive_sampling / init_spp already gated as RayTracer::init does
    const uint32_t world = (uint32_t)std::max(1, st.shard_world);
    if (world > 1 && st.shard_mode == 0) return fail(RTX_ERR_STATE, "shard_world %u with shard_mode 0: set shard_mode 1 (tiles) or 2 (samples)", world);
    if (world > 1 && (!coll || !coll->allreduce_sum_f32 || !coll->allreduce_sum_u32)) return fail(RTX_ERR_INVALID, "a collective is needed for %u ranks", world);
    const size_t n_film = (size_t)c->cam.width * c->cam.height * 3, n_tiles = (size_t)c->tiles_x * c->tiles_y;
    const uint32_t total = st.total_spp, n_local = passes_of_rank(total, st);
    const bool adaptive = st.adaptive_sampling && st.draw_mode == RTX_DM_ALGORITHM && st.init_spp < total;
    int rc;
    if (world > 1 && adaptive) {
        const uint32_t n_init = passes_of_rank(st.init_spp, st);
        if ((rc = rtx_render_passes(c, (int)n_init))) return rc;
        if (st.shard_mode == 1 && st.filter == RTX_FT_BOX) {
            // every rank has computed targets for all tiles from its own film; only those of its own tiles count
            // (box filter: a tile's pixels are complete on its owner but for the few samples whose x + r rounds up
            // across a tile border, which can move a target by one sample)
            hipLaunchKernelGGL(k_keep_own_tiles, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, c->stream, c->b_tile_samples.as<uint32_t>(), (uint32_t)n_tiles,
                               (uint32_t)st.shard_rank, world);
            HIP_TRY(hipGetLastError());
            if ((rc = coll->allreduce_sum_u32(coll->user, c->b_tile_samples.as<uint32_t>(), n_tiles, c->stream))) return rc;
        } else {
            // the variance is not linear in the per-rank sums: sum the films of the first initSPP samples, compute the
            // targets from the sum on every rank (identical input, identical targets), put the own partial film back.
            // Tile split with a Gaussian or Mitchell filter takes this way too: their splats reach two pixels into
            // the neighbouring tiles (RTBase/Imaging.h:324-339), so no tile is complete on its owner.
            HIP_TRY(c->b_io_a.ensure(n_film * sizeof(float)));
            HIP_TRY(hipMemcpyAsync(c->b_io_a.p, c->b_film.p, n_film * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            if ((rc = coll->allreduce_sum_f32(coll->user, c->b_film.as<float>(), n_film, c->stream))) return rc;
            const int keep = c->spp;
            c->spp = (int)st.init_spp;                   // the ranks' passes add up to exactly initSPP global samples
            if ((rc = rtx_update_tile_samples(c))) return rc;
            c->spp = keep;
            HIP_TRY(hipMemcpyAsync(c->b_film.p, c->b_io_a.p, n_film * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        if ((rc = rtx_render_passes(c, (int)(n_local - n_init)))) return rc;
    } else if ((rc = rtx_render_passes(c, (int)n_local))) return rc;
    if (world > 1) {
        if ((rc = coll->allreduce_sum_f32(coll->user, c->b_film.as<float>(), n_film, c->stream))) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->spp = (int)total;
    }
    return RTX_OK;
}

} // extern "C"
