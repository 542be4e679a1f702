// context.h — the renderer context behind the opaque rtx_ctx handle.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "device_scene.h"
#include "scene_host.h"

namespace rtx {

constexpr int kBlock = 256;        // threads per workgroup (4 waves)
constexpr int kLdsStack = 12;      // traversal-stack entries per lane kept in LDS
constexpr int kMinLdsStack = 0;    // smallest LDS part of any kernel (sizes the global spill slab): the exact path of the packet kernels keeps its whole stack in global memory
constexpr int kMaxGridBlocks = 2048;   // 256 CUs x 8 resident workgroups
constexpr int kPipeline = 4;       // render passes in flight at most (one stream and one set of path state each)

// Device buffer that frees itself.
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    hipError_t ensure(size_t n) { if (n <= bytes) return hipSuccess; release(); hipError_t e = hipMalloc(&p, n); if (e == hipSuccess) bytes = n; return e; }
    template <class T> T* as() const { return (T*)p; }
};

// Pinned host memory for the host-pointer entry points: two slots per direction, so that the host's copy into
// (out of) one slot runs beside the DMA of the other.
struct PinBuf {
    static constexpr size_t kSlot = 8u << 20;
    char* p = nullptr; hipEvent_t ev[2] = {nullptr, nullptr};
    ~PinBuf() { if (p) (void)hipHostFree(p); for (auto& e : ev) if (e) (void)hipEventDestroy(e); }
    hipError_t ensure()
    {
        if (p) return hipSuccess;
        hipError_t e = hipHostMalloc((void**)&p, 2 * kSlot, hipHostMallocDefault);
        for (int i = 0; e == hipSuccess && i < 2; i++) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
        return e;
    }
};

struct WavefrontBuffers;   // path-tracing state (wavefront.h)

// records the message rtx_last_error() returns on the calling thread and hands `code` back
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

} // namespace rtx

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return rtx::fail(RTX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
#define CTX_TRY(ctx) do { if (!(ctx)) return rtx::fail(RTX_ERR_INVALID, "null context"); HIP_TRY(hipSetDevice((ctx)->device)); } while (0)

struct rtx_ctx {
    int device = 0;
    int variant = 1;               // traversal schedule (RTX_VARIANT): 1 = default, 0 = static baseline, >= 2 = schedules under test
    int resident_blocks = rtx::kMaxGridBlocks;   // traversal workgroups the device holds at once
    int resident_packet_blocks = rtx::kMaxGridBlocks;   // the same for the packet kernels (traverse_packet.h)
    int primary_packet = 0;        // (tuning build, RTX_PRIMARY_PACKET=1) camera rays of a render pass: packets of 64 (k_closest_packet), the rest per lane
    int query_packet = 0;          // (tuning build, RTX_QUERY_PACKET=1) rtx_trace_closest_device: the same for coherent runs of the caller's rays
    int ir_packet = 1;             // instant-radiosity gather: one traversal per wave (RTX_IR_PACKET=0: one per lane)
    int rays_per_block = 16;       // grid of a ray-query launch too small to fill the device: one workgroup per this many rays
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    bool have_scene = false, have_camera = false;
    rtx::HostScene host;
    rtx::DeviceScene dscene{};
    rtx::CameraDev dcam{};
    rtx_camera cam{};
    rtx_settings settings{};

    rtx::DevBuf b_nodes4, b_nodes2, b_tris, b_tri_ids, b_shade, b_materials, b_texels, b_lights, b_env_texels, b_env_rows, b_env_cols, b_env_guide_rows, b_env_guide_cols;
    rtx::DevBuf b_spill;           // global part of the traversal stacks
    uint32_t spill_entries = 0;    // entries per lane in b_spill
    uint32_t stack_need = 0;       // traversal-stack entries a ray of the current scene can need at most
    void* na_scratch = nullptr;    // this device's copy of the static scratch of the no-allocation entry point
    rtx::DevBuf b_rest;            // ray queries: items the packet kernel left to the per-lane kernel
    rtx::DevBuf b_io_a, b_io_b;    // staging for the host-pointer entry points
    rtx::PinBuf pin_in, pin_out;
    hipEvent_t io_kernel_done[2] = {nullptr, nullptr};   // host-buffer ray queries: the kernel of a chunk has finished
    rtx::DevBuf b_counters;

    // progressive state
    rtx::DevBuf b_film, b_tile_samples;
    int spp = 0;
    uint32_t tiles_x = 0, tiles_y = 0;
    rtx::WavefrontBuffers* wf[rtx::kPipeline] = {};   // path state of the passes in flight; [0] always exists once a camera is set
    int pipeline = 0;                                  // passes rtx_render_passes keeps in flight: 0 = three, four for a small film or share of one (RTX_PIPELINE fixes it); a process has 4 hardware queues
    uint32_t tail_from_depth = 1;                      // path tracing: not before this depth one launch carries every path to its end (k_pt_tail); 0: never (RTX_TAIL_FROM)
    uint32_t* depth_stats = nullptr;                   // pinned host memory the path tracer's last kernel reports its queue sizes per depth to (see publish_depth_stats)
    bool ir_block_cull = true;                         // instant radiosity: (block of 256 gather points, VPL) pairs are skipped whole where their bounds allow it (RTX_IR_BLOCK_CULL)
    bool lt_regroup = true;                            // light tracing: shade items sorted by class inside a workgroup (RTX_LT_REGROUP)
    int tail_grid_limit = 0;                           // (tuning build, RTX_TAIL_GRID) at most this many workgroups for the tail kernel, 0 = as many as fit
    int tail_rounds = 0;                               // blocks of 64 queue positions a wave of the tail kernel takes at a time: 1, 2, or 0 = by the passes in flight (RTX_TAIL_ROUNDS)
    bool tail_stats = false;                           // (tuning build, RTX_DEBUG_COUNTERS) the tail kernel times its phases, the packet gather counts its steps
    int resident_tail_blocks[2] = {256, 256};          // workgroups of that kernel the device holds at once, by blocks per wave (1, 2)
    uint32_t regroup_from_depth = 1;                   // path tracing: the shade kernels from this depth on sort their items by class inside the workgroup (RTX_REGROUP_FROM)
    int passes_in_flight = 1;                          // set by rtx_render_passes for the passes it is enqueueing
    int spill_lanes = 1;                               // traversal-stack slabs in b_spill
    hipStream_t pipe_stream[rtx::kPipeline] = {};
    hipEvent_t pipe_film[rtx::kPipeline] = {}, pipe_done[rtx::kPipeline] = {}, ev_fork = nullptr, ev_tiles = nullptr;

    bool count_work = false;       // render passes run the instrumented traversal kernels (rtx_count_render_work)
    rtx_stats stats{};
};
