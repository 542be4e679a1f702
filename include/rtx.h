/*
 * rtx.h — C ABI of the MI355X renderer core.
 *
 * The reference renderer has no FFI layer; its seam is the object boundary between
 * RayTracer (RTBase/Renderer.h) and Scene / Film (RTBase/Scene.h, RTBase/Imaging.h).
 * Every entry point below names the reference interface it stands in for.
 *
 * Conventions
 *   - every function returns 0 (RTX_OK) or a negative RTX_ERR_* code; rtx_last_error()
 *     gives the message of the last failure on the calling thread
 *   - handles are opaque; host pointers are borrowed for the duration of the call only
 *   - one context drives one GPU; one host thread per context (the reference's caller,
 *     RTBase/Main.cpp:129-221, is single threaded as well)
 *   - all arithmetic is fp32 as in the reference; matrices are row-major float[16]
 *     (RTBase/Core.h:217-224)
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_OK 0
#define RTX_ERR_INVALID (-1) /* bad argument / shape mismatch            */
#define RTX_ERR_HIP     (-2) /* a HIP runtime call failed                */
#define RTX_ERR_STATE   (-3) /* call order violated (e.g. no scene yet)  */
#define RTX_ERR_NOMEM   (-4) /* host or device allocation failed         */

typedef struct rtx_ctx rtx_ctx;

/* ---- plain data ------------------------------------------------------------------- */

/* Ray origin + direction. The reciprocal direction of RTBase/Geometry.h:25 is recomputed
 * on the device with the same three IEEE divides. */
typedef struct { float o[3]; float d[3]; } rtx_ray;

/* IntersectionData, RTBase/Geometry.h:276-283. On a miss t == FLT_MAX and
 * id == 0xFFFFFFFF (the reference leaves ID uninitialised there). */
typedef struct { uint32_t id; float t, alpha, beta, gamma; } rtx_hit;

/* ALGORITHM / DRAW_MODE, RTBase/Settings.h:3-16; IMAGE_FILTER, RTBase/Imaging.h:16-21 */
enum { RTX_AL_PATH_TRACE = 0, RTX_AL_LIGHT_TRACE = 1, RTX_AL_INSTANT_RADIOSITY = 2 };
enum { RTX_DM_ALGORITHM = 0, RTX_DM_NORMALS = 1, RTX_DM_ALBEDO = 2, RTX_DM_DIRECT = 3 };
enum { RTX_FT_BOX = 0, RTX_FT_GAUSSIAN = 1, RTX_FT_MITCHELL_NETRAVALI = 2 };

/* BSDF kinds, RTBase/Materials.h:139-678 (the material factory is
 * RTBase/SceneLoader.h:212-272). */
enum {
    RTX_BSDF_DIFFUSE   = 0, /* DiffuseBSDF   Materials.h:139 */
    RTX_BSDF_MIRROR    = 1, /* MirrorBSDF    Materials.h:179 */
    RTX_BSDF_CONDUCTOR = 2, /* ConductorBSDF Materials.h:234 ("Glossy") */
    RTX_BSDF_GLASS     = 3, /* GlassBSDF     Materials.h:351 */
    RTX_BSDF_PLASTIC   = 4, /* PlasticBSDF   Materials.h:564 */
    RTX_BSDF_DIELECTRIC = 5, /* DielectricBSDF Materials.h:436: cosine lobe, one-sided (the reference's placeholder body) */
    RTX_BSDF_ORENNAYAR = 6, /* OrenNayarBSDF Materials.h:485, sigma passed in `roughness` */
    /* LayeredBSDF Materials.h:680 forwards everything to its base BSDF but reports isTwoSided() = true:
     * describe it as (base type | RTX_BSDF_LAYERED) with the base's parameters */
    RTX_BSDF_LAYERED   = 0x100
};

/* One material. `roughness` is the constructor argument of the reference classes
 * (alpha = 1.62142*sqrt(roughness), Materials.h:248,582); emission > 0 luminance makes
 * every triangle that uses the material an area light (Scene.h:101-110). */
typedef struct {
    int32_t type;
    int32_t texture;     /* index into rtx_scene_desc.textures (albedo / reflectance) */
    float emission[3];
    float eta[3];        /* conductor */
    float k[3];          /* conductor */
    float roughness;     /* conductor, plastic */
    float int_ior;       /* glass, plastic */
    float ext_ior;       /* glass, plastic */
} rtx_material;

/* Texture texels as the reference keeps them after load: float RGB, row-major
 * (RTBase/Imaging.h:50-67). */
typedef struct { int32_t width, height; const float* rgb; } rtx_texture;

/* Scene hand-over: what Scene::init + Scene::build (RTBase/Scene.h:94-110,151-169) receive,
 * as structure-of-arrays. Triangle i has vertices positions[i*9 + 3*k .. +2], k = 0..2.
 * env_rgb == NULL selects BackgroundColour(background) (RTBase/Lights.h:86), otherwise an
 * EnvironmentMap (RTBase/Lights.h:307) over env_width x env_height float RGB texels. */
typedef struct {
    uint64_t n_triangles;
    const float* positions;          /* [n][3][3] */
    const float* normals;            /* [n][3][3] */
    const float* uvs;                /* [n][3][2] */
    const uint32_t* material_index;  /* [n]       */
    uint32_t n_materials;
    const rtx_material* materials;
    uint32_t n_textures;
    const rtx_texture* textures;
    int32_t env_width, env_height;
    const float* env_rgb;
    float background[3];
} rtx_scene_desc;

/* Camera state exactly as RTBase/Scene.h:10-41 holds it. */
typedef struct {
    float proj[16];            /* projectionMatrix                */
    float inv_proj[16];        /* inverseProjectionMatrix         */
    float camera[16];          /* camera (view-to-world)          */
    float camera_to_view[16];  /* cameraToView (world-to-view)    */
    float origin[3];
    float view_dir[3];
    float afilm;
    int32_t width, height;
} rtx_camera;

/* SETTINGS, RTBase/Settings.h:18-62, plus the counter-RNG seed and the multi-GPU shard.
 * shard_mode 0: this context renders everything; 1: 16x16 tiles t with
 * t % shard_world == shard_rank; 2: every pixel, and sample indices s with
 * s % shard_world == shard_rank (the caller sums films across ranks). */
typedef struct {
    int32_t algorithm, draw_mode, filter;
    int32_t use_mis, can_hit_light, adaptive_sampling;
    uint32_t max_bounces, init_spp, total_spp;
    uint32_t vpl_rays_per_tile, num_threads; /* VPL light paths per pass = product, Renderer.h:174 */
    uint32_t seed;
    int32_t shard_mode, shard_rank, shard_world;
} rtx_settings;

typedef struct {
    uint64_t closest_rays;     /* rays through the closest-hit kernel since create/reset */
    uint64_t shadow_rays;      /* rays through the visibility kernel                      */
    uint64_t samples;          /* pixel samples started                                   */
    uint32_t bvh2_nodes, bvh2_depth, bvh4_nodes, bvh4_depth, n_lights;
    float build_ms;            /* BVH build (device, or host under RTX_BVH_BUILD=host) + flatten */
} rtx_stats;

/* ---- context ---------------------------------------------------------------------- */

/* No reference counterpart (the reference is CPU only): binds a context to GPU `device`
 * and creates its stream. */
int rtx_create(rtx_ctx** out, int device);
void rtx_destroy(rtx_ctx* ctx);
const char* rtx_last_error(void);

/* Work is enqueued on `hip_stream` (a hipStream_t; NULL is HIP's null stream, as everywhere in HIP)
 * instead of the context's own stream, which is created non-blocking and is therefore NOT ordered
 * against the null stream; RTX_STREAM_OWN restores it. For callers that time or order work
 * themselves. */
#define RTX_STREAM_OWN ((void*)(intptr_t)-1)
int rtx_set_stream(rtx_ctx* ctx, void* hip_stream);
int rtx_synchronize(rtx_ctx* ctx);

/* ---- scene / camera / settings ---------------------------------------------------- */

/* Scene::init + Scene::build, RTBase/Scene.h:94-110,151-169 and BVHTree::build,
 * RTBase/Geometry.h:458-504: builds the reference's binary SAH tree on the host
 * (identical topology and leaf order), flattens it into the 4-wide device layout and
 * uploads triangles, materials, textures, lights and the environment tables
 * (TabulatedDistribution::init, RTBase/Lights.h:161-208). Triangles of zero area must
 * already be dropped (RTBase/SceneLoader.h:330). */
int rtx_upload_scene(rtx_ctx* ctx, const rtx_scene_desc* scene);

/* loadScene's camera set-up, RTBase/SceneLoader.h:342-360 + Camera::init/updateView,
 * RTBase/Scene.h:22-41: perspective(0.001, 10000, w/h, fov), lookAt(from,to,up).invert(). */
int rtx_camera_look_at(rtx_camera* out, const float from[3], const float to[3], const float up[3],
                       float fov_deg, int width, int height, int flip_x);
int rtx_set_camera(rtx_ctx* ctx, const rtx_camera* cam);

/* The vertex transform of loadInstance, RTBase/SceneLoader.h:298-316, in place on n vertices
 * (positions and normals are n x 3 floats): p = world.mulPoint(p), normal =
 * world.invert().transpose().mulVec(normal).normalize(). `world` is the 16 floats of the
 * instance's "world" entry in scene.json, row-major. Host only, touches no GPU. */
int rtx_host_transform_instance(const float world[16], size_t n, float* positions, float* normals);

/* RayTracer::init's settings hand-over, RTBase/Renderer.h:69-87 (adaptive sampling is
 * gated to path tracing there; initSPP falls back to totalSPP without it). */
int rtx_set_settings(rtx_ctx* ctx, const rtx_settings* s);

/* ---- progressive rendering -------------------------------------------------------- */

/* RayTracer::clear, RTBase/Renderer.h:117 → Film::clear, RTBase/Imaging.h:397. */
int rtx_clear(rtx_ctx* ctx);

/* n calls of RayTracer::renderMT, RTBase/Renderer.h:753-782: each pass adds one sample
 * per active pixel (or one light path per pixel for light tracing), and after the pass
 * in which SPP == initSPP computes the adaptive per-tile sample targets
 * (calculateTileSamples, RTBase/Renderer.h:643-671). Blocking. */
int rtx_render_passes(rtx_ctx* ctx, int n_passes);

/* RayTracer::getSPP, RTBase/Renderer.h:856. */
int rtx_get_spp(rtx_ctx* ctx, int* spp);

/* Film::film / Film::SPP / RayTracer::tileSamples as the host reads them
 * (RTBase/Imaging.h:299-302, RTBase/Renderer.h:41): accumulated RGB sums [H][W][3].
 * tile_samples may be NULL; otherwise it receives ceil(W/16)*ceil(H/16) entries. */
int rtx_film_read(rtx_ctx* ctx, float* rgb_sums, int* spp, uint32_t* tile_samples);

/* Restore an accumulated film and its sample count (resume of a progressive render, or a
 * film prepared by the caller); the inverse of rtx_film_read. */
int rtx_film_write(rtx_ctx* ctx, const float* rgb_sums, int spp);

/* Device address of the film accumulator (W*H*3 floats) and of the tile-sample table, for
 * callers that combine films across GPUs with a collective (no reference counterpart). */
int rtx_film_device(rtx_ctx* ctx, void** film_dev, size_t* n_floats);
int rtx_tile_samples_device(rtx_ctx* ctx, void** tiles_dev, size_t* n_tiles);
/* Recompute the adaptive tile targets from the film as it stands (after a cross-GPU sum),
 * calculateTileSamples, RTBase/Renderer.h:643-671. */
int rtx_update_tile_samples(rtx_ctx* ctx);
/* Overwrite SPP (after a cross-GPU sum of films rendered with shard_mode 2). */
int rtx_set_spp(rtx_ctx* ctx, int spp);

/* ---- one render over several GPUs (no reference counterpart: the reference's only parallelism is the std::thread
 * tile queue of RTBase/Renderer.h:677-782, which is what the partitioning follows) ------------------------------------
 *
 * One process and one context per GPU, the scene uploaded to each; rtx_settings.shard_mode / shard_rank / shard_world
 * say which tiles (1) or which samples (2) the context renders. The contexts meet through an rtx_collective: two
 * in-place all-reduce (sum) functions over device memory, which either enqueue on `hip_stream` or are complete when
 * they return; they return 0 or a negative RTX_ERR_* code. */
typedef struct {
    void* user;
    int (*allreduce_sum_f32)(void* user, float* dev, size_t n, void* hip_stream);
    int (*allreduce_sum_u32)(void* user, uint32_t* dev, size_t n, void* hip_stream);
} rtx_collective;

/* RCCL (the collectives library of ROCm; xGMI between the GPUs of a node) behind an rtx_collective. The library is
 * loaded when one of these is first called; RTX_ERR_STATE if it cannot be found. rtx_rccl_unique_id: ncclGetUniqueId, to
 * be called on one rank and handed to the others by the host (pipe, file, MPI ...); rtx_rccl_comm_create:
 * ncclCommInitRank on the context's GPU, collective over all ranks; rtx_collective_rccl fills `out` for that
 * communicator (user = the ncclComm_t, which a host that links RCCL itself may also pass directly). */
int rtx_rccl_unique_id(char id[128]);
int rtx_rccl_comm_create(rtx_ctx* ctx, void** nccl_comm, int world, int rank, const char id[128]);
int rtx_rccl_comm_destroy(void* nccl_comm);
int rtx_collective_rccl(rtx_collective* out, void* nccl_comm);

/* Sum the film accumulators of all ranks in place; blocking. */
int rtx_film_allreduce(rtx_ctx* ctx, const rtx_collective* coll);

/* settings.total_spp GLOBAL samples per pixel, this context's share of them according to its shard settings, and
 * the complete film with SPP = total_spp on every rank when it returns. With adaptive sampling the ranks meet once
 * more, when the global sample count reaches initSPP: tile split - every rank computes the targets of its own tiles
 * (complete pixels) and the tables are summed; sample split - the films are summed, every rank computes the same
 * targets from the sum (calculateTileSamples, RTBase/Renderer.h:643-671) and carries on with its own partial film.
 * (a tile split with a Gaussian or Mitchell filter, whose splats cross tile borders, does the same). The film must be
 * empty when the call starts (SPP == 0, i.e. after rtx_clear or rtx_set_camera): RTX_ERR_STATE otherwise.
 * shard_world 1 needs no collective (coll may be NULL) and equals rtx_render_passes(total_spp). Blocking. */
int rtx_render_sharded(rtx_ctx* ctx, const rtx_collective* coll);

/* RayTracer::createAOV, RTBase/Renderer.h:786-820: the three input planes of the denoiser, W*H*3
 * floats each, any pointer may be NULL. color = film / spp (per-tile spp once adaptive sampling
 * has taken over, :798-802); albedo = RayTracer::albedo (:553-565) and normal =
 * RayTracer::viewNormals (:567-575) through the pixel centres. Blocking; the _device variant takes
 * device pointers and only enqueues. */
int rtx_create_aov(rtx_ctx* ctx, float* color, float* albedo, float* normal);
int rtx_create_aov_device(rtx_ctx* ctx, float* color, float* albedo, float* normal);

/* ---- ray queries ------------------------------------------------------------------ */

/* Scene::traverse, RTBase/Scene.h:137-141 → BVHTree::traverse, RTBase/Geometry.h:506-566,
 * for n rays in host memory. */
int rtx_trace_closest(rtx_ctx* ctx, const rtx_ray* rays, size_t n, rtx_hit* hits);
/* Same, rays and hits already in device memory (no copies, asynchronous on the stream). */
int rtx_trace_closest_device(rtx_ctx* ctx, const rtx_ray* rays_dev, size_t n, rtx_hit* hits_dev);

/* No counterpart in the reference: rtx_trace_closest_device for callers that cannot allocate device
 * memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc). Same
 * kernel, same results; the work counters and the global part of the traversal stacks live in a
 * static device array of ~60 MiB that every GPU the library runs on holds until the process ends,
 * instead of in buffers that grow on demand. It calls no allocating routine, and fails with
 * RTX_ERR_INVALID beyond its fixed limits: n > RTX_NOALLOC_MAX_RAYS, or a scene whose tree can need
 * more than RTX_NOALLOC_MAX_STACK traversal-stack entries per ray (max(3 * wide depth + 2, binary
 * depth + 3); 32 for the 1 M-triangle soup, 38 for the mixed-material scene). All calls on one GPU
 * share the array: calls that overlap on one GPU (two streams or two contexts) race. */
#define RTX_NOALLOC_MAX_RAYS  0xF0000000ull
#define RTX_NOALLOC_MAX_STACK 42
int rtx_trace_closest_device_noalloc(rtx_ctx* ctx, const rtx_ray* rays_dev, size_t n, rtx_hit* hits_dev);

/* Scene::visible, RTBase/Scene.h:170-178 → BVHTree::traverseVisible,
 * RTBase/Geometry.h:568-610, for n segments p1→p2 (6 floats each); visible[i] is 1 or 0. */
int rtx_trace_visible(rtx_ctx* ctx, const float* p1p2, size_t n, uint8_t* visible);
int rtx_trace_visible_device(rtx_ctx* ctx, const float* p1p2_dev, size_t n, uint8_t* visible_dev);

/* Camera::generateRay, RTBase/Scene.h:49-59, at the pixel centres (x+0.5, y+0.5) of the
 * whole film, written to device memory in the core's 8x8-tile queue order; pixel_dev[i]
 * (may be NULL) receives y*W+x of ray i. n_capacity must be >= W*H. */
int rtx_generate_primary_rays_device(rtx_ctx* ctx, rtx_ray* rays_dev, uint32_t* pixel_dev,
                                     size_t n_capacity);

/* ---- function-level parity tables (test support; host buffers, blocking) ------------ */

/* Scene::calculateShadingData, RTBase/Scene.h:183-213, for n (ray, hit) pairs:
 * out[i] = x(3) wo(3) sNormal(3) gNormal(3) tu tv frame.u(3) frame.v(3) frame.w(3) material index (24 floats; zeros on a miss). */
int rtx_debug_shading_data(rtx_ctx* ctx, const rtx_ray* rays, const rtx_hit* hits, size_t n, float* out);

/* BSDF::sample / evaluate / PDF, RTBase/Materials.h:139-724, of material `material` on a table of inputs:
 * in[i] = normal(3) wo(3) tu tv wi_eval(3) u0 u1 u2 (14 floats; u* are what Sampler::next() returns, in call order,
 * cyclic); out[i] = sampled wi(3) colour(3) pdf, evaluate(wi_eval)(3), PDF(wi_eval) (11 floats). */
int rtx_debug_bsdf_table(rtx_ctx* ctx, uint32_t material, const float* in, size_t n, float* out);

/* EnvironmentMap::sample, RTBase/Lights.h:252-277,338, on n (u0, u1) pairs: out[i] = wi(3) pdf Le(3); needs an
 * environment map. Light::evaluate / PDF of the background, RTBase/Lights.h:101-108,344-366, on n directions:
 * out[i] = Le(3) pdf. */
int rtx_debug_env_sample(rtx_ctx* ctx, const float* u01, size_t n, float* out);
int rtx_debug_env_eval(rtx_ctx* ctx, const float* dirs, size_t n, float* out);

/* ---- introspection ---------------------------------------------------------------- */

/* The binary tree exactly as BVHTree keeps it (RTBase/Geometry.h:293-304): per node
 * bounds_min[3], bounds_max[3], start, end, child_l, child_r; and the index permutation.
 * Pass NULL arrays to query the counts only. */
int rtx_get_bvh(rtx_ctx* ctx, uint32_t* n_nodes, float* bounds /*[n][6]*/,
                uint32_t* topo /*[n][4]*/, uint32_t* indices /*[n_triangles]*/);

/* The same tree without a GPU: runs only the host-side build of rtx_upload_scene
 * (BVHTree::build, RTBase/Geometry.h:458-504) and keeps the result in an opaque handle.
 * wide_nodes / wide_depth describe the flattened 4-wide layout. */
typedef struct rtx_host_bvh rtx_host_bvh;
int rtx_host_bvh_build(const rtx_scene_desc* scene, rtx_host_bvh** out);
int rtx_host_bvh_get(const rtx_host_bvh* h, uint32_t* n_nodes, uint32_t* depth, float* bounds /*[n][6]*/,
                     uint32_t* topo /*[n][4]*/, uint32_t* indices, uint32_t* wide_nodes, uint32_t* wide_depth);
void rtx_host_bvh_free(rtx_host_bvh* h);

int rtx_get_stats(rtx_ctx* ctx, rtx_stats* out);
int rtx_reset_stats(rtx_ctx* ctx);

/* hipEvent pair on the context's current stream: elapsed GPU time of the work enqueued
 * between the two calls, in milliseconds. */
int rtx_timer_begin(rtx_ctx* ctx);
int rtx_timer_end(rtx_ctx* ctx, float* ms);

/* Algorithmic traffic of the closest-hit kernel for the rays of the last
 * rtx_trace_closest* call, counted by an instrumented device pass over the same layout:
 * wide nodes visited (128 B each, seven 16-byte loads per lane), triangles tested (36 B each) and leaf fetches
 * (up to four triangles = nine 16-byte loads per lane; tri_groups may be NULL), summed over rays. */
int rtx_count_closest_work(rtx_ctx* ctx, const rtx_ray* rays_dev, size_t n,
                           uint64_t* nodes_visited, uint64_t* tris_tested, uint64_t* tri_groups);

/* The same for the renderer: n_passes passes exactly like rtx_render_passes renders them (the film advances, results
 * are identical), one at a time, with the traversal kernels instrumented. out[0] = closest-hit rays (live paths
 * summed over depths), out[1] = shadow / camera-connection segments, out[2] = wide nodes visited and out[3] = triangles
 * tested by all of them, out[4] = pixel samples (light paths) started, out[5] = leaf fetches. Instant radiosity:
 * out[6] = the pixel-to-VPL segments the gather walked (those with a positive geometry term, RTBase/Renderer.h:229-240;
 * they are part of out[1] too), out[7] = valid VPLs summed over the passes (vpls.size(), RTBase/Renderer.h:244). */
int rtx_count_render_work(rtx_ctx* ctx, int n_passes, uint64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
