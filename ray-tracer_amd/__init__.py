"""ctypes binding of the MI355X renderer core (``librtx.so``, C ABI in ``include/rtx.h``).

The package directory name contains a hyphen, so import it with
``importlib.import_module("ray-tracer_amd")`` (``tests/conftest.py`` does).

There is no CPU fallback: if ``librtx.so`` has not been built (``__graft_entry__.build()``)
importing :class:`Core` raises, and every entry point that needs a GPU fails with the
library's own error message when none is visible.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtx.so")
LIB_TUNING_PATH = os.path.join(_HERE, "librtx_tuning.so")

AL_PATH_TRACE, AL_LIGHT_TRACE, AL_INSTANT_RADIOSITY = 0, 1, 2
DM_ALGORITHM, DM_NORMALS, DM_ALBEDO, DM_DIRECT = 0, 1, 2, 3
FT_BOX, FT_GAUSSIAN, FT_MITCHELL_NETRAVALI = 0, 1, 2
STREAM_OWN = C.c_void_p(-1).value      # RTX_STREAM_OWN of include/rtx.h: rtx_set_stream back to the context's own stream
BSDF_DIFFUSE, BSDF_MIRROR, BSDF_CONDUCTOR, BSDF_GLASS, BSDF_PLASTIC, BSDF_DIELECTRIC, BSDF_ORENNAYAR = 0, 1, 2, 3, 4, 5, 6
BSDF_LAYERED = 0x100   # OR-ed onto the base type: LayeredBSDF forwards to its base but is two-sided

RAY_DTYPE = np.dtype([("o", np.float32, 3), ("d", np.float32, 3)])
HIT_DTYPE = np.dtype([("id", np.uint32), ("t", np.float32), ("alpha", np.float32),
                      ("beta", np.float32), ("gamma", np.float32)])


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("texture", C.c_int32), ("emission", C.c_float * 3),
                ("eta", C.c_float * 3), ("k", C.c_float * 3), ("roughness", C.c_float),
                ("int_ior", C.c_float), ("ext_ior", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_float))]


class SceneDesc(C.Structure):
    _fields_ = [("n_triangles", C.c_uint64), ("positions", C.POINTER(C.c_float)),
                ("normals", C.POINTER(C.c_float)), ("uvs", C.POINTER(C.c_float)),
                ("material_index", C.POINTER(C.c_uint32)), ("n_materials", C.c_uint32),
                ("materials", C.POINTER(Material)), ("n_textures", C.c_uint32),
                ("textures", C.POINTER(Texture)), ("env_width", C.c_int32), ("env_height", C.c_int32),
                ("env_rgb", C.POINTER(C.c_float)), ("background", C.c_float * 3)]


class Camera(C.Structure):
    _fields_ = [("proj", C.c_float * 16), ("inv_proj", C.c_float * 16), ("camera", C.c_float * 16),
                ("camera_to_view", C.c_float * 16), ("origin", C.c_float * 3), ("view_dir", C.c_float * 3),
                ("afilm", C.c_float), ("width", C.c_int32), ("height", C.c_int32)]


class Settings(C.Structure):
    _fields_ = [("algorithm", C.c_int32), ("draw_mode", C.c_int32), ("filter", C.c_int32),
                ("use_mis", C.c_int32), ("can_hit_light", C.c_int32), ("adaptive_sampling", C.c_int32),
                ("max_bounces", C.c_uint32), ("init_spp", C.c_uint32), ("total_spp", C.c_uint32),
                ("vpl_rays_per_tile", C.c_uint32), ("num_threads", C.c_uint32), ("seed", C.c_uint32),
                ("shard_mode", C.c_int32), ("shard_rank", C.c_int32), ("shard_world", C.c_int32)]


_ALLREDUCE_F32 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
_ALLREDUCE_U32 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Collective(C.Structure):
    """rtx_collective: two in-place all-reduce (sum) callbacks over device memory."""
    _fields_ = [("user", C.c_void_p), ("allreduce_sum_f32", _ALLREDUCE_F32), ("allreduce_sum_u32", _ALLREDUCE_U32)]


class Stats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("samples", C.c_uint64),
                ("bvh2_nodes", C.c_uint32), ("bvh2_depth", C.c_uint32), ("bvh4_nodes", C.c_uint32),
                ("bvh4_depth", C.c_uint32), ("n_lights", C.c_uint32), ("build_ms", C.c_float)]


def make_settings(**kw) -> Settings:
    """SETTINGS with the defaults of createSettings() in RTBase/Main.cpp:11-36 (minus the
    interactive / denoise fields), overridable by keyword."""
    s = Settings(algorithm=AL_PATH_TRACE, draw_mode=DM_ALGORITHM, filter=FT_BOX, use_mis=1, can_hit_light=1,
                 adaptive_sampling=0, max_bounces=5, init_spp=10, total_spp=64, vpl_rays_per_tile=1,
                 num_threads=20, seed=1, shard_mode=0, shard_rank=0, shard_world=1)
    for k, v in kw.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class SceneArrays:
    """Owns the numpy arrays behind a SceneDesc so that the pointers stay valid."""

    def __init__(self, positions, normals, uvs, material_index, materials, textures, env=None,
                 background=(0.0, 0.0, 0.0)):
        self.positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3, 3)
        n = self.positions.shape[0]
        self.normals = np.ascontiguousarray(normals, np.float32).reshape(n, 3, 3)
        self.uvs = np.ascontiguousarray(uvs, np.float32).reshape(n, 3, 2)
        self.material_index = np.ascontiguousarray(material_index, np.uint32).reshape(n)
        self.textures = [np.ascontiguousarray(t, np.float32) for t in textures]
        for t in self.textures:
            assert t.ndim == 3 and t.shape[2] == 3
        self.materials = list(materials)
        self.env = None if env is None else np.ascontiguousarray(env, np.float32)
        self._mat = (Material * len(self.materials))(*self.materials)
        self._tex = (Texture * len(self.textures))()
        for i, t in enumerate(self.textures):
            self._tex[i].width, self._tex[i].height, self._tex[i].rgb = t.shape[1], t.shape[0], _fp(t)
        d = SceneDesc()
        d.n_triangles = n
        d.positions, d.normals, d.uvs = _fp(self.positions), _fp(self.normals), _fp(self.uvs)
        d.material_index = self.material_index.ctypes.data_as(C.POINTER(C.c_uint32))
        d.n_materials, d.materials = len(self.materials), self._mat
        d.n_textures, d.textures = len(self.textures), self._tex
        if self.env is not None:
            d.env_height, d.env_width, d.env_rgb = self.env.shape[0], self.env.shape[1], _fp(self.env)
        d.background = (C.c_float * 3)(*background)
        self.desc = d

    @property
    def n_triangles(self):
        return self.positions.shape[0]

    def save(self, path, frm, to, up, fov, width, height):
        """Write the scene + camera as an .rtxs dump for the headless C++ driver
        (ray-tracer_amd/host/main.cpp documents the layout)."""
        import struct
        with open(path, "wb") as f:
            f.write(b"RTXS" + struct.pack("<I", 1))
            f.write(struct.pack("<10f2I", *frm, *to, *up, fov, width, height))
            f.write(struct.pack("<I", len(self.textures)))
            for t in self.textures:
                f.write(struct.pack("<2I", t.shape[1], t.shape[0]) + t.tobytes())
            f.write(struct.pack("<I", len(self.materials)))
            for m in self.materials:
                f.write(bytes(m))
            f.write(struct.pack("<Q", self.n_triangles))
            f.write(self.positions.tobytes() + self.normals.tobytes() + self.uvs.tobytes() + self.material_index.tobytes())
            if self.env is not None:
                f.write(struct.pack("<2I", self.env.shape[1], self.env.shape[0]) + self.env.tobytes())
            else:
                f.write(struct.pack("<2I", 0, 0))
            f.write(struct.pack("<3f", *self.desc.background))


def material(type=BSDF_DIFFUSE, texture=0, emission=(0, 0, 0), eta=(0, 0, 0), k=(0, 0, 0), roughness=1.0,
             int_ior=1.33, ext_ior=1.0) -> Material:
    """Defaults follow the getValue(...) fallbacks of RTBase/SceneLoader.h:227-243."""
    return Material(type, texture, (C.c_float * 3)(*emission), (C.c_float * 3)(*eta), (C.c_float * 3)(*k),
                    roughness, int_ior, ext_ior)


def load_library(path: str = "") -> C.CDLL:
    """The shipped library, or with RTX_LIBRARY=tuning in the environment (or path=LIB_TUNING_PATH) its tuning build:
    the same code plus the schedules under test and the measurement switches that tools/ use."""
    if not path:
        path = LIB_TUNING_PATH if os.environ.get("RTX_LIBRARY") == "tuning" else LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()'); "
                           "there is no CPU fallback")
    lib = C.CDLL(path)
    vp, sz = C.c_void_p, C.c_size_t
    sig = {
        "rtx_create": [C.POINTER(vp), C.c_int], "rtx_set_stream": [vp, vp], "rtx_synchronize": [vp],
        "rtx_upload_scene": [vp, C.POINTER(SceneDesc)],
        "rtx_camera_look_at": [C.POINTER(Camera), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                               C.c_float, C.c_int, C.c_int, C.c_int],
        "rtx_set_camera": [vp, C.POINTER(Camera)], "rtx_set_settings": [vp, C.POINTER(Settings)],
        "rtx_clear": [vp], "rtx_render_passes": [vp, C.c_int], "rtx_get_spp": [vp, C.POINTER(C.c_int)],
        "rtx_set_spp": [vp, C.c_int],
        "rtx_film_read": [vp, vp, C.POINTER(C.c_int), vp],
        "rtx_film_device": [vp, C.POINTER(vp), C.POINTER(sz)], "rtx_film_write": [vp, vp, C.c_int],
        "rtx_tile_samples_device": [vp, C.POINTER(vp), C.POINTER(sz)], "rtx_update_tile_samples": [vp],
        "rtx_trace_closest": [vp, vp, sz, vp], "rtx_trace_closest_device": [vp, vp, sz, vp],
        "rtx_trace_closest_device_noalloc": [vp, vp, sz, vp],
        "rtx_trace_visible": [vp, vp, sz, vp], "rtx_trace_visible_device": [vp, vp, sz, vp],
        "rtx_generate_primary_rays_device": [vp, vp, vp, sz],
        "rtx_get_bvh": [vp, C.POINTER(C.c_uint32), vp, vp, vp],
        "rtx_create_aov": [vp, vp, vp, vp], "rtx_create_aov_device": [vp, vp, vp, vp],
        "rtx_get_stats": [vp, C.POINTER(Stats)], "rtx_reset_stats": [vp],
        "rtx_debug_shading_data": [vp, vp, vp, sz, vp], "rtx_debug_bsdf_table": [vp, C.c_uint32, vp, sz, vp],
        "rtx_debug_env_sample": [vp, vp, sz, vp], "rtx_debug_env_eval": [vp, vp, sz, vp],
        "rtx_timer_begin": [vp], "rtx_timer_end": [vp, C.POINTER(C.c_float)],
        "rtx_count_closest_work": [vp, vp, sz, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
        "rtx_count_render_work": [vp, C.c_int, C.POINTER(C.c_uint64)],
        "rtx_rccl_unique_id": [C.c_char_p], "rtx_rccl_comm_create": [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_char_p],
        "rtx_rccl_comm_destroy": [vp], "rtx_collective_rccl": [C.POINTER(Collective), vp],
        "rtx_film_allreduce": [vp, C.POINTER(Collective)], "rtx_render_sharded": [vp, C.POINTER(Collective)],
    }
    sig["rtx_host_bvh_build"] = [C.POINTER(SceneDesc), C.POINTER(vp)]
    sig["rtx_host_bvh_get"] = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), vp, vp, vp, C.POINTER(C.c_uint32),
                               C.POINTER(C.c_uint32)]
    for name, args in sig.items():
        f = getattr(lib, name)
        f.argtypes, f.restype = args, C.c_int
    lib.rtx_destroy.argtypes, lib.rtx_destroy.restype = [vp], None
    lib.rtx_host_bvh_free.argtypes, lib.rtx_host_bvh_free.restype = [vp], None
    lib.rtx_last_error.argtypes, lib.rtx_last_error.restype = [], C.c_char_p
    return lib


class RtxError(RuntimeError):
    pass


def host_bvh(lib, scene: "SceneArrays"):
    """Host-only BVH build (no GPU needed): returns bounds[n,6], topo[n,4], indices, depth,
    wide node count, wide depth."""
    h = C.c_void_p()
    if lib.rtx_host_bvh_build(C.byref(scene.desc), C.byref(h)):
        raise RtxError(lib.rtx_last_error().decode())
    try:
        n, d, wn, wd = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib.rtx_host_bvh_get(h, C.byref(n), C.byref(d), None, None, None, C.byref(wn), C.byref(wd))
        bounds = np.empty((n.value, 6), np.float32)
        topo = np.empty((n.value, 4), np.uint32)
        idx = np.empty(scene.n_triangles, np.uint32)
        lib.rtx_host_bvh_get(h, C.byref(n), C.byref(d), bounds.ctypes.data, topo.ctypes.data, idx.ctypes.data,
                             C.byref(wn), C.byref(wd))
        return bounds, topo, idx, d.value, wn.value, wd.value
    finally:
        lib.rtx_host_bvh_free(h)


def look_at(lib, frm, to, up, fov, width, height, flip_x=0) -> Camera:
    cam = Camera()
    a = [(C.c_float * 3)(*v) for v in (frm, to, up)]
    rc = lib.rtx_camera_look_at(C.byref(cam), a[0], a[1], a[2], fov, width, height, flip_x)
    if rc:
        raise RtxError(lib.rtx_last_error().decode())
    return cam


class Core:
    """Thin object wrapper over one rtx_ctx (one GPU)."""

    def __init__(self, device: int = 0, lib: C.CDLL | None = None):
        self.lib = lib or load_library()
        self.ctx = C.c_void_p()
        self._check(self.lib.rtx_create(C.byref(self.ctx), device))
        self.width = self.height = 0

    def _check(self, rc):
        if rc:
            raise RtxError(f"rtx error {rc}: {self.lib.rtx_last_error().decode()}")

    def close(self):
        if self.ctx:
            self.lib.rtx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # scene / camera / settings
    def upload_scene(self, scene: SceneArrays):
        self._check(self.lib.rtx_upload_scene(self.ctx, C.byref(scene.desc)))
        self._ntris = scene.n_triangles

    def set_camera(self, cam: Camera):
        self._check(self.lib.rtx_set_camera(self.ctx, C.byref(cam)))
        self.width, self.height = cam.width, cam.height

    def set_settings(self, s: Settings):
        self._check(self.lib.rtx_set_settings(self.ctx, C.byref(s)))

    def set_stream(self, stream_ptr):
        """Work goes to the hipStream_t `stream_ptr` from now on. None or 0 is HIP's null stream (as for any hipStream_t);
        STREAM_OWN gives the context its own non-blocking stream back."""
        self._check(self.lib.rtx_set_stream(self.ctx, C.c_void_p(stream_ptr or 0)))

    def synchronize(self):
        self._check(self.lib.rtx_synchronize(self.ctx))

    # rendering
    def clear(self):
        self._check(self.lib.rtx_clear(self.ctx))

    def render_passes(self, n: int = 1):
        self._check(self.lib.rtx_render_passes(self.ctx, n))

    def spp(self) -> int:
        v = C.c_int()
        self._check(self.lib.rtx_get_spp(self.ctx, C.byref(v)))
        return v.value

    def set_spp(self, v: int):
        self._check(self.lib.rtx_set_spp(self.ctx, v))

    def film(self):
        rgb = np.empty((self.height, self.width, 3), np.float32)
        tiles = np.empty(((self.height + 15) // 16) * ((self.width + 15) // 16), np.uint32)
        spp = C.c_int()
        self._check(self.lib.rtx_film_read(self.ctx, rgb.ctypes.data, C.byref(spp), tiles.ctypes.data))
        return rgb, spp.value, tiles

    def film_write(self, rgb, spp: int):
        rgb = np.ascontiguousarray(rgb, np.float32)
        assert rgb.shape == (self.height, self.width, 3)
        self._check(self.lib.rtx_film_write(self.ctx, rgb.ctypes.data, spp))

    def film_device(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self.lib.rtx_film_device(self.ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def tile_samples_device(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self.lib.rtx_tile_samples_device(self.ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def update_tile_samples(self):
        self._check(self.lib.rtx_update_tile_samples(self.ctx))

    # ray queries (host arrays)
    def trace_closest(self, rays: np.ndarray, out: np.ndarray | None = None) -> np.ndarray:
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        hits = out if out is not None else np.empty(rays.shape[0], HIT_DTYPE)     # (a fresh array costs its page faults: reuse `out` in a loop)
        assert hits.dtype == HIT_DTYPE and hits.shape[0] == rays.shape[0] and hits.flags.c_contiguous
        self._check(self.lib.rtx_trace_closest(self.ctx, rays.ctypes.data, rays.shape[0], hits.ctypes.data))
        return hits

    def trace_visible(self, p1p2: np.ndarray) -> np.ndarray:
        p = np.ascontiguousarray(p1p2, np.float32).reshape(-1, 6)
        vis = np.empty(p.shape[0], np.uint8)
        self._check(self.lib.rtx_trace_visible(self.ctx, p.ctypes.data, p.shape[0], vis.ctypes.data))
        return vis

    # ray queries (device pointers, asynchronous)
    def trace_closest_device(self, rays_ptr: int, n: int, hits_ptr: int):
        self._check(self.lib.rtx_trace_closest_device(self.ctx, rays_ptr, n, hits_ptr))

    def trace_closest_device_noalloc(self, rays_ptr: int, n: int, hits_ptr: int):
        """The variant that allocates nothing (static scratch, fixed limits; see include/rtx.h)."""
        self._check(self.lib.rtx_trace_closest_device_noalloc(self.ctx, rays_ptr, n, hits_ptr))

    def trace_visible_device(self, seg_ptr: int, n: int, vis_ptr: int):
        self._check(self.lib.rtx_trace_visible_device(self.ctx, seg_ptr, n, vis_ptr))

    def generate_primary_rays_device(self, rays_ptr: int, pixel_ptr: int, capacity: int):
        self._check(self.lib.rtx_generate_primary_rays_device(self.ctx, rays_ptr, pixel_ptr, capacity))

    def count_closest_work(self, rays_ptr: int, n: int):
        a, b, g = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.lib.rtx_count_closest_work(self.ctx, rays_ptr, n, C.byref(a), C.byref(b), C.byref(g)))
        return a.value, b.value, g.value

    def render_sharded(self, collective: "Collective | None" = None):
        """rtx_render_sharded: this context's share of settings.total_spp global samples, then the film sum."""
        self._check(self.lib.rtx_render_sharded(self.ctx, C.byref(collective) if collective is not None else None))

    def film_allreduce(self, collective: "Collective"):
        self._check(self.lib.rtx_film_allreduce(self.ctx, C.byref(collective)))

    def count_render_work(self, n_passes: int = 1) -> dict:
        """n instrumented passes (the film advances as usual): rays, nodes visited, triangles tested."""
        out = (C.c_uint64 * 8)()
        self._check(self.lib.rtx_count_render_work(self.ctx, n_passes, out))
        return dict(zip(("closest_rays", "shadow_rays", "nodes", "tris", "samples", "tri_groups", "gather_segments", "vpls"), (int(v) for v in out)))

    # function-level parity tables (layouts in include/rtx.h)
    def debug_shading_data(self, rays, hits):
        rays = np.ascontiguousarray(rays); hits = np.ascontiguousarray(hits)
        out = np.empty((rays.shape[0], 24), np.float32)
        self._check(self.lib.rtx_debug_shading_data(self.ctx, rays.ctypes.data, hits.ctypes.data, rays.shape[0], out.ctypes.data))
        return out

    def debug_bsdf_table(self, material: int, table):
        t = np.ascontiguousarray(table, np.float32).reshape(-1, 14)
        out = np.empty((t.shape[0], 11), np.float32)
        self._check(self.lib.rtx_debug_bsdf_table(self.ctx, material, t.ctypes.data, t.shape[0], out.ctypes.data))
        return out

    def debug_env_sample(self, u01):
        u = np.ascontiguousarray(u01, np.float32).reshape(-1, 2)
        out = np.empty((u.shape[0], 7), np.float32)
        self._check(self.lib.rtx_debug_env_sample(self.ctx, u.ctypes.data, u.shape[0], out.ctypes.data))
        return out

    def debug_env_eval(self, dirs):
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.empty((d.shape[0], 4), np.float32)
        self._check(self.lib.rtx_debug_env_eval(self.ctx, d.ctypes.data, d.shape[0], out.ctypes.data))
        return out

    # introspection
    def bvh(self):
        n = C.c_uint32()
        self._check(self.lib.rtx_get_bvh(self.ctx, C.byref(n), None, None, None))
        st = self.stats()
        bounds = np.empty((n.value, 6), np.float32)
        topo = np.empty((n.value, 4), np.uint32)
        idx = np.empty(self._ntris, np.uint32) if hasattr(self, "_ntris") else None
        self._check(self.lib.rtx_get_bvh(self.ctx, C.byref(n), bounds.ctypes.data, topo.ctypes.data,
                                         idx.ctypes.data if idx is not None else None))
        return bounds, topo, idx, st.bvh2_depth

    def stats(self) -> Stats:
        s = Stats()
        self._check(self.lib.rtx_get_stats(self.ctx, C.byref(s)))
        return s

    def create_aov(self):
        """RayTracer::createAOV, RTBase/Renderer.h:786-820: (color, albedo, normal), H x W x 3 each."""
        w, h = self.width, self.height
        planes = [np.empty((h, w, 3), np.float32) for _ in range(3)]
        self._check(self.lib.rtx_create_aov(self.ctx, *(p.ctypes.data for p in planes)))
        return tuple(planes)

    def reset_stats(self):
        self._check(self.lib.rtx_reset_stats(self.ctx))

    def timer_begin(self):
        self._check(self.lib.rtx_timer_begin(self.ctx))

    def timer_end(self) -> float:
        ms = C.c_float()
        self._check(self.lib.rtx_timer_end(self.ctx, C.byref(ms)))
        return ms.value
