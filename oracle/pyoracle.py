"""ctypes loader of the CPU oracle (``oracle/liboracle.so``) and of the real-reference
math layer (``oracle/_ref/libref_core.so``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")
LIB_O3 = os.path.join(_HERE, "liboracle_o3.so")
REF_LIB = os.path.join(_HERE, "_ref", "libref_core.so")

_f = C.POINTER(C.c_float)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, rtx_pkg, lib_path: str = LIB):
        """rtx_pkg: the product package module (for the shared ctypes struct definitions). lib_path: LIB, or LIB_O3 for
        the -O3 / AVX2 build that bench.py times beside it (same source, same results)."""
        if not os.path.exists(lib_path):
            raise RuntimeError(f"{lib_path} missing: run make -C oracle")
        self.pkg = rtx_pkg
        lib = self.lib = C.CDLL(lib_path)
        vp = C.c_void_p
        lib.orc_scene_create.restype = vp
        lib.orc_scene_create.argtypes = [C.POINTER(rtx_pkg.SceneDesc)]
        lib.orc_scene_build_seconds.restype = C.c_double
        lib.orc_scene_build_seconds.argtypes = [C.POINTER(rtx_pkg.SceneDesc)]
        lib.orc_scene_destroy.argtypes = [vp]
        lib.orc_camera_look_at.argtypes = [C.POINTER(rtx_pkg.Camera), _f, _f, _f, C.c_float, C.c_int, C.c_int, C.c_int]
        lib.orc_set_camera.argtypes = [vp, C.POINTER(rtx_pkg.Camera)]
        lib.orc_num_lights.argtypes = [vp]; lib.orc_num_lights.restype = C.c_uint
        lib.orc_get_bvh.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), vp, vp, vp]
        lib.orc_ray_aabb.argtypes = [_f, _f, _f, _f]; lib.orc_ray_aabb.restype = C.c_int
        lib.orc_generate_rays.argtypes = [vp, vp, C.c_size_t, vp]
        lib.orc_trace_closest.argtypes = [vp, vp, C.c_size_t, vp, C.c_uint]
        lib.orc_trace_visible.argtypes = [vp, vp, C.c_size_t, vp, C.c_uint]
        lib.orc_shading_data.argtypes = [vp, vp, vp, C.c_size_t, vp]
        lib.orc_bsdf_table.argtypes = [vp, C.c_uint, vp, C.c_size_t, vp]
        lib.orc_env_tables.argtypes = [vp, vp, vp, _f]; lib.orc_env_tables.restype = C.c_int
        lib.orc_env_sample.argtypes = [vp, vp, C.c_size_t, vp]; lib.orc_env_sample.restype = C.c_int
        lib.orc_env_eval.argtypes = [vp, vp, C.c_size_t, vp]; lib.orc_env_eval.restype = C.c_int
        lib.orc_renderer_create.restype = vp
        lib.orc_renderer_create.argtypes = [vp, C.POINTER(rtx_pkg.Settings), C.c_uint, C.c_int]
        lib.orc_renderer_destroy.argtypes = [vp]
        lib.orc_renderer_clear.argtypes = [vp]
        lib.orc_render_passes.argtypes = [vp, C.c_int]
        lib.orc_num_vpls.argtypes = [vp]; lib.orc_num_vpls.restype = C.c_uint
        lib.orc_get_vpls.argtypes = [vp, vp]
        lib.orc_film_read.argtypes = [vp, vp, C.POINTER(C.c_int), vp]
        lib.orc_film_write.argtypes = [vp, vp, C.c_int]
        lib.orc_calculate_tile_samples.argtypes = [vp]
        lib.orc_tile_samples_write.argtypes = [vp, vp]
        lib.orc_create_aov.argtypes = [vp, vp, vp, vp]
        lib.orc_film_splat.argtypes = [vp, vp, C.c_size_t]
        lib.orc_rng.argtypes = [C.c_uint32] * 5; lib.orc_rng.restype = C.c_float

    def look_at(self, frm, to, up, fov, w, h, flip=0):
        cam = self.pkg.Camera()
        a = [(C.c_float * 3)(*v) for v in (frm, to, up)]
        self.lib.orc_camera_look_at(C.byref(cam), a[0], a[1], a[2], fov, w, h, flip)
        return cam

    def ray_aabb(self, bmin, bmax, o, d):
        a = [(C.c_float * 3)(*v) for v in (bmin, bmax, o, d)]
        return bool(self.lib.orc_ray_aabb(*a))

    def scene(self, arrays):
        return OracleScene(self, arrays)

    def build_seconds(self, arrays):
        return self.lib.orc_scene_build_seconds(C.byref(arrays.desc))


class OracleScene:
    def __init__(self, orc: Oracle, arrays):
        self.o, self.lib, self.arrays = orc, orc.lib, arrays
        self.h = self.lib.orc_scene_create(C.byref(arrays.desc))
        self.width = self.height = 0

    def close(self):
        if self.h:
            self.lib.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, cam):
        self.lib.orc_set_camera(self.h, C.byref(cam))
        self.width, self.height = cam.width, cam.height

    def num_lights(self):
        return self.lib.orc_num_lights(self.h)

    def bvh(self):
        n, d = C.c_uint32(), C.c_uint32()
        self.lib.orc_get_bvh(self.h, C.byref(n), C.byref(d), None, None, None)
        bounds = np.empty((n.value, 6), np.float32)
        topo = np.empty((n.value, 4), np.uint32)
        idx = np.empty(self.arrays.n_triangles, np.uint32)
        self.lib.orc_get_bvh(self.h, C.byref(n), C.byref(d), _p(bounds), _p(topo), _p(idx))
        return bounds, topo, idx, d.value

    def generate_rays(self, xy):
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        rays = np.empty(xy.shape[0], self.o.pkg.RAY_DTYPE)
        self.lib.orc_generate_rays(self.h, _p(xy), xy.shape[0], _p(rays))
        return rays

    def trace_closest(self, rays, threads=8):
        rays = np.ascontiguousarray(rays, self.o.pkg.RAY_DTYPE)
        hits = np.empty(rays.shape[0], self.o.pkg.HIT_DTYPE)
        self.lib.orc_trace_closest(self.h, _p(rays), rays.shape[0], _p(hits), threads)
        return hits

    def trace_visible(self, p1p2, threads=8):
        p = np.ascontiguousarray(p1p2, np.float32).reshape(-1, 6)
        vis = np.empty(p.shape[0], np.uint8)
        self.lib.orc_trace_visible(self.h, _p(p), p.shape[0], _p(vis), threads)
        return vis

    def shading_data(self, rays, hits):
        out = np.empty((rays.shape[0], 24), np.float32)
        self.lib.orc_shading_data(self.h, _p(rays), _p(hits), rays.shape[0], _p(out))
        return out

    def bsdf_table(self, mat, table):
        t = np.ascontiguousarray(table, np.float32).reshape(-1, 14)
        out = np.empty((t.shape[0], 11), np.float32)
        self.lib.orc_bsdf_table(self.h, mat, _p(t), t.shape[0], _p(out))
        return out

    def env_tables(self):
        h, w = self.arrays.env.shape[:2]
        rows, cols, avg = np.empty(h, np.float32), np.empty((h, w), np.float32), C.c_float()
        assert self.lib.orc_env_tables(self.h, _p(rows), _p(cols), C.byref(avg)) == 0
        return rows, cols, avg.value

    def env_sample(self, u01):
        u = np.ascontiguousarray(u01, np.float32).reshape(-1, 2)
        out = np.empty((u.shape[0], 7), np.float32)
        assert self.lib.orc_env_sample(self.h, _p(u), u.shape[0], _p(out)) == 0
        return out

    def env_eval(self, dirs):
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.empty((d.shape[0], 4), np.float32)
        self.lib.orc_env_eval(self.h, _p(d), d.shape[0], _p(out))
        return out

    def renderer(self, settings, threads=8, counter_rng=True):
        return OracleRenderer(self, settings, threads, counter_rng)


class OracleRenderer:
    def __init__(self, scene: OracleScene, settings, threads, counter_rng):
        self.s, self.lib = scene, scene.lib
        self.h = self.lib.orc_renderer_create(scene.h, C.byref(settings), threads, 1 if counter_rng else 0)

    def close(self):
        if self.h:
            self.lib.orc_renderer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        self.lib.orc_renderer_clear(self.h)

    def render_passes(self, n=1):
        self.lib.orc_render_passes(self.h, n)

    def film(self):
        w, h = self.s.width, self.s.height
        rgb = np.empty((h, w, 3), np.float32)
        tiles = np.empty(((h + 15) // 16) * ((w + 15) // 16), np.uint32)
        spp = C.c_int()
        self.lib.orc_film_read(self.h, _p(rgb), C.byref(spp), _p(tiles))
        return rgb, spp.value, tiles

    def film_write(self, rgb, spp):
        rgb = np.ascontiguousarray(rgb, np.float32)
        self.lib.orc_film_write(self.h, _p(rgb), spp)

    def create_aov(self):
        w, h = self.s.width, self.s.height
        c, a, n = (np.empty((h, w, 3), np.float32) for _ in range(3))
        self.lib.orc_create_aov(self.h, _p(c), _p(a), _p(n))
        return c, a, n

    def calculate_tile_samples(self):
        self.lib.orc_calculate_tile_samples(self.h)

    def tile_samples_write(self, tiles):
        t = np.ascontiguousarray(tiles, np.uint32)
        self.lib.orc_tile_samples_write(self.h, _p(t))

    def splat(self, xyrgb):
        a = np.ascontiguousarray(xyrgb, np.float32).reshape(-1, 5)
        self.lib.orc_film_splat(self.h, _p(a), a.shape[0])

    def vpls(self):
        n = self.lib.orc_num_vpls(self.h)
        out = np.empty((n, 9), np.float32)
        if n:
            self.lib.orc_get_vpls(self.h, _p(out))
        return out


class RefCore:
    """The real RTBase/Core.h + RTBase/Sampling.h (only present where oracle/_ref was built)."""

    def __init__(self):
        if not os.path.exists(REF_LIB):
            raise FileNotFoundError(REF_LIB)
        lib = self.lib = C.CDLL(REF_LIB)
        for n in ("ref_matrix_invert",):
            getattr(lib, n).argtypes = [_f, _f]
        lib.ref_look_at.argtypes = [_f, _f, _f, _f]
        lib.ref_perspective.argtypes = [C.c_float] * 4 + [_f]
        for n in ("ref_mul_point", "ref_mul_vec", "ref_mul_point_divide", "ref_cross", "ref_frame"):
            getattr(lib, n).argtypes = [_f, _f, _f]
        lib.ref_normalize.argtypes = [_f, _f]
        lib.ref_dot.argtypes = [_f, _f]; lib.ref_dot.restype = C.c_float
        lib.ref_lum.argtypes = [_f]; lib.ref_lum.restype = C.c_float
        lib.ref_warp.argtypes = [C.c_int, C.c_float, C.c_float, _f]
        lib.ref_mt_stream.argtypes = [C.c_uint, C.c_int, _f]
        lib.ref_warp_from_sampler.argtypes = [C.c_int, C.c_uint, _f]
        lib.ref_spherical.argtypes = [C.c_float, C.c_float, _f, _f]

    @staticmethod
    def _a(v):
        a = np.ascontiguousarray(v, np.float32)
        return a, a.ctypes.data_as(_f)

    def invert(self, m):
        a, pa = self._a(m); o = np.empty(16, np.float32)
        self.lib.ref_matrix_invert(pa, o.ctypes.data_as(_f)); return o

    def look_at(self, frm, to, up):
        (a, pa), (b, pb), (c, pc) = self._a(frm), self._a(to), self._a(up); o = np.empty(16, np.float32)
        self.lib.ref_look_at(pa, pb, pc, o.ctypes.data_as(_f)); return o

    def perspective(self, n, f, aspect, fov):
        o = np.empty(16, np.float32); self.lib.ref_perspective(n, f, aspect, fov, o.ctypes.data_as(_f)); return o

    def _mv(self, fn, m, v):
        (a, pa), (b, pb) = self._a(m), self._a(v); o = np.empty(3, np.float32)
        fn(pa, pb, o.ctypes.data_as(_f)); return o

    def mul_point(self, m, v): return self._mv(self.lib.ref_mul_point, m, v)
    def mul_vec(self, m, v): return self._mv(self.lib.ref_mul_vec, m, v)
    def mul_point_divide(self, m, v): return self._mv(self.lib.ref_mul_point_divide, m, v)

    def normalize(self, v):
        a, pa = self._a(v); o = np.empty(3, np.float32); self.lib.ref_normalize(pa, o.ctypes.data_as(_f)); return o

    def frame(self, n, vec):
        (a, pa), (b, pb) = self._a(n), self._a(vec); o = np.empty(15, np.float32)
        self.lib.ref_frame(pa, pb, o.ctypes.data_as(_f)); return o

    def warp(self, kind, r1, r2):
        o = np.empty(4, np.float32); self.lib.ref_warp(kind, r1, r2, o.ctypes.data_as(_f)); return o

    def mt_stream(self, seed, n):
        o = np.empty(n, np.float32); self.lib.ref_mt_stream(seed, n, o.ctypes.data_as(_f)); return o

    def warp_from_sampler(self, kind, seed):
        o = np.empty(3, np.float32); self.lib.ref_warp_from_sampler(kind, seed, o.ctypes.data_as(_f)); return o

    def spherical(self, theta, phi, wi):
        (a, pa) = self._a(wi); o = np.empty(5, np.float32)
        self.lib.ref_spherical(theta, phi, pa, o.ctypes.data_as(_f)); return o


class RefGem:
    """The reference's own scene reader (RTBase/GEMLoader.h compiled as it is into
    oracle/_ref/libref_gem.so): digest of a scene directory, see oracle/ref_gem.cpp."""

    def __init__(self):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_gem.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.lib.ref_gem_digest.argtypes = [C.c_char_p, C.POINTER(C.c_uint64)]

    def digest(self, directory):
        out = (C.c_uint64 * 8)()
        assert self.lib.ref_gem_digest(str(directory).encode(), out) == 0
        return [int(v) for v in out]


class RefStb:
    """The reference's image decoder (RTBase/stb_image.h compiled as it is into
    oracle/_ref/libref_stb.so): width, height, channels and pixel hash of a file, see oracle/ref_stb.cpp."""

    def __init__(self):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_stb.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.lib.ref_stb_digest.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint64)]

    def digest(self, path):
        out = (C.c_uint64 * 4)()
        rc = self.lib.ref_stb_digest(str(path).encode(), 1 if str(path).endswith(".hdr") else 0, out)
        return None if rc else [int(v) for v in out]
