"""Deterministic synthetic scenes for the benchmark configurations and the parity tests.

The reference's own scene assets live in a separate repository that is not available
offline (RTBase/Main.cpp:39-63 only names them), so every scene here is generated.
All generators return a :class:`SceneArrays` (see the package ``__init__``).
"""
from __future__ import annotations

import importlib

import numpy as np

_pkg = importlib.import_module(__name__.rsplit(".", 1)[0])
SceneArrays, material = _pkg.SceneArrays, _pkg.material

WHITE = np.ones((1, 1, 3), np.float32)


def _face_normals(pos):
    e1 = pos[:, 0] - pos[:, 2]
    e2 = pos[:, 1] - pos[:, 2]
    n = np.cross(e1, e2)
    l = np.linalg.norm(n, axis=1, keepdims=True)
    l[l == 0] = 1
    return (n / l).astype(np.float32)


def _drop_degenerate(pos, *rest):
    """Triangles of zero area are dropped by the loader (RTBase/SceneLoader.h:330)."""
    e1 = pos[:, 0] - pos[:, 2]
    e2 = pos[:, 1] - pos[:, 2]
    c = np.cross(e1.astype(np.float32), e2.astype(np.float32))
    area = np.sqrt((c.astype(np.float32) ** 2).sum(1, dtype=np.float32)) * np.float32(0.5)
    keep = area > 0
    return (pos[keep],) + tuple(r[keep] for r in rest)


def soup(n_triangles=1_000_000, seed=12345, extent=1.0, size=0.01):
    """Triangle soup: centres uniform in [-extent, extent]^3, vertices centre + U[-size, size]^3."""
    rs = np.random.RandomState(seed)
    centre = rs.uniform(-extent, extent, (n_triangles, 1, 3)).astype(np.float32)
    pos = (centre + rs.uniform(-size, size, (n_triangles, 3, 3)).astype(np.float32)).astype(np.float32)
    uv = rs.uniform(0, 1, (n_triangles, 3, 2)).astype(np.float32)
    pos, uv = _drop_degenerate(pos, uv)
    nrm = np.repeat(_face_normals(pos)[:, None, :], 3, axis=1)
    mats = [material(_pkg.BSDF_DIFFUSE, 0)]
    return SceneArrays(pos, nrm, uv, np.zeros(pos.shape[0], np.uint32), mats, [WHITE * 0.8])


def _quad(a, b, c, d):
    return [[a, b, c], [a, c, d]]


def _box(lo, hi):
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    p = lambda x, y, z: (x, y, z)
    t = []
    t += _quad(p(x0, y0, z1), p(x1, y0, z1), p(x1, y1, z1), p(x0, y1, z1))  # front
    t += _quad(p(x1, y0, z0), p(x0, y0, z0), p(x0, y1, z0), p(x1, y1, z0))  # back
    t += _quad(p(x0, y0, z0), p(x0, y0, z1), p(x0, y1, z1), p(x0, y1, z0))  # left
    t += _quad(p(x1, y0, z1), p(x1, y0, z0), p(x1, y1, z0), p(x1, y1, z1))  # right
    t += _quad(p(x0, y1, z1), p(x1, y1, z1), p(x1, y1, z0), p(x0, y1, z0))  # top
    t += _quad(p(x0, y0, z0), p(x1, y0, z0), p(x1, y0, z1), p(x0, y0, z1))  # bottom
    return t


def cornell_box(materials="diffuse", background=(0.0, 0.0, 0.0), emissive=True):
    """Cornell-box stand-in for scene 0 (scenes/cornell-box, RTBase/Main.cpp:39): five walls,
    two boxes, one emissive quad Le = (17, 12, 4), black background.

    materials="mixed" swaps the two boxes and the back wall for Mirror / Glass / Plastic /
    Conductor so that every BSDF of RTBase/Materials.h is exercised in a tiny scene;
    materials="next" uses the remaining classes instead: Oren-Nayar walls, a Dielectric tall
    box, a Layered(Glass) short box and a Layered(Plastic) small box.

    background: the constant background colour (BackgroundColour, RTBase/Lights.h:86-135); a non-black
    one joins the light list (RTBase/Scene.h:165-168) and shines in through the open front.
    emissive=False turns the ceiling quad into a plain diffuse one, so that the background is the
    only light.
    """
    tris, mat = [], []

    def add(ts, m):
        tris.extend(ts)
        mat.extend([m] * len(ts))

    add(_quad((-1, -1, -1), (1, -1, -1), (1, -1, 1), (-1, -1, 1)), 0)      # floor (white)
    add(_quad((-1, 1, 1), (1, 1, 1), (1, 1, -1), (-1, 1, -1)), 0)          # ceiling
    extra = materials in ("mixed", "next")
    add(_quad((-1, -1, -1), (-1, 1, -1), (1, 1, -1), (1, -1, -1)), 5 if extra else 0)  # back
    add(_quad((-1, -1, 1), (-1, 1, 1), (-1, 1, -1), (-1, -1, -1)), 1)      # left (red)
    add(_quad((1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1)), 2)          # right (green)
    add(_quad((-0.25, 0.999, -0.25), (0.25, 0.999, -0.25), (0.25, 0.999, 0.25), (-0.25, 0.999, 0.25)), 3)  # light
    add(_box((-0.7, -1, -0.6), (-0.1, 0.2, 0.0)), 4 if extra else 0)   # tall box
    add(_box((0.1, -1, 0.0), (0.7, -0.4, 0.6)), 6 if extra else 0)     # short box
    if extra:
        add(_box((-0.2, -1, 0.5), (0.1, -0.7, 0.8)), 7)
    pos = np.array(tris, np.float32)
    nrm = np.repeat(_face_normals(pos)[:, None, :], 3, axis=1)
    # emissive quad must face down: make its vertex normals agree with its winding
    uv = np.tile(np.array([[0, 0], [1, 0], [1, 1]], np.float32), (pos.shape[0], 1, 1))
    tex = [WHITE * 0.73, np.array([[[0.63, 0.065, 0.05]]], np.float32), np.array([[[0.14, 0.45, 0.091]]], np.float32),
           WHITE.copy()]
    mats = [material(_pkg.BSDF_DIFFUSE, 0), material(_pkg.BSDF_DIFFUSE, 1), material(_pkg.BSDF_DIFFUSE, 2),
            material(_pkg.BSDF_DIFFUSE, 0, emission=(17, 12, 4) if emissive else (0, 0, 0)),
            material(_pkg.BSDF_MIRROR, 3),
            material(_pkg.BSDF_PLASTIC, 1, roughness=0.1, int_ior=1.5, ext_ior=1.0),
            material(_pkg.BSDF_GLASS, 3, int_ior=1.5, ext_ior=1.0),
            material(_pkg.BSDF_CONDUCTOR, 3, eta=(0.17, 0.35, 1.5), k=(3.1, 2.7, 1.9), roughness=0.2)]
    if materials == "next":
        L = _pkg.BSDF_LAYERED
        mats = [material(_pkg.BSDF_ORENNAYAR, 0, roughness=0.6), material(_pkg.BSDF_ORENNAYAR, 1, roughness=0.3),
                material(_pkg.BSDF_ORENNAYAR, 2, roughness=1.0),
                material(_pkg.BSDF_DIFFUSE, 0, emission=(17, 12, 4) if emissive else (0, 0, 0)),
                material(_pkg.BSDF_DIELECTRIC, 3, roughness=0.2, int_ior=1.5, ext_ior=1.0),
                material(_pkg.BSDF_ORENNAYAR | L, 1, roughness=0.45),
                material(_pkg.BSDF_GLASS | L, 3, int_ior=1.5, ext_ior=1.0),
                material(_pkg.BSDF_PLASTIC | L, 2, roughness=0.1, int_ior=1.5, ext_ior=1.0)]
    return SceneArrays(pos, nrm, uv, np.array(mat, np.uint32), mats, tex, background=background)


def cornell_camera():
    return dict(frm=(0.0, 0.0, 3.4), to=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=45.0)


def _sphere(centre, radius, nu, nv):
    """UV sphere, smooth normals; returns (pos[n,3,3], nrm[n,3,3], uv[n,3,2])."""
    u = np.linspace(0, 2 * np.pi, nu + 1)
    v = np.linspace(0, np.pi, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="xy")
    n = np.stack([np.sin(vv) * np.cos(uu), np.cos(vv), np.sin(vv) * np.sin(uu)], -1)
    p = np.asarray(centre) + radius * n
    t = np.stack([uu / (2 * np.pi), vv / np.pi], -1)
    i0 = (slice(0, nv), slice(0, nu))
    i1 = (slice(0, nv), slice(1, nu + 1))
    i2 = (slice(1, nv + 1), slice(1, nu + 1))
    i3 = (slice(1, nv + 1), slice(0, nu))

    def tri(a, b, c, arr):
        return np.stack([arr[a].reshape(-1, arr.shape[-1]), arr[b].reshape(-1, arr.shape[-1]),
                         arr[c].reshape(-1, arr.shape[-1])], 1)

    pos = np.concatenate([tri(i0, i2, i1, p), tri(i0, i3, i2, p)]).astype(np.float32)
    nrm = np.concatenate([tri(i0, i2, i1, n), tri(i0, i3, i2, n)]).astype(np.float32)
    uv = np.concatenate([tri(i0, i2, i1, t), tri(i0, i3, i2, t)]).astype(np.float32)
    return pos, nrm, uv


def checker(n=256, cells=8, a=(0.8, 0.8, 0.8), b=(0.15, 0.2, 0.6)):
    y, x = np.mgrid[0:n, 0:n]
    m = ((x * cells // n) + (y * cells // n)) % 2
    return np.where(m[..., None] == 0, np.array(a, np.float32), np.array(b, np.float32)).astype(np.float32)


def sky_env(width=1024, height=512, sun_dir=(0.4, 0.6, 0.3), sun_radiance=400.0):
    """Procedural lat-long environment: gradient sky + a small sun disc (analytic)."""
    v = (np.arange(height) + 0.5) / height
    u = (np.arange(width) + 0.5) / width
    theta = v[:, None] * np.pi
    phi = u[None, :] * 2 * np.pi
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta) * np.ones_like(phi), np.sin(theta) * np.sin(phi)], -1)
    up = np.clip(d[..., 1], -1, 1)
    sky = np.stack([0.25 + 0.35 * (1 - up), 0.4 + 0.3 * (1 - up), 0.9 - 0.1 * up], -1) * np.where(up[..., None] > 0, 1.0, 0.15)
    s = np.asarray(sun_dir, np.float64)
    s = s / np.linalg.norm(s)
    cosang = (d * s).sum(-1)
    sun = np.where(cosang > np.cos(np.radians(3.0)), sun_radiance, 0.0)[..., None] * np.array([1.0, 0.93, 0.8])
    return (sky + sun).astype(np.float32)


def mixed_scene(tess=128, env_size=(1024, 512)):
    """Ground plane + five spheres (Diffuse checker / Mirror / Glass 1.5 / gold Conductor
    roughness 0.2 / Plastic 1.5 roughness 0.1) under a procedural HDR environment map."""
    parts, mats_idx = [], []
    g = 64
    xs = np.linspace(-6, 6, g + 1)
    gp, gn, gu = [], [], []
    for i in range(g):
        for j in range(g):
            a = (xs[i], 0.0, xs[j]); b = (xs[i + 1], 0.0, xs[j]); c = (xs[i + 1], 0.0, xs[j + 1]); d = (xs[i], 0.0, xs[j + 1])
            ua = (i / g, j / g); ub = ((i + 1) / g, j / g); uc = ((i + 1) / g, (j + 1) / g); ud = (i / g, (j + 1) / g)
            gp += [[a, c, b], [a, d, c]]
            gu += [[ua, uc, ub], [ua, ud, uc]]
    gp = np.array(gp, np.float32); gu = np.array(gu, np.float32)
    gn = np.tile(np.array([0, 1, 0], np.float32), (gp.shape[0], 3, 1))
    parts.append((gp, gn, gu)); mats_idx.append(np.full(gp.shape[0], 0, np.uint32))
    centres = [(-3.0, 1.0, 0.0), (-1.5, 1.0, 0.6), (0.0, 1.0, 0.0), (1.5, 1.0, 0.6), (3.0, 1.0, 0.0)]
    for k, c in enumerate(centres):
        p, n, t = _sphere(c, 0.7, tess, tess // 2)
        parts.append((p, n, t)); mats_idx.append(np.full(p.shape[0], k + 1, np.uint32))
    pos = np.concatenate([p[0] for p in parts]); nrm = np.concatenate([p[1] for p in parts]); uv = np.concatenate([p[2] for p in parts])
    mi = np.concatenate(mats_idx)
    pos, nrm, uv, mi = _drop_degenerate(pos, nrm, uv, mi)
    tex = [checker(), WHITE.copy(), np.array([[[0.7, 0.3, 0.25]]], np.float32)]
    mats = [material(_pkg.BSDF_DIFFUSE, 0),
            material(_pkg.BSDF_DIFFUSE, 2),
            material(_pkg.BSDF_MIRROR, 1),
            material(_pkg.BSDF_GLASS, 1, int_ior=1.5, ext_ior=1.0),
            material(_pkg.BSDF_CONDUCTOR, 1, eta=(0.17, 0.35, 1.5), k=(3.1, 2.7, 1.9), roughness=0.2),
            material(_pkg.BSDF_PLASTIC, 2, roughness=0.1, int_ior=1.5, ext_ior=1.0)]
    env = sky_env(*env_size) if env_size else None
    return SceneArrays(pos, nrm, uv, mi, mats, tex, env=env)


def mixed_camera():
    return dict(frm=(0.0, 2.2, 7.5), to=(0.0, 0.8, 0.0), up=(0.0, 1.0, 0.0), fov=40.0)


def terrain(grid=2236, seed=7):
    """Procedural height field with 2*grid^2 triangles (grid=2236 gives ~10 M), one emissive
    strip overhead and a diffuse checker ground; the ~10 M-triangle 4K configuration."""
    rs = np.random.RandomState(seed)
    xs = np.linspace(-10, 10, grid + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    Y = (0.6 * np.sin(0.7 * X) * np.cos(0.9 * Z) + 0.15 * np.sin(3.1 * X + 1.3) * np.sin(2.7 * Z)).astype(np.float32)
    Y += rs.uniform(-0.01, 0.01, Y.shape).astype(np.float32)
    P = np.stack([X, Y, Z], -1)
    dYdx = np.gradient(Y, xs, axis=0); dYdz = np.gradient(Y, xs, axis=1)
    N = np.stack([-dYdx, np.ones_like(Y), -dYdz], -1); N /= np.linalg.norm(N, axis=-1, keepdims=True)
    T = np.stack([(X + 10) / 20, (Z + 10) / 20], -1)
    a = (slice(0, grid), slice(0, grid)); b = (slice(1, grid + 1), slice(0, grid)); c = (slice(1, grid + 1), slice(1, grid + 1)); d = (slice(0, grid), slice(1, grid + 1))

    def tri(i, j, k, arr):
        return np.stack([arr[i].reshape(-1, arr.shape[-1]), arr[j].reshape(-1, arr.shape[-1]), arr[k].reshape(-1, arr.shape[-1])], 1)

    pos = np.concatenate([tri(a, c, b, P), tri(a, d, c, P)]).astype(np.float32)
    nrm = np.concatenate([tri(a, c, b, N), tri(a, d, c, N)]).astype(np.float32)
    uv = np.concatenate([tri(a, c, b, T), tri(a, d, c, T)]).astype(np.float32)
    mi = np.zeros(pos.shape[0], np.uint32)
    light = np.array(_quad((-4, 6, -4), (4, 6, -4), (4, 6, 4), (-4, 6, 4)), np.float32)
    pos = np.concatenate([pos, light]); nrm = np.concatenate([nrm, np.tile(np.array([0, -1, 0], np.float32), (2, 3, 1))])
    uv = np.concatenate([uv, np.zeros((2, 3, 2), np.float32)]); mi = np.concatenate([mi, np.full(2, 1, np.uint32)])
    pos, nrm, uv, mi = _drop_degenerate(pos, nrm, uv, mi)
    mats = [material(_pkg.BSDF_DIFFUSE, 0), material(_pkg.BSDF_DIFFUSE, 1, emission=(20, 18, 15))]
    return SceneArrays(pos, nrm, uv, mi, mats, [checker(256, 32), WHITE.copy()])


def terrain_camera():
    return dict(frm=(0.0, 5.0, 14.0), to=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=45.0)


def deep_chains(k=55, ratio=2.0, arms=3):
    """Small triangles at geometrically growing distances along the three axes. Every SAH split peels
    a few of the farthest off one arm, so the reference's tree is a caterpillar about 40 levels deep
    (k=55, ratio 2: binary depth 40, 4-wide depth 14) from only 165 triangles: the scene that
    exercises the deepest traversal stacks."""
    g = float(ratio) ** np.arange(1, k + 1)
    one = np.ones(k)
    c = np.concatenate([np.stack([g, one, one], 1), np.stack([one, g, one], 1), np.stack([one, one, g], 1)][:arms])
    pos = np.repeat(c[:, None, :], 3, 1).astype(np.float32)
    pos[:, 1, 0] += 0.25; pos[:, 2, 1] += 0.25; pos[:, 2, 2] += 0.25
    uv = np.zeros((len(pos), 3, 2), np.float32)
    nrm = np.repeat(_face_normals(pos)[:, None, :], 3, axis=1)
    return SceneArrays(pos, nrm, uv, np.zeros(len(pos), np.uint32), [material(_pkg.BSDF_DIFFUSE, 0)], [WHITE * 0.8])


def transformed(scene, scale=1.0, offset=(0.0, 0.0, 0.0)):
    """The same scene with every vertex at p * scale + offset (fp32, so large offsets quantise the
    geometry; degenerate results are dropped like the loader does)."""
    pos = (scene.positions * np.float32(scale) + np.asarray(offset, np.float32)).astype(np.float32)
    pos, nrm, uv, mi = _drop_degenerate(pos, scene.normals, scene.uvs, scene.material_index)
    bg = tuple(scene.desc.background)
    return SceneArrays(pos, nrm, uv, mi, scene.materials, scene.textures, env=scene.env, background=bg)


def transformed_camera(cam_kw, scale=1.0, offset=(0.0, 0.0, 0.0)):
    o = np.asarray(offset, np.float64)
    out = dict(cam_kw)
    out["frm"] = tuple(float(v) for v in np.asarray(cam_kw["frm"], np.float64) * scale + o)
    out["to"] = tuple(float(v) for v in np.asarray(cam_kw["to"], np.float64) * scale + o)
    return out
