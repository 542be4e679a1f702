"""Writer for the reference's scene directories (scene.json + .gem meshes + PNG / Radiance HDR
textures) and reader for .rtxs dumps.

The reference's own assets are a separate download that is not available offline, so synthetic
scenes are emitted in the same on-disk format and go through the same loader path:

  scene.json   RTBase/GEMLoader.h:685-738 (scene properties + instance arrays), camera keys of
               RTBase/SceneLoader.h:342-354, material keys of RTBase/SceneLoader.h:212-288
  *.gem        RTBase/GEMLoader.h:344-365: magic 4058972161, animated flag, meshes; 44-byte
               vertices (position, normal, tangent, u, v), u32 indices
  textures     8-bit PNG (albedo is quantised to k/255 — what any scene in this format carries),
               Radiance RGBE for the environment map

`quantised(scene)` returns the SceneArrays a loader must reproduce from such a directory.
"""
import json
import os
import struct
import zlib

import numpy as np

from . import (BSDF_CONDUCTOR, BSDF_DIELECTRIC, BSDF_DIFFUSE, BSDF_GLASS, BSDF_LAYERED, BSDF_MIRROR, BSDF_ORENNAYAR,
               BSDF_PLASTIC, Material, SceneArrays)

GEM_MAGIC = 4058972161


def _png(path, rgb8):
    """8-bit RGB PNG, scanline filter chosen per row to exercise all five filter types."""
    h, w, _ = rgb8.shape
    raw = bytearray()
    prev = np.zeros((w * 3,), np.int32)
    for y in range(h):
        cur = rgb8[y].reshape(-1).astype(np.int32)
        left = np.concatenate([np.zeros(3, np.int32), cur[:-3]])
        upleft = np.concatenate([np.zeros(3, np.int32), prev[:-3]])
        ft = y % 5
        if ft == 0:
            out = cur
        elif ft == 1:
            out = cur - left
        elif ft == 2:
            out = cur - prev
        elif ft == 3:
            out = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            out = cur - pred
        raw.append(ft)
        raw += (out & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(kind, data):
        c = struct.pack(">I", len(data)) + kind + data
        return c + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        comp = zlib.compress(bytes(raw), 6)
        half = len(comp) // 2                       # two IDAT chunks: decoders must concatenate them
        f.write(chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b""))


def _rgbe(rgb):
    """float RGB -> RGBE bytes and the floats a decoder (mantissa * 2^(e-136)) gets back."""
    rgb = np.asarray(rgb, np.float32)
    m = rgb.max(axis=-1)
    e = np.zeros(m.shape, np.int32)
    nz = m > 1e-32
    mant, ex = np.frexp(m[nz])
    scale = np.zeros(m.shape, np.float32)
    scale[nz] = (mant * 256.0 / m[nz]).astype(np.float32)
    e[nz] = ex + 128
    q = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    q[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    q[..., 3] = np.where(nz, e, 0).astype(np.uint8)
    back = q[..., :3].astype(np.float32) * np.ldexp(np.float32(1.0), q[..., 3].astype(np.int32) - 136)[..., None].astype(np.float32)
    back[q[..., 3] == 0] = 0
    return q, back.astype(np.float32)


def _hdr(path, rgb, rle):
    q, _ = _rgbe(rgb)
    h, w, _ = q.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode())
        if not rle or w < 8 or w >= 32768:
            f.write(q.tobytes())
            return
        for y in range(h):
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for c in range(4):
                row = q[y, :, c]
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 4:
                        f.write(bytes([128 + run, int(row[x])]))
                        x += run
                    else:
                        # literal block: up to 128 bytes, ending where a run of four equal bytes begins
                        k = 1
                        while k < 128 and x + k < w:
                            j = x + k
                            if j + 3 < w and row[j] == row[j + 1] == row[j + 2] == row[j + 3]:
                                break
                            k += 1
                        f.write(bytes([k]) + row[x:x + k].tobytes())
                        x += k


_BSDF_NAMES = {BSDF_DIFFUSE: "diffuse", BSDF_MIRROR: "mirror", BSDF_CONDUCTOR: "conductor", BSDF_GLASS: "glass",
               BSDF_PLASTIC: "plastic", BSDF_DIELECTRIC: "dielectric", BSDF_ORENNAYAR: "orennayar"}


def _tex8(t):
    return np.clip(np.rint(np.asarray(t, np.float32) * 255.0), 0, 255).astype(np.uint8)


def quantised(scene: SceneArrays) -> SceneArrays:
    """The scene as it comes back from a scene directory: 8-bit albedo textures, RGBE
    environment map, LayeredBSDF wrappers gone (the loader never hands them to the scene,
    RTBase/SceneLoader.h:279-295)."""
    tex = [(_tex8(t).astype(np.float32) / np.float32(255.0)).astype(np.float32) for t in scene.textures]
    env = None if scene.env is None else _rgbe(scene.env)[1]
    mats = [Material.from_buffer_copy(bytes(m)) for m in scene.materials]
    for m in mats:
        m.type &= ~BSDF_LAYERED
    return SceneArrays(scene.positions, scene.normals, scene.uvs, scene.material_index, mats, tex, env=env,
                       background=tuple(scene.desc.background))


def save_gem(scene: SceneArrays, directory, frm, to, up, fov, width, height, world=None, rle_env=True, flip_x=0):
    """Write `scene` as a scene directory. One instance per material; `world` (4x4, row-major)
    is given to every instance and its inverse applied to the stored vertices, so that a loader
    has to apply the instance transform (positions then match to rounding, not bit for bit)."""
    os.makedirs(directory, exist_ok=True)
    W = np.eye(4, dtype=np.float64) if world is None else np.asarray(world, np.float64).reshape(4, 4)
    Winv = np.linalg.inv(W)
    for i, t in enumerate(scene.textures):
        _png(os.path.join(directory, f"tex{i}.png"), _tex8(t))
    top = {"width": width, "height": height, "fov": float(fov), "from": " ".join(repr(float(v)) for v in frm),
           "to": " ".join(repr(float(v)) for v in to), "up": " ".join(repr(float(v)) for v in up)}
    if flip_x:
        top["flipX"] = 1
    if scene.env is not None:
        _hdr(os.path.join(directory, "env.hdr"), scene.env, rle_env)
        top["envmap"] = "env.hdr"
    instances = []
    for mi, m in enumerate(scene.materials):
        sel = np.nonzero(scene.material_index == mi)[0]
        base = m.type & ~BSDF_LAYERED
        inst = {"filename": f"mesh{mi}.gem", "world": [float(v) for v in W.reshape(-1)], "bsdf": _BSDF_NAMES[base],
                "reflectance": f"tex{m.texture}.png"}
        # numbers are written as text: a JSON number reaches the loader through std::to_string(float),
        # six decimals (RTBase/GEMLoader.h:467-470), a string is parsed as it stands
        if base in (BSDF_GLASS, BSDF_PLASTIC, BSDF_DIELECTRIC):
            inst["intIOR"] = repr(float(m.int_ior)); inst["extIOR"] = repr(float(m.ext_ior))
        if base in (BSDF_PLASTIC, BSDF_DIELECTRIC, BSDF_CONDUCTOR):
            inst["roughness"] = repr(float(m.roughness))
        if base == BSDF_ORENNAYAR:
            inst["alpha"] = repr(float(m.roughness))
        if base == BSDF_CONDUCTOR:
            inst["eta"] = " ".join(repr(float(v)) for v in m.eta); inst["k"] = " ".join(repr(float(v)) for v in m.k)
        if any(float(v) != 0 for v in m.emission):
            inst["emission"] = " ".join(repr(float(v)) for v in m.emission)
        if m.type & BSDF_LAYERED:
            inst["coatingThickness"] = "1.0"; inst["coatingSigmaA"] = "0.1 0.1 0.1"
        instances.append(inst)
        # vertices: one per triangle corner (no sharing), stored in the instance's object space
        p = scene.positions[sel].reshape(-1, 3).astype(np.float64)
        n = scene.normals[sel].reshape(-1, 3).astype(np.float64)
        p_obj = (p @ Winv[:3, :3].T + Winv[:3, 3]).astype(np.float32)
        n_obj = (n @ W[:3, :3]).astype(np.float32)          # loader applies (W^-1)^T: undo with W^T, i.e. n @ W
        uv = scene.uvs[sel].reshape(-1, 2).astype(np.float32)
        rec = np.zeros((p_obj.shape[0], 11), np.float32)
        rec[:, 0:3] = p_obj; rec[:, 3:6] = n_obj; rec[:, 6:9] = (1.0, 0.0, 0.0); rec[:, 9:11] = uv
        idx = np.arange(p_obj.shape[0], dtype=np.uint32)
        # two meshes per file when there is enough to split: exercises the index offsetting of the loader
        cut = (len(sel) // 2) * 3 if len(sel) >= 2 else p_obj.shape[0]
        parts = [(rec[:cut], idx[:cut])] + ([(rec[cut:], idx[cut:] - cut)] if cut < p_obj.shape[0] else [])
        with open(os.path.join(directory, f"mesh{mi}.gem"), "wb") as f:
            f.write(struct.pack("<III", GEM_MAGIC, 0, len(parts)))
            for r, ix in parts:
                props = [("name", f"mesh{mi}")]
                f.write(struct.pack("<I", len(props)))
                for k, v in props:
                    f.write(struct.pack("<i", len(k)) + k.encode() + struct.pack("<i", len(v)) + v.encode())
                f.write(struct.pack("<I", r.shape[0]) + r.tobytes())
                f.write(struct.pack("<I", ix.shape[0]) + ix.astype(np.uint32).tobytes())
    top["instances"] = instances
    with open(os.path.join(directory, "scene.json"), "w") as f:
        json.dump(top, f, indent=1)


def load_rtxs(path):
    """Read an .rtxs dump: returns (SceneArrays, camera dict)."""
    with open(path, "rb") as f:
        d = f.read()
    assert d[:4] == b"RTXS"
    version = struct.unpack_from("<I", d, 4)[0]
    o = 8
    cam = struct.unpack_from("<10f2I", d, o); o += 48
    flip = 0
    if version == 2:
        flip = struct.unpack_from("<I", d, o)[0]; o += 4
    ntex = struct.unpack_from("<I", d, o)[0]; o += 4
    tex = []
    for _ in range(ntex):
        w, h = struct.unpack_from("<2I", d, o); o += 8
        tex.append(np.frombuffer(d, np.float32, w * h * 3, o).reshape(h, w, 3).copy()); o += w * h * 12
    nmat = struct.unpack_from("<I", d, o)[0]; o += 4
    import ctypes
    ms = ctypes.sizeof(Material)
    mats = [Material.from_buffer_copy(d[o + i * ms:o + (i + 1) * ms]) for i in range(nmat)]; o += nmat * ms
    ntri = struct.unpack_from("<Q", d, o)[0]; o += 8
    pos = np.frombuffer(d, np.float32, ntri * 9, o).reshape(ntri, 3, 3).copy(); o += ntri * 36
    nrm = np.frombuffer(d, np.float32, ntri * 9, o).reshape(ntri, 3, 3).copy(); o += ntri * 36
    uv = np.frombuffer(d, np.float32, ntri * 6, o).reshape(ntri, 3, 2).copy(); o += ntri * 24
    mi = np.frombuffer(d, np.uint32, ntri, o).copy(); o += ntri * 4
    ew, eh = struct.unpack_from("<2I", d, o); o += 8
    env = None
    if ew and eh:
        env = np.frombuffer(d, np.float32, ew * eh * 3, o).reshape(eh, ew, 3).copy(); o += ew * eh * 12
    bg = struct.unpack_from("<3f", d, o)
    camera = dict(frm=cam[0:3], to=cam[3:6], up=cam[6:9], fov=cam[9], width=cam[10], height=cam[11], flip_x=flip)
    return SceneArrays(pos, nrm, uv, mi, mats, tex, env=env, background=bg), camera
