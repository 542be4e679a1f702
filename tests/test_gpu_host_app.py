"""The C++ host mirror (ray-tracer_amd/host: RayTracer / Scene / Camera / Film over the C
ABI) driven by the headless main with the reference's flags must produce the same film as
the Python binding over the same library, and write the HDR / PNG the reference writes."""
import importlib
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "ray-tracer_amd", "host", "rt_headless")


def test_headless_driver_matches_python_binding(tmp_path, pkg, lib, core, scenes):
    assert os.path.exists(APP), "host app not built (run __graft_entry__.build())"
    sc, kw = scenes.cornell_box("mixed"), scenes.cornell_camera()
    w, h, spp = 96, 64, 12
    path = str(tmp_path / "box.rtxs")
    sc.save(path, kw["frm"], kw["to"], kw["up"], kw["fov"], w, h)
    hdr, png, dump = (str(tmp_path / n) for n in ("out.hdr", "out.png", "film.bin"))
    out = subprocess.run([APP, "-scene", path, "-outputFilename", hdr, "-SPP", str(spp), "-png", png, "-dumpFilm", dump,
                          "-adaptive", "1", "-initSPP", "4", "-seed", "5", "-quiet", "1"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "Loading scene" in out.stdout and f"Done: {spp} spp" in out.stdout
    raw = open(dump, "rb").read()
    fw, fh, fspp = struct.unpack("<3I", raw[:12])
    assert (fw, fh, fspp) == (w, h, spp)
    film = np.frombuffer(raw, np.float32, w * h * 3, 12).reshape(h, w, 3)
    tiles = np.frombuffer(raw, np.uint32, -1, 12 + w * h * 12)
    core.upload_scene(sc)
    core.set_camera(pkg.look_at(lib, width=w, height=h, **kw))
    core.set_settings(pkg.make_settings(total_spp=spp, init_spp=4, adaptive_sampling=1, seed=5, num_threads=20))
    core.clear()
    core.render_passes(spp)
    f2, s2, t2 = core.film()
    assert np.array_equal(film, f2) and np.array_equal(tiles, t2) and s2 == spp
    # outputs: Radiance header + one RGBE quad per pixel; PNG signature + IHDR size
    b = open(hdr, "rb").read()
    assert b.startswith(b"#?RADIANCE") and f"-Y {h} +X {w}".encode() in b[:80] and len(b) - b.index(b"\n", b.index(b"-Y")) - 1 == w * h * 4
    p = open(png, "rb").read()
    assert p[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">2I", p[16:24]) == (w, h)


def test_png_is_decodable(tmp_path, scenes):
    """zlib accepts the stored-block stream and the rows carry the tone-mapped pixels."""
    import zlib
    sc, kw = scenes.cornell_box(), scenes.cornell_camera()
    path = str(tmp_path / "b.rtxs")
    sc.save(path, kw["frm"], kw["to"], kw["up"], kw["fov"], 40, 30)
    png = str(tmp_path / "o.png")
    subprocess.run([APP, "-scene", path, "-outputFilename", str(tmp_path / "o.hdr"), "-SPP", "4", "-png", png, "-quiet", "1"], check=True, capture_output=True, timeout=120)
    p = open(png, "rb").read()
    i = p.index(b"IDAT")
    n = struct.unpack(">I", p[i - 4:i])[0]
    rows = zlib.decompress(p[i + 4:i + 4 + n])
    assert len(rows) == 30 * (40 * 3 + 1) and max(rows) > 50


def test_gem_scene_directory_renders_like_its_dump(tmp_path, pkg, lib, core, scenes):
    """A scene directory in the reference's format (scene.json + .gem + PNG + HDR env) rendered
    by the headless driver gives the film the Python binding gives for the same, quantised,
    scene: the loader path and the direct upload are one and the same scene."""
    gemio = importlib.import_module("ray-tracer_amd.gemio")
    sc, kw = scenes.mixed_scene(16, env_size=(64, 32)), scenes.mixed_camera()
    w, h, spp = 112, 64, 6
    d = str(tmp_path / "scene")
    gemio.save_gem(sc, d, kw["frm"], kw["to"], kw["up"], kw["fov"], w, h)
    dump = str(tmp_path / "film.bin")
    out = subprocess.run([APP, "-scene", d, "-outputFilename", str(tmp_path / "o.hdr"), "-SPP", str(spp), "-dumpFilm", dump,
                          "-adaptive", "0", "-seed", "9", "-quiet", "1"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = open(dump, "rb").read()
    assert struct.unpack("<3I", raw[:12]) == (w, h, spp)
    film = np.frombuffer(raw, np.float32, w * h * 3, 12).reshape(h, w, 3)
    # same scene through the binding: triangles grouped by material as the loader emits them
    q = gemio.quantised(sc)
    order = np.concatenate([np.nonzero(sc.material_index == mi)[0] for mi in range(len(sc.materials))])
    ref = pkg.SceneArrays(q.positions[order], q.normals[order], q.uvs[order], q.material_index[order], q.materials, q.textures, env=q.env)
    core.upload_scene(ref)
    core.set_camera(pkg.look_at(lib, width=w, height=h, **kw))
    core.set_settings(pkg.make_settings(total_spp=spp, init_spp=spp, adaptive_sampling=0, seed=9, num_threads=20))
    core.clear()
    core.render_passes(spp)
    f2, s2, _ = core.film()
    assert s2 == spp and (f2 > 0).any()
    # normals are re-normalised by the loader (last-bit differences in shading normals)
    err = np.abs(film - f2) / np.maximum(np.abs(f2), 1e-2)
    assert np.mean((err > 1e-3).any(axis=-1)) < 0.01


def test_fly_through_camera_yaw(tmp_path, pkg, lib, core, scenes):
    """RTCamera::yaw + updateCamera (SceneLoader.h:68-83, 115-120) re-aim the device camera:
    the film after -orbit equals a render with the rotated look-at point."""
    sc, kw = scenes.cornell_box("diffuse"), scenes.cornell_camera()
    w, h, spp, deg = 64, 48, 4, 20.0
    path = str(tmp_path / "box.rtxs")
    sc.save(path, kw["frm"], kw["to"], kw["up"], kw["fov"], w, h)
    dump = str(tmp_path / "film.bin")
    out = subprocess.run([APP, "-scene", path, "-outputFilename", str(tmp_path / "o.hdr"), "-SPP", str(spp), "-dumpFilm", dump, "-adaptive", "0",
                          "-seed", "3", "-orbit", str(deg), "-quiet", "1"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    film = np.frombuffer(open(dump, "rb").read(), np.float32, w * h * 3, 12).reshape(h, w, 3)
    frm, to, up = (np.array(kw[k], np.float32) for k in ("frm", "to", "up"))
    d = (to - frm) / np.linalg.norm(to - frm); k = up / np.linalg.norm(up)
    rad = np.float32(deg * (np.pi / 180.0)); c, s = np.cos(rad), np.sin(rad)
    rot = d * c + np.cross(k, d) * s + k * (np.dot(k, d) * (1 - c))
    core.upload_scene(sc)
    core.set_camera(pkg.look_at(lib, tuple(frm), tuple((frm + rot).astype(np.float32)), tuple(up), kw["fov"], w, h))
    core.set_settings(pkg.make_settings(total_spp=spp, init_spp=spp, seed=3, num_threads=20))
    core.clear(); core.render_passes(spp)
    f2, _, _ = core.film()
    assert not np.allclose(f2, 0)
    err = np.abs(film - f2) / np.maximum(np.abs(f2), 1e-2)
    assert np.mean((err > 1e-3).any(axis=-1)) < 0.02


def test_a_rank_that_dies_takes_the_others_with_it(tmp_path, scenes):
    """rt_headless -gpus 2 on a box with ONE GPU: rank 1 finds no device and exits with an error while rank 0 waits for
    it in the RCCL rendezvous. The parent must end rank 0 and report the failure instead of waiting for ever."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box with exactly one GPU")
    sc, kw = scenes.cornell_box(), scenes.cornell_camera()
    path = str(tmp_path / "b.rtxs")
    sc.save(path, kw["frm"], kw["to"], kw["up"], kw["fov"], 32, 24)
    p = subprocess.Popen([APP, "-scene", path, "-outputFilename", str(tmp_path / "o.hdr"), "-SPP", "2", "-gpus", "2", "-quiet", "1"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=90)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, 9)
        p.communicate()
        pytest.fail("rt_headless -gpus 2 kept waiting after one of its ranks had failed")
    assert p.returncode != 0, (out, err)
