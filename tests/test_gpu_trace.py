"""GPU parity of the ray-query entry points (rtx_trace_closest / rtx_trace_visible) against
the CPU oracle: hit id, t and barycentrics must agree BIT FOR BIT, visibility exactly.

Ray sets follow what a BVH traversal can get wrong: coherent primary rays, incoherent
rays, rays with zero direction components (the reference's slab test produces NaNs there,
RTBase/Geometry.h:213-228), rays starting on surfaces, coplanar duplicate triangles
(equal-t ties, RTBase/Geometry.h:535) and leaves that cannot be split.
"""
import numpy as np
import pytest

from conftest import pixel_centres

pytestmark = pytest.mark.gpu


def _rand_rays(pkg, n, seed, extent=1.2, centre=(0.0, 0.0, 0.0)):
    rs = np.random.RandomState(seed)
    rays = np.empty(n, pkg.RAY_DTYPE)
    rays["o"] = (rs.uniform(-extent, extent, (n, 3)) + np.asarray(centre)).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32)
    rays["d"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return rays


def _axis_rays(pkg, n, seed, extent=1.2, snap=None, centre=(0.0, 0.0, 0.0)):
    """Directions with one or two exactly-zero components; origins optionally snapped to
    coordinates that occur in the scene (box faces)."""
    rs = np.random.RandomState(seed)
    rays = np.empty(n, pkg.RAY_DTYPE)
    o = (rs.uniform(-extent, extent, (n, 3)) + np.asarray(centre)).astype(np.float32)
    if snap is not None:
        pick = rs.randint(0, len(snap), (n, 3))
        mask = rs.rand(n, 3) < 0.5
        o = np.where(mask, snap[pick].astype(np.float32), o)
    d = rs.normal(size=(n, 3)).astype(np.float32)
    zero = rs.rand(n, 3) < 0.45
    zero[zero.all(1)] = [True, False, True]
    d[zero] = 0.0
    neg = rs.rand(n, 3) < 0.5
    d = np.where(zero & neg, np.float32(-0.0), d)
    nz = np.linalg.norm(d, axis=1, keepdims=True)
    nz[nz == 0] = 1
    rays["o"], rays["d"] = o, (d / nz).astype(np.float32)
    return rays


def _assert_hits_equal(got, want):
    assert np.array_equal(got["id"], want["id"]), f"{(got['id'] != want['id']).sum()} hit ids differ"
    for f in ("t", "alpha", "beta", "gamma"):
        hit = want["id"] != 0xFFFFFFFF
        assert np.array_equal(got[f][hit].view(np.uint32), want[f][hit].view(np.uint32)), f"{f} differs"
    assert np.all(got["t"][~hit] == np.finfo(np.float32).max)


def _check_scene(pkg, lib, core, orc, scene, cam_kw, w=160, h=120, n_rand=40000, snap=None, extent=1.2, centre=(0.0, 0.0, 0.0)):
    """Primary rays, random rays and rays with zero direction components from a box of half-width
    `extent` around `centre`; secondary rays from the hit points (origins on surfaces, and pushed off
    them by EPSILON along the new direction the way the integrators do, RTBase/Renderer.h:513);
    visibility segments between hit points and to random points."""
    core.upload_scene(scene)
    cam = pkg.look_at(lib, width=w, height=h, **cam_kw)
    core.set_camera(cam)
    osc = orc.scene(scene)
    osc.set_camera(cam)
    rays = np.concatenate([osc.generate_rays(pixel_centres(w, h)), _rand_rays(pkg, n_rand, 1, extent, centre),
                           _axis_rays(pkg, n_rand // 2, 2, extent, snap, centre)])
    want = osc.trace_closest(rays)
    got = core.trace_closest(rays)
    _assert_hits_equal(got, want)
    # secondary rays leaving the hit points (origins on surfaces)
    hit = want["id"] != 0xFFFFFFFF
    o2 = rays["o"][hit] + rays["d"][hit] * want["t"][hit][:, None]
    sec = _rand_rays(pkg, int(hit.sum()), 3)
    sec["o"] = o2.astype(np.float32)
    _assert_hits_equal(core.trace_closest(sec), osc.trace_closest(sec))
    sec["o"] = (o2.astype(np.float32) + sec["d"] * np.float32(0.001)).astype(np.float32)
    _assert_hits_equal(core.trace_closest(sec), osc.trace_closest(sec))
    # visibility: hit point -> random point, hit point -> hit point
    rs = np.random.RandomState(4)
    p2 = np.concatenate([(rs.uniform(-1.25 * extent, 1.25 * extent, (len(o2) // 2, 3)) + np.asarray(centre)).astype(np.float32),
                         np.roll(o2, 7, axis=0)[: len(o2) - len(o2) // 2]])
    seg = np.concatenate([o2.astype(np.float32), p2.astype(np.float32)], 1)
    assert np.array_equal(core.trace_visible(seg), osc.trace_visible(seg))
    osc.close()
    return int(hit.sum())


def test_reference_unit_test_rays_through_the_device_root_box_test(pkg, lib, core, orc):
    """The reference's own known answers (UnitTests/test.cpp:3-34: AABB [1,3]^3; rays (0,0,0)->(1,1,1) hit,
    (0,0,0)->(-1,-1,-1) miss, (2,2,2)->(1,1,1) hit) through rtx_trace_closest: the scene is that box as 12
    triangles, so the scene's root box IS the test's AABB and the device's restatement of AABB::rayAABB decides
    first. Directions are not normalised, as in the reference's test."""
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "aabb_kat.json")))
    lo, hi = (1.0, 1.0, 1.0), (3.0, 3.0, 3.0)
    c = [(x, y, z) for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])]
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    pos = np.array([[c[a], c[b], c[d]] for a, b, d, e in quads] + [[c[a], c[d], c[e]] for a, b, d, e in quads], np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (12, 3, 1)); uv = np.zeros((12, 3, 2), np.float32)
    sc = pkg.SceneArrays(pos, nrm, uv, np.zeros(12, np.uint32), [pkg.material()], [np.ones((1, 1, 3), np.float32)])
    core.upload_scene(sc)
    b = core.bvh()[0]
    assert np.array_equal(b[0], [1, 1, 1, 3, 3, 3])
    rays = np.empty(len(kat["cases"]), pkg.RAY_DTYPE)
    rays["o"] = [k["origin"] for k in kat["cases"]]; rays["d"] = [k["dir"] for k in kat["cases"]]
    hits = core.trace_closest(rays)
    for k, h in zip(kat["cases"], hits):
        assert (h["id"] != 0xFFFFFFFF) == k["hit"], k["name"]
    assert hits["t"][0] == 1.0 and hits["t"][2] == 1.0      # the corners (1,1,1) and (3,3,3) along (1,1,1)
    osc = orc.scene(sc)
    _assert_hits_equal(hits, osc.trace_closest(rays))
    more = _rand_rays(pkg, 20000, 13, extent=4.0, centre=(2, 2, 2))
    _assert_hits_equal(core.trace_closest(more), osc.trace_closest(more))
    osc.close()


def test_closest_and_visible_soup(pkg, lib, core, orc, scenes):
    n = _check_scene(pkg, lib, core, orc, scenes.soup(100_000, seed=11), dict(frm=(0, 0, 3.5), to=(0, 0, 0), up=(0, 1, 0), fov=45.0))
    assert n > 10000


def test_closest_and_visible_cornell_axis_aligned(pkg, lib, core, orc, scenes):
    snap = np.array([-1, 1, 0.999, -0.7, -0.1, 0.2, 0.0, 0.1, 0.7, -0.4, 0.6, -0.25, 0.25, -0.6], np.float32)
    _check_scene(pkg, lib, core, orc, scenes.cornell_box("mixed"), scenes.cornell_camera(), snap=snap)


def test_closest_mixed_scene_smooth_meshes(pkg, lib, core, orc, scenes):
    _check_scene(pkg, lib, core, orc, scenes.mixed_scene(48, (64, 32)), scenes.mixed_camera())


def test_ties_and_unsplittable_leaves(pkg, lib, core, orc, scenes):
    """40 identical copies of a few triangles: every copy returns the same t, the reference
    keeps the first one it visits; their shared centre makes the node unsplittable, so the
    leaf holds more triangles than a device leaf slot."""
    base = scenes.soup(200, seed=5, size=0.2)
    pos = np.tile(base.positions[:6], (40, 1, 1))
    extra = scenes.soup(3000, seed=6, size=0.05)
    P = np.concatenate([pos, extra.positions]); N = np.concatenate([np.tile(base.normals[:6], (40, 1, 1)), extra.normals])
    U = np.concatenate([np.tile(base.uvs[:6], (40, 1, 1)), extra.uvs])
    sc = pkg.SceneArrays(P, N, U, np.zeros(len(P), np.uint32), base.materials, base.textures)
    core.upload_scene(sc)
    osc = orc.scene(sc)
    rays = _rand_rays(pkg, 60000, 9)
    # aim half of the rays at the duplicated triangles
    tgt = base.positions[:6].mean(1)[np.random.RandomState(1).randint(0, 6, 30000)]
    d = tgt - rays["o"][:30000]
    rays["d"][:30000] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    want = osc.trace_closest(rays)
    assert (want["id"][:30000] < 240).sum() > 5000
    _assert_hits_equal(core.trace_closest(rays), want)
    osc.close()


def test_empty_and_tiny_inputs(pkg, lib, core, orc, scenes):
    sc = scenes.soup(3, seed=2, size=0.5)
    core.upload_scene(sc)
    osc = orc.scene(sc)
    assert core.trace_closest(np.empty(0, pkg.RAY_DTYPE)).shape == (0,)
    rays = _rand_rays(pkg, 1000, 5)
    _assert_hits_equal(core.trace_closest(rays), osc.trace_closest(rays))
    assert core.trace_visible(np.empty((0, 6), np.float32)).shape == (0,)
    osc.close()


def test_bvh_matches_oracle_on_device_context(pkg, lib, core, orc, scenes):
    sc = scenes.soup(30_000, seed=21)
    core.upload_scene(sc)
    b, t, i, d = core.bvh()
    ob, ot, oi, od = orc.scene(sc).bvh()
    assert d == od and np.array_equal(t, ot) and np.array_equal(i, oi) and np.array_equal(b, ob)


def test_full_size_1m_soup_1080p_bit_exact(pkg, lib, core, orc, scenes):
    """The benchmark configuration itself: 1 M-triangle soup, 1920x1080 pixel-centre rays."""
    sc = scenes.soup(1_000_000, seed=12345)
    core.upload_scene(sc)
    cam = pkg.look_at(lib, (0, 0, 3.5), (0, 0, 0), (0, 1, 0), 45.0, 1920, 1080)
    core.set_camera(cam)
    osc = orc.scene(sc)
    osc.set_camera(cam)
    rays = osc.generate_rays(pixel_centres(1920, 1080))
    want = osc.trace_closest(rays, threads=16)
    got = core.trace_closest(rays)
    _assert_hits_equal(got, want)
    assert (want["id"] != 0xFFFFFFFF).sum() > 500_000
    osc.close()


@pytest.mark.parametrize("which", ["soup", "cornell", "mixed", "ties", "terrain"])
def test_device_bvh_build_equals_host_build(pkg, lib, core, scenes, which):
    """The level-by-level GPU build (binning with atomics, createSplit's swap order through
    pointer jumping) must give the tree of the host build: same numbering, bounds, index order."""
    if which == "soup":
        sc = scenes.soup(300_000, seed=3)
    elif which == "cornell":
        sc = scenes.cornell_box("mixed")
    elif which == "mixed":
        sc = scenes.mixed_scene(48, env_size=(32, 16))
    elif which == "terrain":
        sc = scenes.terrain(300)
    else:
        base = scenes.soup(200, seed=5, size=0.2)
        extra = scenes.soup(3000, seed=6, size=0.05)
        P = np.concatenate([np.tile(base.positions[:6], (40, 1, 1)), extra.positions])
        N = np.concatenate([np.tile(base.normals[:6], (40, 1, 1)), extra.normals])
        U = np.concatenate([np.tile(base.uvs[:6], (40, 1, 1)), extra.uvs])
        sc = pkg.SceneArrays(P, N, U, np.zeros(len(P), np.uint32), base.materials, base.textures)
    core.upload_scene(sc)                     # device build (the default)
    b, t, i, d = core.bvh()
    hb, ht, hi, hd = pkg.host_bvh(lib, sc)[:4]
    assert d == hd
    assert np.array_equal(t, ht), "topology / ranges differ"
    assert np.array_equal(i, hi), "index permutation differs"
    assert np.array_equal(b, hb), "bounds differ"


def test_ragged_batch_sizes_of_the_dynamic_fetch(pkg, lib, core, orc, scenes):
    """Ray counts around the chunk (64), counter (64 x 4 chunks) and grid boundaries: every ray
    is traced exactly once whatever the remainder."""
    sc = scenes.soup(5000, seed=31, size=0.05)
    core.upload_scene(sc)
    osc = orc.scene(sc)
    big = _rand_rays(pkg, 1_000_003, 77)
    want = osc.trace_closest(big, threads=8)
    seg = np.random.RandomState(5).uniform(-1.1, 1.1, (300_007, 6)).astype(np.float32)
    vis = osc.trace_visible(seg, threads=8)
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 4097, 16384 + 1, 262_144 + 63, 1_000_003):
        _assert_hits_equal(core.trace_closest(big[:n]), want[:n])
    for n in (1, 63, 65, 257, 300_007):
        assert np.array_equal(core.trace_visible(seg[:n]), vis[:n])
    osc.close()


@pytest.mark.parametrize("fine", [None, "0", "48", "4096"], ids=["default", "chunks_only", "48_per_wave", "everything_item_by_item"])
def test_hand_out_boundaries(pkg, lib, lib_tuning, orc, scenes, fine, monkeypatch):
    """The work hand-out changes its way with the size of a launch (5 120 waves on the device): below 32 rays per
    wave it is spread thin, the last 192 rays per wave (RTX_FINE overrides) go out item by item and whole chunks of
    64 before them. Sizes on both sides of every boundary, for several shares of the item-by-item part: each ray is
    traced exactly once (hits equal the oracle's, ray by ray)."""
    if fine is None: monkeypatch.delenv("RTX_FINE", raising=False)
    else: monkeypatch.setenv("RTX_FINE", fine); lib = lib_tuning        # only the tuning build reads the switch
    c = pkg.Core(0, lib)                      # the switch is read when a context is created
    sc = scenes.soup(5000, seed=37, size=0.05)
    c.upload_scene(sc)
    osc = orc.scene(sc)
    big = _rand_rays(pkg, 1_048_576 + 77, 79)
    want = osc.trace_closest(big, threads=8)
    per_wave = 192 if fine is None else int(fine)
    sizes = {1, 64, 20_479, 20_481, 163_839, 163_840, 163_841, 300_000, 1_048_576, 1_048_576 + 77}
    for edge in (5120 * per_wave, 5120 * per_wave + 64 * 64):      # where whole chunks start to exist / every counter has one
        sizes |= {n for n in (edge - 1, edge, edge + 1, edge + 65) if 0 < n <= len(big)}
    for n in sorted(sizes):
        _assert_hits_equal(c.trace_closest(big[:n]), want[:n])
    seg = np.random.RandomState(6).uniform(-1.1, 1.1, (400_003, 6)).astype(np.float32)
    vis = osc.trace_visible(seg, threads=8)
    for n in (163_841, 400_003):
        assert np.array_equal(c.trace_visible(seg[:n]), vis[:n])
    osc.close(); c.close()


@pytest.mark.parametrize("seed,n_tris,size", [(101, 30_000, 0.02), (102, 8_000, 0.3), (103, 60_000, 0.05), (104, 500, 1.0)])
def test_randomised_soups_closest_and_visible(pkg, lib, core, orc, scenes, seed, n_tris, size):
    """Different triangle sizes change how heavy-tailed the per-ray cost is (large triangles: deep
    overlapping boxes, long rays beside short ones in one wave), which is what the work sharing at
    the end of a launch reacts to; hits and visibility must not depend on it."""
    sc = scenes.soup(n_tris, seed=seed, size=size)
    core.upload_scene(sc)
    osc = orc.scene(sc)
    rays = np.concatenate([_rand_rays(pkg, 50_000, seed), _axis_rays(pkg, 6_000, seed + 1)])
    want = osc.trace_closest(rays, threads=8)
    _assert_hits_equal(core.trace_closest(rays), want)
    for n in (17, 1000, 20_001):       # launches that are all tail
        _assert_hits_equal(core.trace_closest(rays[:n]), want[:n])
    seg = np.random.RandomState(seed).uniform(-1.3, 1.3, (40_000, 6)).astype(np.float32)
    vis = osc.trace_visible(seg, threads=8)
    assert np.array_equal(core.trace_visible(seg), vis)
    assert np.array_equal(core.trace_visible(seg[:333]), vis[:333])
    osc.close()


def test_closest_packets_of_the_tuning_build(pkg, lib_tuning, orc, scenes, monkeypatch):
    """The tuning build can walk coherent runs of 64 rays as one packet per wave (k_closest_packet; measured slower
    than the per-lane kernel at 1080p, so the shipped library does not). Camera rays in pixel-block order mixed with
    incoherent and axis-parallel rays (which the packet kernel hands on to the per-lane kernel): hit records equal
    the oracle's bit for bit."""
    monkeypatch.setenv("RTX_QUERY_PACKET", "1")
    c = pkg.Core(0, lib_tuning)
    for sc, cam_kw in ((scenes.soup(60_000, seed=5), dict(frm=(0, 0, 3.5), to=(0, 0, 0), up=(0, 1, 0), fov=45.0)), (scenes.mixed_scene(32, (128, 64)), scenes.mixed_camera())):
        c.upload_scene(sc)
        osc = orc.scene(sc)
        cam = pkg.look_at(lib_tuning, width=320, height=200, **cam_kw)
        osc.set_camera(cam)
        ys, xs = np.mgrid[0:200, 0:320]
        # 8x8 pixel blocks, row-major inside a block: 64 consecutive rays are one block
        order = np.argsort(((ys // 8) * 40 + xs // 8).ravel() * 64 + ((ys % 8) * 8 + xs % 8).ravel(), kind="stable")
        prim = osc.generate_rays(np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], 1)[order])
        rays = np.concatenate([prim, _rand_rays(pkg, 30_000, 3), prim[:6400], _axis_rays(pkg, 3_000, 4)])
        _assert_hits_equal(c.trace_closest(rays), osc.trace_closest(rays, threads=8))
        osc.close()
    c.close()
