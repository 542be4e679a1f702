"""CPU tests of the oracle itself: the reference's own known-answer cases, the golden
vectors produced by the real RTBase/Core.h + RTBase/Sampling.h, and internal consistency of
the restated BVH (no GPU needed)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_f = C.POINTER(C.c_float)


def _fp(a):
    return a.ctypes.data_as(_f)


def test_aabb_known_answers_of_the_reference_unit_tests(orc):
    kat = json.load(open(os.path.join(GOLD, "aabb_kat.json")))
    assert len(kat["cases"]) == 3
    for c in kat["cases"]:
        assert orc.ray_aabb(c["min"], c["max"], c["origin"], c["dir"]) == c["hit"], c["name"]


def test_aabb_nan_semantics_of_the_reference_slab_test(orc):
    """0 * inf = NaN inside AABB::rayAABB (RTBase/Geometry.h:213-228): Min/Max pick the second
    operand, std::max/std::min keep the first on an unordered compare."""
    # origin on the box's max-x plane, direction with x == 0: x slab yields NaN entry/exit and,
    # x being the first operand of std::max/min, the whole test degenerates to "hit"
    assert orc.ray_aabb((0, 0, 0), (1, 1, 1), (1, 5, 5), (0, 1, 0)) is True
    # the same on the y axis: NaN is the second operand there and is ignored -> real miss
    assert orc.ray_aabb((0, 0, 0), (1, 1, 1), (5, 1, 5), (1, 0, 0)) is False
    # origin on the min-x plane: entry = +inf -> miss although the ray lies in the face
    assert orc.ray_aabb((0, 0, 0), (1, 1, 1), (0, 0.5, -1), (0, 0, 1)) is False
    # strictly inside the slab with a zero component: (-inf, +inf) -> decided by the other axes
    assert orc.ray_aabb((0, 0, 0), (1, 1, 1), (0.5, 0.5, -1), (0, 0, 1)) is True


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "ref_core_vectors.npz"))


def _call(lib, name, *arrays, n_out):
    out = np.empty(n_out, np.float32)
    getattr(lib, name)(*[_fp(np.ascontiguousarray(a, np.float32)) for a in arrays], _fp(out))
    return out


def test_math_layer_matches_real_reference_vectors_bit_for_bit(orc, gold):
    lib = orc.lib
    for n in ("orc_matrix_invert", "orc_look_at", "orc_mul_point", "orc_mul_vec", "orc_mul_point_divide", "orc_frame"):
        getattr(lib, n).restype = None
    for m, inv in zip(gold["matrices"], gold["inverse"]):
        assert np.array_equal(_call(lib, "orc_matrix_invert", m, n_out=16), inv)
    for r, want in zip(gold["lookat_in"], gold["lookat"]):
        assert np.array_equal(_call(lib, "orc_look_at", r[0:3], r[3:6], r[6:9], n_out=16), want)
    lib.orc_perspective.argtypes = [C.c_float] * 4 + [_f]
    for r, want in zip(gold["persp_in"], gold["persp"]):
        out = np.empty(16, np.float32)
        lib.orc_perspective(float(r[0]), float(r[1]), float(r[2]), float(r[3]), _fp(out))
        assert np.array_equal(out, want)
    for m, v, a, b, c in zip(gold["matrices"], gold["vecs"], gold["mul_point"], gold["mul_vec"], gold["mul_point_divide"]):
        assert np.array_equal(_call(lib, "orc_mul_point", m, v, n_out=3), a)
        assert np.array_equal(_call(lib, "orc_mul_vec", m, v, n_out=3), b)
        assert np.array_equal(_call(lib, "orc_mul_point_divide", m, v, n_out=3), c)
    for n, v, want in zip(gold["frame_n"], gold["frame_v"], gold["frame"]):
        assert np.array_equal(_call(lib, "orc_frame", n, v, n_out=15), want)
    lib.orc_lum.restype = C.c_float
    for c, want in zip(gold["lum_in"], gold["lum"]):
        assert np.float32(lib.orc_lum(_fp(np.ascontiguousarray(c)))) == want


def test_sampling_layer_matches_real_reference_vectors(orc, gold):
    lib = orc.lib
    lib.orc_warp.argtypes = [C.c_int, C.c_float, C.c_float, _f]
    for kind, name in ((0, "cosine_hemisphere"), (1, "uniform_sphere"), (2, "uniform_hemisphere")):
        for (a, b), want in zip(gold["warp_u"], gold["warp_" + name]):
            out = np.empty(4, np.float32)
            lib.orc_warp(kind, float(a), float(b), _fp(out))
            assert np.array_equal(out, want), (name, a, b)
    lib.orc_mt_stream.argtypes = [C.c_uint, C.c_int, _f]
    for seed, want in zip(gold["mt_seeds"], gold["mt_stream"]):
        out = np.empty(64, np.float32)
        lib.orc_mt_stream(int(seed), 64, _fp(out))
        assert np.array_equal(out, want)


def test_math_layer_against_live_reference_build(orc, refcore):
    """Where oracle/_ref exists (reference tree present at build time): random inputs beyond
    the committed vectors."""
    rs = np.random.RandomState(7)
    lib = orc.lib
    for _ in range(200):
        m = rs.uniform(-3, 3, 16).astype(np.float32)
        assert np.array_equal(_call(lib, "orc_matrix_invert", m, n_out=16), refcore.invert(m))
        n, v = rs.normal(size=3).astype(np.float32), rs.normal(size=3).astype(np.float32)
        assert np.array_equal(_call(lib, "orc_frame", n, v, n_out=15), refcore.frame(n, v))


def test_bvh_invariants_and_brute_force_agreement(pkg, orc, scenes):
    sc = scenes.soup(4000, seed=3, size=0.05)
    osc = orc.scene(sc)
    bounds, topo, idx, depth = osc.bvh()
    assert sorted(idx.tolist()) == list(range(sc.n_triangles))          # a permutation
    leaf = (topo[:, 2] == 0) & (topo[:, 3] == 0)
    assert ((topo[leaf, 1] - topo[leaf, 0]) <= 8).all() and depth <= 50   # MAXNODE_TRIANGLES / maxDepth
    for i in np.nonzero(~leaf)[0]:
        l, r = topo[i, 2], topo[i, 3]
        assert topo[l, 0] == topo[i, 0] and topo[l, 1] == topo[r, 0] and topo[r, 1] == topo[i, 1]
        for c in (l, r):
            assert (bounds[c, :3] >= bounds[i, :3]).all() and (bounds[c, 3:] <= bounds[i, 3:]).all()
    # closest hit through the tree == brute force over all triangles (Scene::traverseAll,
    # RTBase/Scene.h:113-135) on rays without exact ties
    rs = np.random.RandomState(1)
    rays = np.empty(3000, pkg.RAY_DTYPE)
    rays["o"] = rs.uniform(-1.5, 1.5, (3000, 3)); d = rs.normal(size=(3000, 3))
    rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    hits = osc.trace_closest(rays, threads=4)
    P = sc.positions.astype(np.float64)
    e1, e2, v2 = P[:, 0] - P[:, 2], P[:, 1] - P[:, 2], P[:, 2]
    checked = 0
    for k in np.nonzero(hits["id"] != 0xFFFFFFFF)[0][:300]:
        o, dd = rays["o"][k].astype(np.float64), rays["d"][k].astype(np.float64)
        p = np.cross(dd, e2); det = (p * e1).sum(1)
        ok = np.abs(det) > 1e-8
        inv = np.where(ok, 1.0 / np.where(ok, det, 1), 0)
        T = o - v2; u = (T * p).sum(1) * inv; q = np.cross(T, e1); v = (q * dd).sum(1) * inv; t = (e2 * q).sum(1) * inv
        cand = ok & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 1e-8)
        best = np.where(cand, t, np.inf).argmin()
        assert best == hits["id"][k] and abs(t[best] - hits["t"][k]) <= 1e-4 * t[best]
        checked += 1
    assert checked > 100


def test_renderer_is_deterministic_and_thread_count_independent_with_counter_rng(pkg, lib, orc, scenes):
    sc = scenes.cornell_box("mixed")
    cam = orc.look_at((0, 0, 3.4), (0, 0, 0), (0, 1, 0), 45.0, 40, 32)
    osc = orc.scene(sc); osc.set_camera(cam)
    films = []
    for threads in (1, 5):
        r = osc.renderer(pkg.make_settings(total_spp=2, seed=4), threads=threads, counter_rng=True)
        r.render_passes(2)
        films.append(r.film()[0])
    assert np.array_equal(films[0], films[1]) and films[0].sum() > 0


def test_film_splat_box_filter_and_tile_targets(pkg, orc, scenes):
    sc = scenes.cornell_box()
    cam = orc.look_at((0, 0, 3.4), (0, 0, 0), (0, 1, 0), 45.0, 40, 24)
    osc = orc.scene(sc); osc.set_camera(cam)
    r = osc.renderer(pkg.make_settings(total_spp=16, init_spp=4, adaptive_sampling=1), threads=1)
    r.splat(np.array([[3.7, 2.2, 1, 2, 3], [3.1, 2.9, 1, 1, 1], [39.99, 23.5, 5, 5, 5], [40.0, 3.0, 9, 9, 9]], np.float32))
    film = r.film()[0]
    assert np.array_equal(film[2, 3], [2, 3, 4]) and np.array_equal(film[23, 39], [5, 5, 5]) and film.sum() == 24
    # constant film -> zero variance -> weight clamps to EPSILON -> target = SPP + (total-SPP)*0.001 truncated
    r.film_write(np.full((24, 40, 3), 2.0, np.float32), 4)
    r.calculate_tile_samples()
    assert (r.film()[2] == 4).all()


def test_oren_nayar_dielectric_and_layered_bsdfs(pkg, orc, scenes):
    """OrenNayarBSDF / DielectricBSDF / LayeredBSDF (RTBase/Materials.h:436-562, 680-724) against
    the closed forms written out here in numpy: parity unpinned by the reference's own tests."""
    sc = scenes.cornell_box("next")
    osc = orc.scene(sc)
    rs = np.random.RandomState(5)
    n = 256
    nrm = np.tile(np.array([0, 0, 1], np.float32), (n, 1))          # frame = (x, y, z) up to the ONB's choice of tangents
    wo = rs.normal(size=(n, 3)).astype(np.float32); wo[:, 2] = np.abs(wo[:, 2]) + 0.05; wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wi = rs.normal(size=(n, 3)).astype(np.float32); wi[:, 2] = np.abs(wi[:, 2]) + 0.05; wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    u = rs.uniform(0.01, 0.99, size=(n, 3)).astype(np.float32)
    table = np.concatenate([nrm, wo, np.full((n, 2), 0.5, np.float32), wi, u], 1).astype(np.float32)

    # material 0: Oren-Nayar sigma 0.6 on the 0.73 white texture
    out = osc.bsdf_table(0, table)
    sigma = np.float32(0.6)
    A = np.float32(1) - sigma * sigma / (np.float32(2) * (sigma * sigma + np.float32(0.33)))
    B = np.float32(0.45 * float(sigma) * float(sigma) / float(sigma * sigma + np.float32(0.09)))
    # sampled direction: z = sqrt(u0), pdf = z / pi, whatever the tangent frame is
    assert np.allclose(out[:, 6], np.sqrt(u[:, 0]) / np.pi, rtol=2e-6)
    assert np.allclose(np.linalg.norm(out[:, 0:3], axis=1), 1, atol=1e-5)
    assert np.allclose(out[:, 2], np.sqrt(u[:, 0]), atol=1e-6)      # normal = +z, so world z = local z
    # evaluate: albedo/pi * (A + B max(cos(dphi),0) sin(max(ti,to) * tan(min(ti,to)))); dphi is frame independent
    ti, to = np.arccos(wi[:, 2]), np.arccos(wo[:, 2])
    dphi = np.arctan2(wi[:, 1], wi[:, 0]) - np.arctan2(wo[:, 1], wo[:, 0])
    term = A + B * np.maximum(np.cos(dphi), 0) * np.sin(np.maximum(ti, to) * np.tan(np.minimum(ti, to)))
    assert np.allclose(out[:, 7], 0.73 / np.pi * term, rtol=2e-4, atol=2e-6)
    assert np.allclose(out[:, 10], np.abs(wi[:, 2]) / np.pi, rtol=2e-6)

    # material 4: Dielectric = cosine lobe with f = albedo / pi, PDF |z| / pi (u1 drives cos theta: first draw feeds r2)
    out = osc.bsdf_table(4, table)
    assert np.allclose(out[:, 2], np.sqrt(u[:, 1]), atol=1e-6)
    assert np.allclose(out[:, 6], np.sqrt(u[:, 1]) / np.pi, rtol=2e-6)
    assert np.allclose(out[:, 3:6], 1.0 / np.pi, rtol=2e-6) and np.allclose(out[:, 7:10], 1.0 / np.pi, rtol=2e-6)
    assert np.allclose(out[:, 10], np.abs(wi[:, 2]) / np.pi, rtol=2e-6)

    # Layered(X) answers exactly like X: material 6 = Layered(Glass 1.5), material 7 = Layered(Plastic)
    mixed = orc.scene(scenes.cornell_box("mixed"))
    assert np.array_equal(osc.bsdf_table(6, table), mixed.bsdf_table(6, table))      # Glass 1.5 / 1.0, white
    plastic_next, plastic_mixed = osc.bsdf_table(7, table), mixed.bsdf_table(5, table)   # same parameters, different albedo
    assert np.array_equal(plastic_next[:, [0, 1, 2, 6, 10]], plastic_mixed[:, [0, 1, 2, 6, 10]])
    osc.close(); mixed.close()


def test_two_sidedness_of_the_remaining_bsdfs(pkg, lib, orc, scenes):
    """Dielectric is one-sided (Materials.h:475), anything wrapped in LayeredBSDF two-sided
    (:716): seen from behind, only the two-sided surface turns its normal to the viewer
    (Scene.h:199-205)."""
    sc = scenes.cornell_box("next")
    osc = orc.scene(sc)
    mi = sc.material_index
    for mat, flips in ((4, False), (6, True)):
        t = int(np.nonzero(mi == mat)[0][0])
        p = sc.positions[t]; c = p.mean(0)
        gn = np.cross(p[1] - p[0], p[2] - p[0]); gn /= np.linalg.norm(gn)
        o = c - gn * 0.01                                             # just behind the face
        ray = np.zeros(1, pkg.RAY_DTYPE); ray["o"] = o; ray["d"] = gn
        hit = osc.trace_closest(ray, threads=1)
        assert hit["id"][0] == t
        sd = osc.shading_data(ray, hit)[0]
        # shading_data layout: x(3) wo(3) sNormal(3) gNormal(3) ...
        s_n = sd[6:9]
        facing = float(np.dot(s_n, -gn))                             # wo = -d = -gn
        assert (facing > 0) == flips, (mat, facing)
    osc.close()


def test_oracle_reproduces_frozen_path_vectors(pkg, lib, orc, scenes):
    """tests/golden/path_vectors.npz (make_path_vectors.py): the oracle of today gives the hits,
    visibility bits, per-sample films and tile targets that were frozen."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_path_vectors", os.path.join(os.path.dirname(__file__), "golden", "make_path_vectors.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "path_vectors.npz"))
    rays, seg = gen.inputs(pkg, scenes)
    osc = orc.scene(scenes.soup(2000, seed=77, size=0.08))
    hits = osc.trace_closest(rays, threads=4)
    assert np.array_equal(hits["id"], gold["hit_id"]) and np.array_equal(hits["t"], gold["hit_t"])
    assert np.array_equal(np.stack([hits["alpha"], hits["beta"], hits["gamma"]], 1), gold["hit_bary"])
    assert np.array_equal(osc.trace_visible(seg, threads=4), gold["visible"])
    osc.close()
    osc = orc.scene(scenes.cornell_box("mixed"))
    osc.set_camera(pkg.look_at(lib, width=48, height=36, **scenes.cornell_camera()))
    r = osc.renderer(pkg.make_settings(total_spp=64, init_spp=3, adaptive_sampling=1, seed=11), threads=2, counter_rng=True)
    for k in range(3):
        r.render_passes(1)
        assert np.allclose(r.film()[0], gold["pt_films"][k], rtol=1e-6, atol=1e-7), f"pass {k + 1}"
    assert np.array_equal(r.film()[2], gold["tile_targets"])
    r.close(); osc.close()


def test_argument_order_and_spherical_coordinates_match_real_reference_vectors(orc, gold):
    """`f(sampler->next(), sampler->next())` — the call shape of every BSDF / light sample in the
    reference — leaves the order of the two draws to the compiler. The vectors were produced by the
    real headers under g++ on a real MTRandom; the oracle's explicit "second parameter first" must
    reproduce them bit for bit, and the opposite order must not (the test would be vacuous otherwise).
    SphericalCoordinates (RTBase/Core.h:557-572) likewise."""
    lib = orc.lib
    lib.orc_warp_from_sampler.argtypes = [C.c_int, C.c_uint, _f]
    lib.orc_warp.argtypes = [C.c_int, C.c_float, C.c_float, _f]
    lib.orc_mt_stream.argtypes = [C.c_uint, C.c_int, _f]
    for kind, name in ((0, "cosine_hemisphere"), (1, "uniform_sphere"), (2, "uniform_hemisphere")):
        for seed, want in zip(gold["wfs_seeds"], gold["wfs_" + name]):
            out = np.empty(3, np.float32)
            lib.orc_warp_from_sampler(kind, int(seed), _fp(out))
            assert np.array_equal(out, want), (name, seed)
            draws = np.empty(2, np.float32)
            lib.orc_mt_stream(int(seed), 2, _fp(draws))
            first_is_r1 = np.empty(4, np.float32)
            lib.orc_warp(kind, float(draws[0]), float(draws[1]), _fp(first_is_r1))
            assert not np.array_equal(first_is_r1[:3], want), "draw order not observable for this seed"
            first_is_r2 = np.empty(4, np.float32)
            lib.orc_warp(kind, float(draws[1]), float(draws[0]), _fp(first_is_r2))
            assert np.array_equal(first_is_r2[:3], want)
    lib.orc_spherical.argtypes = [C.c_float, C.c_float, _f, _f]
    for (theta, phi), wi, want in zip(gold["sph_in"], gold["sph_wi"], gold["sph"]):
        out = np.empty(5, np.float32)
        lib.orc_spherical(float(theta), float(phi), _fp(np.ascontiguousarray(wi)), _fp(out))
        assert np.array_equal(out, want), (theta, phi, wi)
