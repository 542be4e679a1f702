"""Function-level parity: the device's shading functions on tables of given inputs (rtx_debug_*) against the CPU
restatement of the reference's functions (oracle: orc_shading_data, orc_bsdf_table, orc_env_sample, orc_env_eval),
value by value. A film mismatch can be traced to a function with these; they also cover BSDF::PDF
(RTBase/Materials.h:159,214,320,415,465,541,646), which no integrator of the reference calls.

Tolerance: relative 1e-4 (device sinf / cosf / powf / acosf / atan2f differ from glibc's in the last bits); a sampled
direction can come from the other branch of a Fresnel coin flip or the neighbouring cell of a CDF when the given
number lies within rounding of the threshold, so up to 0.2 % of the rows of a table may differ."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _close(got, want, what, rtol=1e-4, atol=2e-6, rows_allowed=0.002):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    both_nan = np.isnan(got) & np.isnan(want)
    both_inf = np.isinf(got) & np.isinf(want) & (np.sign(got) == np.sign(want))
    with np.errstate(invalid="ignore"):
        bad = ~(np.abs(got - want) <= atol + rtol * np.abs(want)) & ~both_nan & ~both_inf
    rows = bad.reshape(bad.shape[0], -1).any(1)
    assert rows.mean() <= rows_allowed, f"{what}: {int(rows.sum())} of {len(rows)} rows differ, first {np.nonzero(rows)[0][:5]}: {got[rows][:2]} vs {want[rows][:2]}"
    return float(rows.mean())


def _table(n, seed):
    rs = np.random.RandomState(seed)
    nrm = rs.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    def around(nrm, below):
        v = rs.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        d = (v * nrm).sum(1, keepdims=True)
        flip = (d < 0) & ~(rs.rand(n, 1) < below)         # mostly on the normal's side, a share behind the surface
        return np.where(flip, v - 2 * d * nrm, v)
    wo, wi = around(nrm, 0.25), around(nrm, 0.25)
    # some grazing directions
    g = rs.rand(n) < 0.05
    wo[g] = wo[g] - (wo[g] * nrm[g]).sum(1, keepdims=True) * nrm[g] * 0.999
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    uv = rs.uniform(-2.0, 3.0, (n, 2))
    u = rs.uniform(1e-6, 1.0 - 1e-6, (n, 3))
    return np.concatenate([nrm, wo, uv, wi, u], 1).astype(np.float32)


@pytest.mark.parametrize("which", ["mixed", "next"])
def test_bsdf_sample_evaluate_pdf_tables(pkg, lib, core, orc, scenes, which):
    """Every material of the two Cornell variants (together: Diffuse, Mirror, Conductor rough and smooth, Glass, Plastic,
    Dielectric, Oren-Nayar, Layered, emissive Diffuse, textured albedo): sample (wi, colour, pdf), evaluate and PDF."""
    sc = scenes.cornell_box(which)
    core.upload_scene(sc)
    osc = orc.scene(sc)
    table = _table(20_000, 3 if which == "mixed" else 4)
    worst = 0.0
    for m in range(len(sc.materials)):
        got, want = core.debug_bsdf_table(m, table), osc.bsdf_table(m, table)
        worst = max(worst, _close(got[:, 0:7], want[:, 0:7], f"{which} material {m} sample"))
        worst = max(worst, _close(got[:, 7:10], want[:, 7:10], f"{which} material {m} evaluate"))
        worst = max(worst, _close(got[:, 10], want[:, 10], f"{which} material {m} PDF"))
    print(f"{which}: {len(sc.materials)} materials, worst table {worst:.4%} of rows beyond the tolerance")
    osc.close()


def test_mixed_scene_materials_and_shading_data(pkg, lib, core, orc, scenes):
    """The mixed-material scene (checker texture, gold conductor, tessellated spheres with interpolated normals):
    calculateShadingData on primary and bounced hits (front and back faces), and its BSDF tables."""
    sc = scenes.mixed_scene(32, (128, 64))
    core.upload_scene(sc)
    osc = orc.scene(sc)
    cam = pkg.look_at(lib, width=160, height=90, **scenes.mixed_camera())
    osc.set_camera(cam)
    ys, xs = np.mgrid[0:90, 0:160]
    rays = osc.generate_rays(np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], 1))
    hits = osc.trace_closest(rays, threads=8)
    # second generation: random directions from the hit points (inside the glass sphere too)
    rs = np.random.RandomState(9)
    ok = hits["id"] != 0xFFFFFFFF
    d = rs.normal(size=(ok.sum(), 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    r2 = np.empty(ok.sum(), pkg.RAY_DTYPE)
    r2["o"] = rays["o"][ok] + rays["d"][ok] * hits["t"][ok][:, None] + d * 1e-3; r2["d"] = d
    h2 = osc.trace_closest(r2, threads=8)
    allr, allh = np.concatenate([rays, r2]), np.concatenate([hits, h2])
    assert (allh["id"] != 0xFFFFFFFF).sum() > 5000
    got, want = core.debug_shading_data(allr, allh), osc.shading_data(allr, allh)
    assert np.array_equal(got[:, 23], want[:, 23])                     # material index
    _close(got[:, :23], want[:, :23], "calculateShadingData", rtol=2e-5, atol=2e-6, rows_allowed=0.0)
    table = _table(10_000, 6)
    for m in range(len(sc.materials)):
        got, want = core.debug_bsdf_table(m, table), osc.bsdf_table(m, table)
        _close(got, want, f"mixed scene material {m}")
    osc.close()


def test_environment_map_sample_evaluate_pdf(pkg, lib, core, orc, scenes):
    """EnvironmentMap::sample on (u0, u1) pairs incl. the ends of the range (guide-table search against the reference's
    binary search), evaluate and PDF on random and axis directions."""
    sc = scenes.mixed_scene(16, (128, 64))
    core.upload_scene(sc)
    osc = orc.scene(sc)
    rs = np.random.RandomState(2)
    u = np.concatenate([rs.uniform(0, 1, (40_000, 2)), [[0, 0], [1, 1], [0, 1], [1, 0], [1e-8, 0.5], [0.5, 1e-8]], rs.uniform(0, 1, (2000, 2)) ** 8]).astype(np.float32)
    got, want = core.debug_env_sample(u), osc.env_sample(u)
    _close(got, want, "EnvironmentMap::sample")
    d = rs.normal(size=(40_000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = np.concatenate([d, np.eye(3), -np.eye(3)]).astype(np.float32)
    _close(core.debug_env_eval(d), osc.env_eval(d), "EnvironmentMap::evaluate / PDF")
    osc.close()
    # constant background: evaluate = the colour, PDF = 1 / 4 pi
    box = scenes.cornell_box("diffuse", background=(0.2, 0.3, 0.5))
    core.upload_scene(box)
    osc = orc.scene(box)
    _close(core.debug_env_eval(d), osc.env_eval(d), "BackgroundColour::evaluate / PDF", rows_allowed=0.0)
    with pytest.raises(pkg.RtxError):
        core.debug_env_sample(u[:4])
    osc.close()
