"""bench.py's roofline block, computed from the committed counter summaries without a GPU: the fields the round-end
driver reads must be there, `frac` must be `achieved / peak`, and the factorisation the docstring of
bench.roofline() states (VALU busy x live lanes x useful share x 1/2) must reproduce `frac` for the timing the
counters were collected at."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_roofline_block_from_committed_profiles():
    bench = importlib.import_module("bench")
    prof = bench.load_profiles()
    head = prof["pmc"]["headline"]
    n = int(head["rays_per_launch"])
    # work per ray of the headline configuration (instrumented kernel; constants of the committed bench line)
    nodes, tris, groups = int(14.26 * n), int(27.88 * n), int(8.54 * n)
    kernel_ms = head["duration_ns"] / 1e6
    r = bench.roofline(n, nodes, tris, groups, kernel_ms, prof, "headline")
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    assert 0.0 < r["frac"] < 1.0 and 0.0 < r["hbm"]["frac"] < 1.0 and 0.0 < r["l1"]["frac"] <= 1.05
    product = r["valu_busy"] * r["live_lane_frac"] * r["useful_share_of_lane_ops"] * 0.5
    assert abs(product - r["frac"]) < 0.1 * r["frac"], (product, r["frac"])
    assert r["traffic"] == head["hbm_bytes_per_launch"]
    # both peaks, and the check that the counters belong to the timing
    assert abs(r["frac_vs_spec_peak"] - r["achieved"] / r["peak_spec"]) < 2e-3 and r["peak_spec"] == 2 * r["peak"]
    assert r["counters_stale"] is False and abs(r["counters_kernel_ms"] - kernel_ms) < 1e-3
    stale = bench.roofline(n, nodes, tris, groups, 0.8 * kernel_ms, prof, "headline")
    assert stale["counters_stale"] is True
    # a launch with no committed counter run carries no counter fields at all (and no nulls but `traffic`)
    bare = bench.roofline(n, nodes, tris, groups, kernel_ms, prof, None)
    assert bare["traffic"] is None and "valu_busy" not in bare and "counters_from" not in bare

    def nulls(o):
        return [k for k, v in o.items() if v is None] + [x for v in o.values() if isinstance(v, dict) for x in nulls(v)]
    assert nulls(bare) == ["traffic"]
