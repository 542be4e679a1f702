#!/usr/bin/env python3
"""Builds profiles/rNN_pmc.json (what bench.py reads for the counter-derived fields of its roofline blocks) from the
raw summaries tools/pmc_json.py prints:

    python tools/pmc_build.py --headline raw_headline.json --big raw_big.json --pathtrace raw_pt.json --passes 12 --out profiles/r03_pmc.json --tag r03

headline / big: the k_closest<false> row (1080p primaries of the 1 M-triangle soup; 4K primaries of the 10 M-triangle
terrain). pathtrace: every kernel of `--passes` 1080p passes of the mixed-material scene, one pass in flight."""
import argparse
import json


KEEP = ("valu_busy", "live_lane_frac", "l1_lookups_per_clk_per_cu", "ta_busy", "l2_hit", "hbm_bytes_per_launch", "SQ_THREAD_CYCLES_VALU",
        "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "TCP_TOTAL_CACHE_ACCESSES_sum", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE",
        "TCC_HIT_sum", "TCC_MISS_sum", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "duration_ns", "dispatches")


def closest_row(raw, rays, source):
    k = [v for name, v in raw.items() if name.startswith("k_closest<false>") or name.startswith("k_closest<0>")]
    if not k:
        raise SystemExit(f"no k_closest<false> row in {list(raw)}")
    row = {key: k[0][key] for key in KEEP if key in k[0]}
    row["rays_per_launch"] = rays
    row["source"] = source
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--headline"); ap.add_argument("--big"); ap.add_argument("--pathtrace")
    ap.add_argument("--passes", type=int, default=12)
    ap.add_argument("--out", default="profiles/r03_pmc.json")
    ap.add_argument("--tag", default="r03")
    a = ap.parse_args()
    out = {"note": "rocprofv3 --pmc passes (tools/pmc_collect.sh), mean per dispatch (tools/pmc_json.py); formulas in tools/pmc_json.py"}
    if a.headline:
        out["headline"] = closest_row(json.load(open(a.headline)), 1920 * 1080, f"profiles/{a.tag}_pmc_headline_raw.json: tools/closest_probe.py --scene soup (1 M-triangle soup, 1920x1080 primaries)")
    if a.big:
        out["big"] = closest_row(json.load(open(a.big)), 3840 * 2160, f"profiles/{a.tag}_pmc_terrain10m_raw.json: tools/closest_probe.py --scene terrain --width 3840 --height 2160")
    if a.pathtrace:
        raw = json.load(open(a.pathtrace))
        kernels, total_bytes, total_us = {}, 0, 0.0
        for name, v in raw.items():
            if name.startswith("__amd") or "devbuild" in name or v.get("dispatches", 0) < a.passes // 2:
                continue
            per_pass = v["dispatches"] / a.passes
            kernels[name] = {"launches_per_pass": round(per_pass, 2), "avg_us": round(v.get("duration_ns", 0) / 1e3, 1),
                             "us_per_pass": round(per_pass * v.get("duration_ns", 0) / 1e3, 1),
                             "hbm_bytes_per_launch": v.get("hbm_bytes_per_launch"), "valu_busy": v.get("valu_busy"),
                             "live_lane_frac": v.get("live_lane_frac"), "l2_hit": v.get("l2_hit")}
            total_bytes += per_pass * (v.get("hbm_bytes_per_launch") or 0)
            total_us += per_pass * v.get("duration_ns", 0) / 1e3
        out["pathtrace"] = {"kernels": kernels, "hbm_bytes_per_pass": int(total_bytes), "kernel_us_per_pass": round(total_us, 1),
                            "source": f"profiles/{a.tag}_pmc_pathtrace_raw.json: RTX_LIBRARY=tuning RTX_PIPELINE=1 tools/pt_profile.py --single 0 --passes 12 (4 warm-up + 12 passes of the mixed-material scene, 1920x1080, one pass in flight)"}
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out)[:600])


if __name__ == "__main__":
    main()
