#!/usr/bin/env python3
"""What one rank of N has to do, measured on one GPU: milliseconds per GLOBAL sample per pixel for rank 0 of
N = 1, 2, 4, 8 under both partitionings (tiles: every pass, a 1/N share of the tiles; spp: every N-th pass, the whole
film), against the ideal 1/N of the single-GPU time. The part that does not shrink with N is the scaling loss a real
multi-GPU run would show before any communication.

    python tools/rank_share.py [--out profiles/r03_rank_share.json] [--quick]
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    pkg = importlib.import_module("ray-tracer_amd")
    scenes = importlib.import_module("ray-tracer_amd.scenes")
    lib = pkg.load_library()
    core = pkg.Core(0, lib)
    rows = []
    cases = [("mixed_env_mis_1080p", lambda: scenes.mixed_scene(128), scenes.mixed_camera(), 1920, 1080, 48),
             ("terrain10m_4k", lambda: scenes.terrain(707 if args.quick else 2236), scenes.terrain_camera(), 3840, 2160, 12)]
    for name, make, cam, w, h, passes in cases:
        core.upload_scene(make())
        core.set_camera(pkg.look_at(lib, width=w, height=h, **cam))
        base = None
        for mode, mode_id in (("tiles", 1), ("spp", 2)):
            for world in (1, 2, 4, 8):
                st = pkg.make_settings(total_spp=1 << 20, max_bounces=5, use_mis=1, seed=1, shard_mode=mode_id if world > 1 else 0, shard_rank=0, shard_world=world)
                core.set_settings(st)
                core.clear(); core.render_passes(3); core.clear()
                # local passes that cover `passes` global samples
                local = passes if mode == "tiles" else max(1, passes // world)
                t0 = time.perf_counter()
                core.render_passes(local)
                dt = time.perf_counter() - t0
                per_global = dt / (local if mode == "tiles" else local * world)
                if world == 1:
                    base = per_global if base is None else min(base, per_global)
                rows.append({"scene": name, "resolution": [w, h], "split": mode, "world": world, "rank": 0, "local_passes_timed": local,
                             "ms_per_global_sample": round(per_global * 1e3, 4)})
                print(json.dumps(rows[-1]), flush=True)
        for r in rows:
            if r["scene"] == name:
                r["ideal_ms"] = round(base * 1e3 / r["world"], 4)
                r["share_of_single_gpu_time"] = round(r["ms_per_global_sample"] / (base * 1e3), 4)
                r["efficiency_before_communication"] = round(base * 1e3 / r["world"] / r["ms_per_global_sample"], 4)
    if args.out:
        json.dump({"note": "rank 0 of N on one MI355X, three passes in flight; film all-reduce not included (once per render: W*H*12 bytes per rank)", "rows": rows},
                  open(args.out, "w"), indent=1)
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
