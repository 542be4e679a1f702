#!/usr/bin/env python3
"""Milliseconds per pass against the number of passes in flight (tuning build, RTX_PIPELINE) for films and tile shares of
different sizes: the mixed-material scene at 1920x1080 whole, as rank 0 of 4 and of 8 under the tile split, and at 640x360.

    RTX_LIBRARY=tuning python tools/pipeline_depth.py"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTX_LIBRARY", "tuning")


def main():
    pkg = importlib.import_module("ray-tracer_amd")
    scenes = importlib.import_module("ray-tracer_amd.scenes")
    lib = pkg.load_library()
    scene = scenes.mixed_scene(128)
    cases = [("1080p", 1920, 1080, 1), ("1080p, tiles of rank 0 of 4", 1920, 1080, 4), ("1080p, tiles of rank 0 of 8", 1920, 1080, 8), ("640x360", 640, 360, 1)]
    for name, w, h, world in cases:
        row = {"case": name}
        for depth in (1, 2, 3, 4, 6, 8):
            os.environ["RTX_PIPELINE"] = str(depth)
            core = pkg.Core(0, lib)
            core.upload_scene(scene)
            core.set_camera(pkg.look_at(lib, width=w, height=h, **scenes.mixed_camera()))
            core.set_settings(pkg.make_settings(total_spp=1 << 20, max_bounces=5, use_mis=1, seed=1, shard_mode=1 if world > 1 else 0, shard_rank=0, shard_world=world))
            core.clear(); core.render_passes(16); core.clear()
            t0 = time.perf_counter()
            core.render_passes(96)
            row[f"ms_per_pass_{depth}_in_flight"] = round((time.perf_counter() - t0) / 96 * 1e3, 4)
            core.close()
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
