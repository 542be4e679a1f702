#!/usr/bin/env python3
"""1080p path tracing of the mixed-material scene: milliseconds per pass with one pass per call
(what RayTracer::renderMT does) and with N passes per call (rtx_render_passes keeps several in flight).

    python tools/pt_profile.py [--passes 48] [--single 12] [--width 1920 --height 1080] [--scene mixed|cornell|terrain]

Under `rocprofv3 --kernel-trace -d DIR -- python tools/pt_profile.py --single 4 --passes 0` the trace holds
whole single passes; `tools/pt_depths.py DIR/**/*.db` then lists the launches of a pass in order."""
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
    ap.add_argument("--passes", type=int, default=48)
    ap.add_argument("--single", type=int, default=12)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="mixed")
    ap.add_argument("--algorithm", type=int, default=0)
    args = ap.parse_args()
    pkg = importlib.import_module("ray-tracer_amd")
    scenes = importlib.import_module("ray-tracer_amd.scenes")
    lib = pkg.load_library()
    core = pkg.Core(0, lib)
    if args.scene == "mixed":
        scene, cam_kw = scenes.mixed_scene(128), scenes.mixed_camera()
    elif args.scene == "cornell":
        scene, cam_kw = scenes.cornell_box("mixed"), scenes.cornell_camera()
    else:
        scene, cam_kw = scenes.terrain(2236), scenes.terrain_camera()
    core.upload_scene(scene)
    core.set_camera(pkg.look_at(lib, width=args.width, height=args.height, **cam_kw))
    core.set_settings(pkg.make_settings(total_spp=100000, seed=1, algorithm=args.algorithm, num_threads=8, vpl_rays_per_tile=16))
    core.clear()
    core.render_passes(4)
    out = {"scene": args.scene, "resolution": [args.width, args.height], "algorithm": args.algorithm}
    if args.single:
        core.clear()
        t0 = time.perf_counter()
        for _ in range(args.single):
            core.render_passes(1)
        out["single_pass_ms"] = round((time.perf_counter() - t0) / args.single * 1e3, 4)
    if args.passes:
        core.clear()
        t0 = time.perf_counter()
        core.render_passes(args.passes)
        dt = time.perf_counter() - t0
        out["pipelined_ms_per_pass"] = round(dt / args.passes * 1e3, 4)
        out["pipelined_spp_per_s"] = round(args.passes / dt, 2)
    film, spp, _ = core.film()
    out["film_mean"] = float(film.mean() / max(spp, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
