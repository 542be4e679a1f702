#!/usr/bin/env python3
"""Closest-hit launches over the pixel-centre primary rays of one scene and nothing else (for profilers):

    python tools/closest_probe.py --scene soup|terrain|mixed [--width 1920 --height 1080] [--steps 6]

prints kernel milliseconds (HIP events) and Mrays/s."""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="soup")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("ray-tracer_amd")
    scenes = importlib.import_module("ray-tracer_amd.scenes")
    lib = pkg.load_library()
    core = pkg.Core(0, lib)
    if args.scene == "soup":
        scene, cam = scenes.soup(1_000_000, seed=12345), dict(frm=(0, 0, 3.5), to=(0, 0, 0), up=(0, 1, 0), fov=45.0)
    elif args.scene == "terrain":
        scene, cam = scenes.terrain(2236), scenes.terrain_camera()
    else:
        scene, cam = scenes.mixed_scene(128), scenes.mixed_camera()
    core.upload_scene(scene)
    core.set_camera(pkg.look_at(lib, width=args.width, height=args.height, **cam))
    n = args.width * args.height
    rays = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
    hits = torch.empty(n * 20, dtype=torch.uint8, device="cuda")
    core.generate_primary_rays_device(rays.data_ptr(), 0, n)
    torch.cuda.synchronize()
    core.trace_closest_device(rays.data_ptr(), n, hits.data_ptr())
    core.synchronize()
    core.timer_begin()
    for _ in range(args.steps):
        core.trace_closest_device(rays.data_ptr(), n, hits.data_ptr())
    ms = core.timer_end() / args.steps
    print(json.dumps({"scene": args.scene, "rays": n, "kernel_ms": round(ms, 4), "mrays_per_s": round(n / ms / 1e3, 1)}))


if __name__ == "__main__":
    main()
