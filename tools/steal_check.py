#!/usr/bin/env python3
"""Closest hit with and without work stealing between workgroups (tuning build, RTX_STEAL=1 / 0): equal hit buffers and
kernel milliseconds, interleaved rounds.

    python tools/steal_check.py [--scene soup|mixed|terrain] [--width 1920 --height 1080] [--rounds 7]"""
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
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--mode", default="1", help="RTX_STEAL value of the stealing context (1 everything, 2 no offers, 3 no thieves)")
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("ray-tracer_amd")
    scenes = importlib.import_module("ray-tracer_amd.scenes")
    lib = pkg.load_library(pkg.LIB_TUNING_PATH)
    if args.scene == "soup":
        scene, cam = scenes.soup(1_000_000, seed=12345), dict(frm=(0, 0, 3.5), to=(0, 0, 0), up=(0, 1, 0), fov=45.0)
    elif args.scene == "terrain":
        scene, cam = scenes.terrain(2236), scenes.terrain_camera()
    else:
        scene, cam = scenes.mixed_scene(128), scenes.mixed_camera()
    cores = {}
    for mode in ("1", "0"):
        os.environ["RTX_STEAL"] = args.mode if mode == "1" else "0"          # read when a context is created
        c = pkg.Core(0, lib)
        c.upload_scene(scene)
        c.set_camera(pkg.look_at(lib, width=args.width, height=args.height, **cam))
        cores[mode] = c
    n = args.width * args.height
    rays = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
    cores["1"].generate_primary_rays_device(rays.data_ptr(), 0, n)
    torch.cuda.synchronize()
    hits = {m: torch.zeros(n * 20, dtype=torch.uint8, device="cuda") for m in cores}
    for m, c in cores.items():
        c.trace_closest_device(rays.data_ptr(), n, hits[m].data_ptr()); c.synchronize()
    equal = bool(torch.equal(hits["1"], hits["0"]))
    print(json.dumps({"hit_buffers_equal": equal}), flush=True)
    ms = {m: [] for m in cores}
    for _ in range(args.rounds):
        for m, c in cores.items():
            hits[m].zero_(); torch.cuda.synchronize()
            c.timer_begin()
            for _ in range(args.steps):
                c.trace_closest_device(rays.data_ptr(), n, hits[m].data_ptr())
            ms[m].append(c.timer_end() / args.steps)
        equal = equal and bool(torch.equal(hits["1"], hits["0"]))
    out = {"scene": args.scene, "rays": n, "hit_buffers_equal": equal}
    for m in cores:
        v = sorted(ms[m]); out["steal" if m == "1" else "plain"] = {"median_ms": round(v[len(v) // 2], 4), "min_ms": round(v[0], 4), "mrays_per_s": round(n / v[len(v) // 2] / 1e3, 1)}
    print(json.dumps(out))
    sys.exit(0 if equal else 1)


if __name__ == "__main__":
    main()
