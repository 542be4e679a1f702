#!/usr/bin/env python3
"""Full-size runs of the benchmark configurations other than the headline closest-hit one,
on one GPU: path tracing of the Cornell stand-in,
the mixed-material scene at 1080p, the ~10 M-triangle terrain at 4K (whole film and the
tile share of rank 0 of 2), adaptive sampling, light tracing and instant radiosity.
Prints one JSON object per configuration; `--out file` also writes them as a JSON list.

    python tools/run_configs.py [--quick] [--out gpurun_out/configs.json]
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
    ap.add_argument("--quick", action="store_true", help="fewer passes and a 1 M-triangle terrain")
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="", help="comma list of configuration names to run")
    args = ap.parse_args()
    pkg = importlib.import_module("ray-tracer_amd")
    scenes = importlib.import_module("ray-tracer_amd.scenes")
    lib = pkg.load_library()
    core = pkg.Core(0, lib)
    results = []
    only = set(x for x in args.only.split(",") if x)

    def timed(name, scene, cam_kw, w, h, passes, settings, note="", warm=2, extra=None):
        if only and name not in only:
            return None
        t0 = time.perf_counter()
        core.upload_scene(scene)
        t_up = time.perf_counter() - t0
        cam = pkg.look_at(lib, width=w, height=h, **cam_kw)
        core.set_camera(cam)
        core.set_settings(settings)
        core.clear()
        core.render_passes(warm)
        core.clear()
        core.reset_stats()
        t0 = time.perf_counter()
        core.render_passes(passes)
        dt = time.perf_counter() - t0
        st = core.stats()
        film, spp, tiles = core.film()
        r = {"config": name, "triangles": int(scene.n_triangles), "resolution": [w, h], "passes": passes,
             "spp_per_s": round(passes / dt, 3), "ms_per_pass": round(dt / passes * 1e3, 3),
             "msamples_per_s": round(st.samples / dt / 1e6, 2), "scene_upload_s": round(t_up, 2),
             "film_mean": float(film.mean() / max(spp, 1)), "note": note}
        if extra:
            r.update(extra(film, spp, tiles))
        results.append(r)
        print(json.dumps(r), flush=True)
        return r

    q = args.quick
    # configs[0]: Cornell stand-in, path trace, 64 spp, 5 bounces, MIS on
    cornell = scenes.cornell_box("diffuse")
    st = pkg.make_settings(total_spp=64, max_bounces=5, use_mis=1)
    r = timed("cornell_pt_64spp", cornell, scenes.cornell_camera(), 1024, 1024, 16 if q else 64, st,
              "stand-in for Ray-Tracer-Scenes scene 0 (asset not available offline)")
    # configs[2]: mixed materials + env-map MIS, 1080p, 256 spp
    mixed = scenes.mixed_scene(128)
    timed("mixed_env_mis_1080p", mixed, scenes.mixed_camera(), 1920, 1080, 32 if q else 256,
          pkg.make_settings(total_spp=256, max_bounces=5, use_mis=1))

    # configs[4]: adaptive sampling, light tracing, instant radiosity on the mixed scene
    def tile_stats(film, spp, tiles):
        return {"tile_target_min": int(tiles.min()), "tile_target_max": int(tiles.max()), "tile_target_mean": float(tiles.mean())}
    timed("mixed_adaptive_10_256", mixed, scenes.mixed_camera(), 1920, 1080, 32 if q else 256,
          pkg.make_settings(total_spp=256, init_spp=10, adaptive_sampling=1, max_bounces=5), extra=tile_stats)
    timed("mixed_light_trace", mixed, scenes.mixed_camera(), 1920, 1080, 16 if q else 64,
          pkg.make_settings(algorithm=pkg.AL_LIGHT_TRACE, total_spp=64, max_bounces=5))
    # 128 light paths per pass (8 threads x 16 paths, RTBase/Renderer.h:174,283); the row records the valid VPLs they leave
    def vpl_stats(film, spp, tiles):
        w = core.count_render_work(2)
        return {"light_paths_per_pass": 128, "valid_vpls_per_pass": w["vpls"] / 2, "gather_segments_per_pass": w["gather_segments"] / 2}
    timed("mixed_instant_radiosity", mixed, scenes.mixed_camera(), 1920, 1080, 4 if q else 24,
          pkg.make_settings(algorithm=pkg.AL_INSTANT_RADIOSITY, total_spp=1 << 20, max_bounces=5, vpl_rays_per_tile=16, num_threads=8), warm=2, extra=vpl_stats)

    # configs[3]: ~10 M triangles at 4K; whole film, then the tiles of rank 0 of 2
    terr = scenes.terrain(707 if q else 2236)
    timed("terrain_4k_pt", terr, scenes.terrain_camera(), 3840, 2160, 4 if q else 16, pkg.make_settings(total_spp=16, max_bounces=5), warm=1)
    timed("terrain_4k_pt_tiles_rank0_of_2", terr, scenes.terrain_camera(), 3840, 2160, 4 if q else 16,
          pkg.make_settings(total_spp=16, max_bounces=5, shard_mode=1, shard_rank=0, shard_world=2), warm=1,
          note="one rank's share under tile sharding (tiles t % 2 == 0)")

    if args.out:
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
