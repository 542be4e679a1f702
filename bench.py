#!/usr/bin/env python3
"""Headline benchmark: closest-hit Mrays/s on the 1 M-triangle soup at 1920x1080
(benchmark configuration 1), plus the 1080p path-trace spp/s of the mixed-material scene as
extra keys on the same JSON line.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed / RCCL when N > 1). A step is one closest-hit
launch over one batch of 2,073,600 primary rays that are already resident in HBM (the
core's own 8x8-tile queue order). Every rank traces its own batch (weak scaling, no
data-path collective); value = rays of all ranks / max-over-ranks time.
"""
from __future__ import annotations

import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
NODE_BYTES, TRI_BYTES, RAY_BYTES, HIT_BYTES = 128, 36, 24, 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--triangles", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pathtrace", action="store_true")
    ap.add_argument("--pt-spp", type=int, default=64, help="passes of the path-trace side measurement")
    ap.add_argument("--variants", default="", help="A/B: comma list of RTX_VARIANT values, interleaved rounds in one process")
    ap.add_argument("--ab-scene", default="soup", choices=["soup", "mixed"])
    ap.add_argument("--ab-repeat", type=int, default=1, help="A/B: trace the ray set this many times in one launch (tail measurement)")
    ap.add_argument("--ab-count", type=int, default=0, help="A/B: keep only this many rays (random subset)")
    ap.add_argument("--ab-rays", default="primary", choices=["primary", "bounce", "shadow"],
                    help="bounce: random directions from the primary hit points; shadow: visibility segments from the primary hit points, half to far points in random directions (environment light), half to random points of the scene")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks here, before this process touches a GPU
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # RTX_BENCH_REHEARSAL=1: every rank on GPU 0 and gloo instead of RCCL, to walk the multi-rank
    # control flow on a one-GPU box (RCCL refuses two ranks on one device)
    rehearse = os.environ.get("RTX_BENCH_REHEARSAL") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def all_reduce(t, op=None):
        op = op or dist.ReduceOp.SUM
        if rehearse:
            c = t.cpu(); dist.all_reduce(c, op=op); t.copy_(c)
        else:
            dist.all_reduce(t, op=op)

    pkg = importlib.import_module("ray-tracer_amd")
    scenes = importlib.import_module("ray-tracer_amd.scenes")
    lib = pkg.load_library()
    core = pkg.Core(local_rank, lib)
    # the core keeps its own (non-blocking) stream: kernels are timed with HIP events on that stream
    # (rtx_timer_*), and every hand-over between torch's work and the core's is a device-wide
    # torch.cuda.synchronize()

    if args.variants and args.ab_scene == "mixed":
        scene = scenes.mixed_scene(128)
        cam = pkg.look_at(lib, width=args.width, height=args.height, **scenes.mixed_camera())
    else:
        scene = scenes.soup(args.triangles, seed=12345)
        cam = pkg.look_at(lib, (0, 0, 3.5), (0, 0, 0), (0, 1, 0), 45.0, args.width, args.height)
    core.upload_scene(scene)
    core.set_camera(cam)
    n = args.width * args.height
    rays = torch.empty(n * RAY_BYTES, dtype=torch.uint8, device=dev)
    hits = torch.empty(n * HIT_BYTES, dtype=torch.uint8, device=dev)
    pixel = torch.empty(n, dtype=torch.int32, device=dev)
    core.generate_primary_rays_device(rays.data_ptr(), pixel.data_ptr(), n)
    torch.cuda.synchronize()

    if args.variants and args.ab_rays == "bounce":
        core.trace_closest_device(rays.data_ptr(), n, hits.data_ptr())
        torch.cuda.synchronize()
        r = rays.view(torch.float32).view(n, 6)
        t = hits.view(torch.float32).view(n, 5)[:, 1]
        hit = hits.view(torch.int32).view(n, 5)[:, 0] != -1
        g = torch.Generator(device=dev).manual_seed(7)
        d = torch.randn(n, 3, generator=g, device=dev)
        d = d / d.norm(dim=1, keepdim=True)
        o = r[:, :3] + r[:, 3:] * t[:, None] + 1e-3 * d
        rays = torch.cat([o, d], 1)[hit].contiguous().view(torch.uint8).view(-1)
        n = int(hit.sum().item())
        hits = torch.empty(n * HIT_BYTES, dtype=torch.uint8, device=dev)
        print(json.dumps({"bounce_rays": n}))
    if args.variants and args.ab_rays == "shadow":
        core.trace_closest_device(rays.data_ptr(), n, hits.data_ptr())
        torch.cuda.synchronize()
        r = rays.view(torch.float32).view(n, 6)
        t = hits.view(torch.float32).view(n, 5)[:, 1]
        hit = hits.view(torch.int32).view(n, 5)[:, 0] != -1
        g = torch.Generator(device=dev).manual_seed(9)
        d = torch.randn(n, 3, generator=g, device=dev)
        d = d / d.norm(dim=1, keepdim=True)
        p1 = r[:, :3] + r[:, 3:] * t[:, None] + 1e-3 * d
        far = p1 + 1000.0 * d
        near = (torch.rand(n, 3, generator=g, device=dev) * 2 - 1) * 1.2
        pick = (torch.rand(n, 1, generator=g, device=dev) < 0.5)
        seg = torch.cat([p1, torch.where(pick, far, near)], 1)[hit].contiguous()
        ab_visible(pkg, lib, scene, cam, args, seg, torch, local_rank)
        return
    if args.variants and args.ab_count and args.ab_count < n:
        g = torch.Generator(device=dev).manual_seed(11)
        keep = torch.randperm(n, generator=g, device=dev)[:args.ab_count].sort().values
        rays = rays.view(n, RAY_BYTES)[keep].contiguous().view(-1)
        n = args.ab_count
        hits = torch.empty(n * HIT_BYTES, dtype=torch.uint8, device=dev)
    if args.variants and args.ab_repeat > 1:
        rays = rays.view(n, RAY_BYTES).repeat(args.ab_repeat, 1).contiguous().view(-1)
        n *= args.ab_repeat
        hits = torch.empty(n * HIT_BYTES, dtype=torch.uint8, device=dev)
    if args.variants:
        ab_variants(pkg, lib, scene, cam, args, rays, hits, n, torch, local_rank)
        return

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        core.trace_closest_device(rays.data_ptr(), n, hits.data_ptr())
    barrier()
    t0 = time.perf_counter()
    core.timer_begin()
    for _ in range(args.steps):
        core.trace_closest_device(rays.data_ptr(), n, hits.data_ptr())
    kernel_ms = core.timer_end() / max(args.steps, 1)     # HIP events on the launch stream
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
    mrays = world * n * args.steps / elapsed / 1e6

    # algorithmic traffic of the layout for these rays (instrumented pass, untimed)
    nodes, tris = core.count_closest_work(rays.data_ptr(), n)
    bytes_per_launch = n * (RAY_BYTES + HIT_BYTES) + nodes * NODE_BYTES + tris * TRI_BYTES
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_closest_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "closest-hit Mrays/s (1M-triangle soup, 1920x1080 primary rays, hit ids bit-exact vs reference BVH)",
        "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[1]: {scene.n_triangles}-triangle soup (seed 12345), {args.width}x{args.height} pixel-centre primary rays per GPU, closest hit",
                   "rays_per_step_per_gpu": n, "bvh4_nodes": core.stats().bvh4_nodes},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "kernel": "k_closest", "kernel_ms": round(kernel_ms, 4),
                     "algorithmic_bytes_per_ray": round(bytes_per_launch / n, 1),
                     "nodes_per_ray": round(nodes / n, 2), "tris_per_ray": round(tris / n, 2)},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, scene, cam, args)
    if not args.no_pathtrace:
        pt = pathtrace_side(pkg, scenes, lib, core, torch, dist, world, rank, dev, args, all_reduce)
        if pt:
            out.update(pt)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def spawn_ranks(n: int) -> int:
    """One child process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the same command
    line otherwise (what torch.distributed.run would start); rank 0 prints the JSON line. The parent
    never initialises a GPU. Returns the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def ab_variants(pkg, lib, scene, cam, args, rays, hits, n, torch, local_rank):
    """Interleaved rounds of the closest-hit kernel under different RTX_VARIANT schedules
    (one context each, same rays), median and min kernel ms per variant."""
    cores = {}
    for v in [int(x) for x in args.variants.split(",")]:
        os.environ["RTX_VARIANT"] = str(v)
        c = pkg.Core(local_rank, lib)
        c.upload_scene(scene); c.set_camera(cam)
        cores[v] = c
    ref = None
    times = {v: [] for v in cores}
    for rnd in range(7):
        for v, c in cores.items():
            c.trace_closest_device(rays.data_ptr(), n, hits.data_ptr())
            torch.cuda.synchronize()
            c.timer_begin()
            for _ in range(args.steps):
                c.trace_closest_device(rays.data_ptr(), n, hits.data_ptr())
            times[v].append(c.timer_end() / args.steps)
            h = hits.clone()
            if ref is None:
                ref = h
            assert torch.equal(h, ref), f"variant {v} changes the hits"
    for v, t in times.items():
        t = sorted(t)
        print(json.dumps({"variant": v, "median_ms": round(t[len(t) // 2], 4), "min_ms": round(t[0], 4),
                          "mrays_per_s_median": round(n / t[len(t) // 2] / 1e3, 1)}))


def ab_visible(pkg, lib, scene, cam, args, seg, torch, local_rank):
    """The same interleaved comparison for the visibility kernel; the visibility bytes must agree."""
    n = seg.shape[0]
    vis = torch.empty(n, dtype=torch.uint8, device=seg.device)
    print(json.dumps({"shadow_segments": n}))
    cores = {}
    for v in [int(x) for x in args.variants.split(",")]:
        os.environ["RTX_VARIANT"] = str(v)
        c = pkg.Core(local_rank, lib)
        c.upload_scene(scene); c.set_camera(cam)
        cores[v] = c
    ref = None
    times = {v: [] for v in cores}
    for rnd in range(7):
        for v, c in cores.items():
            c.trace_visible_device(seg.data_ptr(), n, vis.data_ptr())
            torch.cuda.synchronize()
            c.timer_begin()
            for _ in range(args.steps):
                c.trace_visible_device(seg.data_ptr(), n, vis.data_ptr())
            times[v].append(c.timer_end() / args.steps)
            h = vis.clone()
            if ref is None:
                ref = h
            assert torch.equal(h, ref), f"variant {v} changes the visibility"
    for v, t in times.items():
        t = sorted(t)
        print(json.dumps({"variant": v, "median_ms": round(t[len(t) // 2], 4), "min_ms": round(t[0], 4),
                          "msegments_per_s_median": round(n / t[len(t) // 2] / 1e3, 1), "visible_share": round(float(ref.float().mean()), 3)}))


def cpu_baseline(pkg, scene, cam, args):
    """The CPU oracle (a port of the reference's BVH traversal, RTBase/Geometry.h:506-566)
    on this box's host cores, over every 2nd ray of the same batch."""
    pyoracle = importlib.import_module("oracle.pyoracle")
    orc = pyoracle.Oracle(pkg)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 8, 16)
    osc = orc.scene(scene)
    osc.set_camera(cam)
    ys, xs = np.mgrid[0:args.height, 0:args.width]
    xy = np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], 1).astype(np.float32)[::2]
    rays = osc.generate_rays(xy)
    t0 = time.perf_counter()
    osc.trace_closest(rays, threads=threads)
    dt = time.perf_counter() - t0
    osc.close()
    out = {"value": round(len(rays) / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"every 2nd ray of the batch ({len(rays)} rays), {threads} threads, BVH build excluded"}
    if not args.no_pathtrace:
        # the port of renderMT (RTBase/Renderer.h:753-782, per-thread MT19937 streams) on the path-trace side scene
        scenes = importlib.import_module("ray-tracer_amd.scenes")
        mixed = scenes.mixed_scene(128)
        osc = orc.scene(mixed)
        osc.set_camera(pkg.look_at(pkg.load_library(), width=args.width, height=args.height, **scenes.mixed_camera()))
        ren = osc.renderer(pkg.make_settings(total_spp=2, max_bounces=5, use_mis=1), threads=threads, counter_rng=False)
        t0 = time.perf_counter()
        ren.render_passes(2)
        dt = time.perf_counter() - t0
        ren.close(); osc.close()
        out["pathtrace_spp_per_s"] = round(2 / dt, 4)
        out["pathtrace_sample"] = f"2 passes of the {args.width}x{args.height} mixed-material scene, {threads} threads"
    return out


def pathtrace_side(pkg, scenes, lib, core, torch, dist, world, rank, dev, args, all_reduce):
    """1080p path tracing of the mixed-material scene under the environment map
    (benchmark configuration 2); samples are split over the ranks (shard_mode 2) and the films
    summed with one RCCL all-reduce at the end."""
    try:
        scene = scenes.mixed_scene(128)
        core.upload_scene(scene)
        cam = pkg.look_at(lib, width=args.width, height=args.height, **scenes.mixed_camera())
        core.set_camera(cam)
        st = pkg.make_settings(total_spp=args.pt_spp * world, seed=1, shard_mode=2 if world > 1 else 0,
                               shard_rank=rank, shard_world=world)
        core.set_settings(st)
        core.clear()
        core.render_passes(4)          # warm-up (also brings up every pipeline lane's buffers)
        core.clear()
        if world > 1:
            ptr, nfl = core.film_device()
            all_reduce(torch.zeros(nfl, dtype=torch.float32, device=dev))     # warm-up of the collective at film size
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        core.render_passes(args.pt_spp)
        if world > 1:
            ptr, nfl = core.film_device()
            film = torch.empty(nfl, dtype=torch.float32, device=dev)
            ctypes.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(ctypes.c_void_p(film.data_ptr()), ctypes.c_void_p(ptr),
                                                                   ctypes.c_size_t(nfl * 4), 3)
            all_reduce(film)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            all_reduce(t, dist.ReduceOp.MAX)
            dt = float(t.item())
        s = core.stats()
        return {"pathtrace_spp_per_s": round(args.pt_spp * world / dt, 3),
                "pathtrace_config": {"workload": f"configs[2]: mixed Diffuse/Mirror/Glass/Conductor/Plastic scene ({scene.n_triangles} triangles) + 1024x512 env-map MIS, {args.width}x{args.height}, maxBounces 5",
                                     "spp_timed": args.pt_spp * world, "parallelism": f"spp split x{world} + film all-reduce" if world > 1 else "1 GPU"}}
    except pkg.RtxError as e:
        return {"pathtrace_spp_per_s": None, "pathtrace_error": str(e)}


if __name__ == "__main__":
    main()
