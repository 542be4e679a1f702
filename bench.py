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
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
    ap.add_argument("--no-big", action="store_true", help="skip the 10 M-triangle 4K closest-hit measurement")
    ap.add_argument("--only", default="", choices=["", "headline", "extras", "pathtrace"], help="profiling: run the headline leg plus only this part")
    ap.add_argument("--pt-spp", type=int, default=64, help="passes of the path-trace side measurement PER RANK (weak scaling: N ranks render N x this)")
    ap.add_argument("--pt-total-spp", type=int, default=0, help="path-trace leg with this many passes IN TOTAL, split over the ranks (strong scaling); overrides --pt-spp")
    ap.add_argument("--no-radiosity", action="store_true", help="skip the light-tracing and instant-radiosity side measurements")
    ap.add_argument("--variants", default="", help="A/B: comma list of RTX_VARIANT values, interleaved rounds in one process")
    ap.add_argument("--ab-env", default="", help="A/B: NAME=v1,v2,... varies this tuning switch instead of RTX_VARIANT (which stays at the first --variants value)")
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
    lib = pkg.load_library(pkg.LIB_TUNING_PATH if args.variants else "")     # the schedules under test exist in the tuning build only
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

    prof = load_profiles()
    t_elapsed, kernel_ms = time_closest(core, rays, hits, n, args.steps, args.warmup, barrier)
    if world > 1:
        t = torch.tensor([t_elapsed], dtype=torch.float64, device=dev)
        all_reduce(t, dist.ReduceOp.MAX)
        t_elapsed = float(t.item())
    mrays = world * n * args.steps / t_elapsed / 1e6

    # algorithmic work of the layout for these rays (instrumented pass, untimed)
    nodes, tris, groups = core.count_closest_work(rays.data_ptr(), n)
    out = {
        "metric": "closest-hit Mrays/s (1M-triangle soup, 1920x1080 primary rays, hit ids bit-exact vs reference BVH)",
        "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(t_elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[1]: {scene.n_triangles}-triangle soup (seed 12345), {args.width}x{args.height} pixel-centre primary rays per GPU, closest hit",
                   "rays_per_step_per_gpu": n, "bvh4_nodes": core.stats().bvh4_nodes},
        "roofline": roofline(n, nodes, tris, groups, kernel_ms, prof, "headline"),
    }
    only = args.only
    if rank == 0 and world == 1 and only in ("", "extras"):
        # two launches in flight (a second context and stream): what the tail of a single 1080p launch costs
        out["closest_two_launches_in_flight"] = closest_two_in_flight(pkg, lib, core, scene, cam, rays, hits, n, torch, dev, args, barrier)
        # the same kernel where the tail of a launch weighs less (4x the rays per launch) ...
        out["closest_soup_4k"] = closest_extra(pkg, lib, core, torch, dev, None, (0, 0, 3.5), (0, 0, 0), 45.0, 3840, 2160, args, prof, None, barrier)
        if not args.no_big:
            # ... and where tree and triangles (1.6 GB) no longer fit the caches: benchmark configuration 3's scene
            out["closest_terrain10m_4k"] = closest_extra(pkg, lib, core, torch, dev, scenes.terrain(2236), scenes.terrain_camera()["frm"], scenes.terrain_camera()["to"], 45.0,
                                                         3840, 2160, args, prof, "big", barrier)
    if not args.no_pathtrace and only in ("", "pathtrace"):
        pt = pathtrace_side(pkg, scenes, lib, core, torch, dist, world, rank, dev, args, all_reduce, prof)
        if pt:
            out.update(pt)
    # (last: sixteen host threads at work just before them showed in the GPU legs that time one call per pass)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and only == "":
        out["cpu_baseline"] = cpu_baseline(pkg, scene, cam, args)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


# ---- ceilings and committed profile summaries --------------------------------------------------------
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak
VALU_PEAK_SPEC_TFLOPS = 157.3    # fp32 vector peak of the data sheet: every lane a fused multiply-add (two flops) per clock
VALU_PEAK_NO_FMA_TFLOPS = 78.65  # the same issue rate with plain IEEE multiplies and adds, one flop each: contraction into multiply-adds
#                                  changes the rounding, and hit ids / distances must match the reference's plain evaluation bit for bit
SLAB_FLOPS, TRI_FLOPS = 25, 55   # per ray-box test (RTBase/Geometry.h:216-227) and per Moller-Trumbore test (:99-120)
NODE_LOADS, GROUP_LOADS = 7, 9   # 16-byte loads per lane: one 4-wide node (112 of its 128 bytes), one leaf fetch (four 36-byte triangle records)


def load_profiles():
    """profiles/r02_ceilings.json: the measured ceiling of per-lane 16-byte gathers (tools/gather_probe.hip);
    profiles/r03_pmc.json (or the one of the round before): hardware-counter summaries of the same commands
    (tools/pmc_collect.sh). Both are committed measurements, taken in a profiler run of their own: the roofline blocks
    say which file the counter fields come from, how long the kernel took in THAT run (`counters_kernel_ms`) and flag
    them `counters_stale` when the kernel timed live differs from it by more than 10 %. A missing file only removes
    the fields that need it."""
    out = {}
    for key, names in (("ceilings", ("r02_ceilings.json",)), ("pmc", ("r03_pmc.json", "r02_pmc.json"))):
        out[key] = {}
        for name in names:
            try:
                out[key] = json.load(open(os.path.join(ROOT, "profiles", name)))
                break
            except Exception:
                pass
    return out


def time_closest(core, rays, hits, n, steps, warmup, barrier):
    """Wall seconds of `steps` launches between two barriers, and the mean kernel duration by HIP events on
    the launch stream."""
    for _ in range(warmup):
        core.trace_closest_device(rays.data_ptr(), n, hits.data_ptr())
    barrier()
    t0 = time.perf_counter()
    core.timer_begin()
    for _ in range(steps):
        core.trace_closest_device(rays.data_ptr(), n, hits.data_ptr())
    kernel_ms = core.timer_end() / max(steps, 1)
    barrier()
    return time.perf_counter() - t0, kernel_ms


def roofline(n, nodes, tris, groups, kernel_ms, prof, pmc_key):
    """What bounds the closest-hit kernel and how far it is from it.

    Not HBM: node and triangle fetches are served on chip (the `hbm` block prices the memory-side bytes of the
    counters against 8 TB/s). Two things saturate together once the tail of a launch is amortised: vector-ALU
    issue slots and L1 line look-ups (every lane follows its own ray, so a 16-byte load instruction looks up one
    line per live lane, and an L1 serves about one look-up per clock, measured by tools/gather_probe.hip).
    `frac` is the work-based VALU figure: flops of the slab and triangle tests the rays need (25 per box,
    55 per triangle) against the fp32 vector peak for plain multiplies and adds (no fused multiply-add: bit
    parity with the reference forbids contraction). It factors into the counter readings as
    frac = valu_busy x live_lane_frac x useful_share x (unpacked / packed issue = 1/2)."""
    s = kernel_ms * 1e-3
    pmc = prof["pmc"].get(pmc_key, {}) if pmc_key else {}
    ceil = prof["ceilings"]
    flops = 4 * SLAB_FLOPS * nodes + TRI_FLOPS * tris
    tflops = flops / s / 1e12
    traffic = pmc.get("hbm_bytes_per_launch")
    lane_ops = pmc.get("SQ_THREAD_CYCLES_VALU")
    rays_pmc = pmc.get("rays_per_launch")
    l1 = pmc.get("l1_lookups_per_clk_per_cu")
    l1_ceiling = ceil.get("l1_lookups_per_clk_per_cu")
    r = {"bound": "valu", "achieved": round(tflops, 2), "peak": VALU_PEAK_NO_FMA_TFLOPS, "unit": "TFLOP/s",
         "frac": round(tflops / VALU_PEAK_NO_FMA_TFLOPS, 4), "traffic": traffic,
         "peak_spec": VALU_PEAK_SPEC_TFLOPS, "frac_vs_spec_peak": round(tflops / VALU_PEAK_SPEC_TFLOPS, 4),
         "peak_note": "peak = fp32 vector rate for un-fused multiplies and adds (bit parity with the reference forbids multiply-add contraction); peak_spec = the data-sheet rate, which counts a fused multiply-add as two flops",
         "kernel": "k_closest", "kernel_ms": round(kernel_ms, 4),
         "flops_per_ray": round(flops / n, 1), "nodes_per_ray": round(nodes / n, 2), "tris_per_ray": round(tris / n, 2),
         "leaf_fetches_per_ray": round(groups / n, 2)}
    lanes = 16.0 * (NODE_LOADS * nodes + GROUP_LOADS * groups) / s / 1e9
    requested = (n * (RAY_BYTES + HIT_BYTES) + nodes * NODE_BYTES + tris * TRI_BYTES) / s / 1e9
    if pmc:
        cms = pmc.get("duration_ns", 0) / 1e6
        r.update({"valu_busy": pmc.get("valu_busy"), "live_lane_frac": pmc.get("live_lane_frac"),
                  "useful_share_of_lane_ops": round(flops / n / (lane_ops / rays_pmc), 3) if lane_ops and rays_pmc else None,
                  "l1": {"lookups_per_clk_per_cu": l1, "ceiling": l1_ceiling, "frac": round(l1 / l1_ceiling, 3) if l1 and l1_ceiling else None,
                         "ta_busy": pmc.get("ta_busy"), "lane_loads_16B_gbs": round(lanes, 1)},
                  "hbm": {"bytes_per_launch": traffic, "gbs": round(traffic / s / 1e9, 1) if traffic else None, "peak": HBM_PEAK_GBS,
                          "frac": round(traffic / s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None, "l2_hit": pmc.get("l2_hit"),
                          "requested_at_cache_level_gbs": round(requested, 1)},
                  "counters_from": pmc.get("source"), "counters_kernel_ms": round(cms, 4) if cms else None,
                  "counters_stale": bool(cms) and abs(cms - kernel_ms) > 0.1 * kernel_ms})
    else:
        # no counter run of this launch is committed: only what this run measures
        r.update({"l1": {"lane_loads_16B_gbs": round(lanes, 1)}, "hbm": {"requested_at_cache_level_gbs": round(requested, 1)}})
    return r


def closest_two_in_flight(pkg, lib, core, scene, cam, rays, hits, n, torch, dev, args, barrier):
    """The headline launch with a second context tracing the same rays into a buffer of its own on its own stream,
    launches alternating between the two: what a host that has independent batches gets when the end of one launch
    (a few long rays, most lanes idle) overlaps the start of the next. Reported beside `value`, never as it:
    `value` is one launch after the other on one stream."""
    other = pkg.Core(torch.cuda.current_device(), lib)
    other.upload_scene(scene)
    other.set_camera(cam)
    hits2 = torch.empty_like(hits)
    cores, outs = (core, other), (hits, hits2)
    for i in range(4):
        cores[i & 1].trace_closest_device(rays.data_ptr(), n, outs[i & 1].data_ptr())
    barrier()
    steps = max(4, args.steps & ~1)
    t0 = time.perf_counter()
    for i in range(steps):
        cores[i & 1].trace_closest_device(rays.data_ptr(), n, outs[i & 1].data_ptr())
    barrier()
    dt = time.perf_counter() - t0
    same = bool(torch.equal(hits, hits2))
    other.close()
    return {"mrays_per_s": round(n * steps / dt / 1e6, 1), "launches": steps, "ms_per_launch": round(dt / steps * 1e3, 4), "streams": 2,
            "equal_hit_buffers": same}


def closest_extra(pkg, lib, core, torch, dev, scene, frm, to, fov, width, height, args, prof, pmc_key, barrier):
    """Closest hit over the pixel-centre primary rays of a width x height film, on `scene` (None: the scene the
    core holds): Mrays/s by HIP events, work per ray and the same roofline block."""
    if scene is not None:
        core.upload_scene(scene)
    core.set_camera(pkg.look_at(lib, frm, to, (0, 1, 0), fov, width, height))
    n = width * height
    rays = torch.empty(n * RAY_BYTES, dtype=torch.uint8, device=dev)
    hits = torch.empty(n * HIT_BYTES, dtype=torch.uint8, device=dev)
    core.generate_primary_rays_device(rays.data_ptr(), 0, n)
    torch.cuda.synchronize()
    steps = max(3, args.steps // 4)
    _, kernel_ms = time_closest(core, rays, hits, n, steps, 2, barrier)
    nodes, tris, groups = core.count_closest_work(rays.data_ptr(), n)
    st = core.stats()
    return {"mrays_per_s": round(n / kernel_ms / 1e3, 1), "rays": n, "steps": steps, "bvh4_nodes": st.bvh4_nodes,
            "roofline": roofline(n, nodes, tris, groups, kernel_ms, prof, pmc_key)}


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
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of these machines supports dmabuf IPC only, and RCCL's exchange of
        # device buffers between the rank processes fails without it (hipIpcGetMemHandle: invalid argument). The image
        # exports it already; it is repeated here so that a launch from a scrubbed environment behaves the same.
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # a rank that dies (no such GPU, a HIP error) leaves the others waiting in a collective for ever: the first failure
    # ends the remaining ranks
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code)
                for other in live:
                    other.terminate()
    return rc


def ab_variants(pkg, lib, scene, cam, args, rays, hits, n, torch, local_rank):
    """Interleaved rounds of the closest-hit kernel under different RTX_VARIANT schedules
    (one context each, same rays), median and min kernel ms per variant."""
    cores = {}
    if args.ab_env:
        name, values = args.ab_env.split("=")
        os.environ["RTX_VARIANT"] = args.variants.split(",")[0]
        for v in values.split(","):
            os.environ[name] = v
            c = pkg.Core(local_rank, lib)
            c.upload_scene(scene); c.set_camera(cam)
            cores[f"{name}={v}"] = c
        del os.environ[name]
    else:
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_has(flag):
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return flag in line.split()
    except OSError:
        pass
    return False


ORACLE_FLAGS = {"O2": "g++ -std=c++17 -O2 -ffp-contract=off -fno-fast-math -pthread", "O3": "g++ -std=c++17 -O3 -march=x86-64-v3 -ffp-contract=off -fno-fast-math -pthread"}


def cpu_baseline(pkg, scene, cam, args):
    """The CPU oracle (a port of the reference's BVH traversal, RTBase/Geometry.h:506-566, and of renderMT,
    RTBase/Renderer.h:753-782) on this box's host cores, over the rays of the same batch, traced three times
    (~15-20 s of CPU work summed over the threads). Threads: every hardware thread this job may use, which is what
    the reference does (numThreads = min(settings.numThreads, numProcs), RTBase/Renderer.h:92) - on a GPU box that is
    the CPU share of one GPU (OMP_NUM_THREADS there), not the whole machine's count. Two builds of the same source: -O2
    (the build the parity tests check, the value of record) and -O3 for AVX2 (the reference's Release configuration
    targets AVX2), reported beside it."""
    pyoracle = importlib.import_module("oracle.pyoracle")
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 8)
    ys, xs = np.mgrid[0:args.height, 0:args.width]
    xy = np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], 1).astype(np.float32)
    builds = [("O2", pyoracle.LIB)]
    if os.path.exists(pyoracle.LIB_O3) and cpu_has("avx2") and cpu_has("fma") and cpu_has("bmi2"):
        builds.append(("O3", pyoracle.LIB_O3))
    res = {}
    for tag, path in builds:
        orc = pyoracle.Oracle(pkg, path)
        osc = orc.scene(scene)
        osc.set_camera(cam)
        rays = osc.generate_rays(xy)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            osc.trace_closest(rays, threads=threads)
        dt = time.perf_counter() - t0
        osc.close()
        r = {"value": round(reps * len(rays) / dt / 1e6, 3), "unit": "Mrays/s", "compiler_flags": ORACLE_FLAGS[tag]}
        if not args.no_pathtrace:
            # the port of renderMT (per-thread MT19937 streams) on the path-trace side scene
            scenes = importlib.import_module("ray-tracer_amd.scenes")
            mixed = scenes.mixed_scene(128)
            osc = orc.scene(mixed)
            osc.set_camera(orc.look_at(scenes.mixed_camera()["frm"], scenes.mixed_camera()["to"], scenes.mixed_camera()["up"], scenes.mixed_camera()["fov"], args.width, args.height))
            passes = 4
            ren = osc.renderer(pkg.make_settings(total_spp=passes, max_bounces=5, use_mis=1), threads=threads, counter_rng=False)
            t0 = time.perf_counter()
            ren.render_passes(passes)
            dt = time.perf_counter() - t0
            ren.close(); osc.close()
            r["pathtrace_spp_per_s"] = round(passes / dt, 4)
            r["pathtrace_sample"] = f"{passes} passes of the {args.width}x{args.height} mixed-material scene"
        res[tag] = r
    out = dict(res["O2"])
    out.update({"cores": threads, "kind": "port", "cpu_model": cpu_model(), "hardware_threads_of_the_machine": os.cpu_count(),
                "sample": f"the whole batch ({args.width * args.height} rays) 3 times, {threads} threads, BVH build excluded"})
    if "O3" in res:
        out["o3_avx2_build"] = res["O3"]
    return out


# algorithmic bytes of path state a sample moves through HBM-side memory between the kernels of a pass. Per sample: the
# closest-hit kernel of depth 0 generates the camera ray where it fetches it and writes the hit (20), the shade kernel of
# depth 0 reads the hit (20) and starts the radiance sum (12), the film kernel reads it and updates the pixel (12 + 24).
# Per closest-hit ray the shade kernel reads the triangle's shading record (96) and a texel (12) and writes what a continued
# path needs (ray 24, throughput 12, generator / mis / flags 16, queue 4). Per closest-hit ray BEHIND depth 0 the traversal
# also reads queue and ray and writes the hit (4 + 24 + 20) and the shade kernel reads the slot (queue, hit, ray,
# throughput, radiance, generator, mis, flags: 88). Per shadow ray the shade kernel writes and the traversal reads segment,
# value and path (40 each way) and adds to the radiance (24).
PT_BYTES_PER_SAMPLE, PT_BYTES_PER_CLOSEST, PT_BYTES_PER_BOUNCE_RAY, PT_BYTES_PER_SHADOW = 20 + 20 + 12 + 36, 96 + 12 + 56, 48 + 88, 80 + 24


def pathtrace_side(pkg, scenes, lib, core, torch, dist, world, rank, dev, args, all_reduce, prof):
    """1080p path tracing of the mixed-material scene under the environment map (benchmark configuration 2).
    With several ranks the samples are split (shard_mode 2) and the films summed with one all-reduce at the end,
    through the same driver a multi-GPU render uses (multigpu.render_sharded)."""
    try:
        mg = importlib.import_module("ray-tracer_amd.multigpu")
        scene = scenes.mixed_scene(128)
        core.upload_scene(scene)
        cam = pkg.look_at(lib, width=args.width, height=args.height, **scenes.mixed_camera())
        core.set_camera(cam)
        # weak scaling by default (every rank adds --pt-spp passes of its own); --pt-total-spp fixes the total instead (strong)
        total = args.pt_total_spp if args.pt_total_spp else args.pt_spp * world
        scaling = "strong" if args.pt_total_spp else "weak"
        base = pkg.make_settings(total_spp=total, seed=1)
        core.set_settings(mg.shard_settings(base, rank, world, "spp"))
        target = mg.CoreTarget(core, torch)
        core.clear()
        core.render_passes(4)          # warm-up (also brings up every pipeline lane's buffers)
        core.clear()
        if world > 1:
            all_reduce(torch.zeros(core.film_device()[1], dtype=torch.float32, device=dev))     # warm-up of the collective at film size
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mg.render_sharded(target, base, dist, rank, world, "spp")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            all_reduce(t, dist.ReduceOp.MAX)
            dt = float(t.item())
        out = {"pathtrace_spp_per_s": round(total / dt, 3),
               "pathtrace_config": {"workload": f"configs[2]: mixed Diffuse/Mirror/Glass/Conductor/Plastic scene ({scene.n_triangles} triangles) + 1024x512 env-map MIS, {args.width}x{args.height}, maxBounces 5",
                                    "spp_timed": total, "scaling": scaling,
                                    "parallelism": f"spp split x{world} + one film all-reduce ({core.film_device()[1] * 4} bytes per rank) at the end, inside the timed region" if world > 1 else "1 GPU"}}
        if rank == 0 and world == 1:
            # what a host that asks for one pass per call gets (RayTracer::renderMT's contract): no passes in flight beside each other
            core.set_settings(base); core.clear(); core.render_passes(3)
            t1 = time.perf_counter()
            for _ in range(16):
                core.render_passes(1)
            out["pathtrace_ms_one_pass_per_call"] = round((time.perf_counter() - t1) / 16 * 1e3, 4)
            out["pathtrace_roofline"] = pathtrace_roofline(core, dt / total, prof)
            if not args.no_radiosity:
                out["light_tracing"] = light_tracing_side(pkg, core, args)
                out["instant_radiosity"] = radiosity_side(pkg, core, args)
        return out
    except pkg.RtxError as e:
        return {"pathtrace_spp_per_s": None, "pathtrace_error": str(e)}


def light_tracing_side(pkg, core, args):
    """Light tracing (RayTracer::lightTrace, RTBase/Renderer.h:379-411) on the scene the core holds: one light path per
    pixel slot and pass, every vertex connected to the camera. Passes per second with passes in flight and the
    milliseconds of a pass when the host asks for one per call; rays per pass from two instrumented passes."""
    core.set_settings(pkg.make_settings(algorithm=pkg.AL_LIGHT_TRACE, total_spp=1 << 20, max_bounces=5, seed=1))
    core.clear(); core.render_passes(3); core.clear()
    passes = 64
    t0 = time.perf_counter()
    core.render_passes(passes)
    dt = (time.perf_counter() - t0) / passes
    core.clear()
    t0 = time.perf_counter()
    for _ in range(16):
        core.render_passes(1)
    single = (time.perf_counter() - t0) / 16
    core.clear()
    w = core.count_render_work(2)
    return {"passes_per_s": round(1.0 / dt, 1), "ms_per_pass": round(dt * 1e3, 4), "ms_per_pass_one_in_flight": round(single * 1e3, 4),
            "closest_rays_per_pass": w["closest_rays"] / 2, "connection_rays_per_pass": w["shadow_rays"] / 2,
            "mrays_per_s": round((w["closest_rays"] + w["shadow_rays"]) / 2 / dt / 1e6, 1), "resolution": [args.width, args.height]}


def radiosity_side(pkg, core, args):
    """Instant radiosity on the scene the core holds (the mixed-material scene at the path-trace leg's resolution), 128
    light paths per pass (numThreads 8 x vplRaysPerTile 16, RTBase/Renderer.h:174,283): every gathering pixel connects to
    every valid VPL of the pass (RTBase/Renderer.h:210-245). The work figures come from two instrumented passes; the
    fraction prices the slab and triangle tests of all rays of a pass (camera rays, light paths, gather segments) at the
    un-fused fp32 vector peak against the measured time of a pass."""
    core.set_settings(pkg.make_settings(algorithm=pkg.AL_INSTANT_RADIOSITY, total_spp=1 << 20, max_bounces=5, num_threads=8, vpl_rays_per_tile=16, seed=1))
    core.clear(); core.render_passes(3); core.clear()
    passes = 24
    t0 = time.perf_counter()
    core.render_passes(passes)
    dt = (time.perf_counter() - t0) / passes
    core.clear()
    t0 = time.perf_counter()
    for _ in range(8):
        core.render_passes(1)
    single = (time.perf_counter() - t0) / 8
    core.clear()
    w = core.count_render_work(2)
    rays = max(1, w["closest_rays"] + w["shadow_rays"])
    flops = (4 * SLAB_FLOPS * w["nodes"] + TRI_FLOPS * w["tris"]) / 2
    return {"passes_per_s": round(1.0 / dt, 1), "ms_per_pass": round(dt * 1e3, 4), "ms_per_pass_one_in_flight": round(single * 1e3, 4),
            "light_paths_per_pass": 128, "valid_vpls_per_pass": w["vpls"] / 2, "gather_segments_per_pass": w["gather_segments"] / 2,
            "gather_gsegments_per_s": round(w["gather_segments"] / 2 / dt / 1e9, 2),
            "nodes_per_ray": round(w["nodes"] / rays, 2), "tris_per_ray": round(w["tris"] / rays, 2),
            "traversal_flops_per_pass": int(flops), "useful_tflops": round(flops / dt / 1e12, 2), "peak": VALU_PEAK_NO_FMA_TFLOPS,
            "frac": round(flops / dt / 1e12 / VALU_PEAK_NO_FMA_TFLOPS, 4), "kernel": "k_ir_trace_packet",
            "resolution": [args.width, args.height]}


def pathtrace_roofline(core, s_per_pass, prof):
    """A pass is a chain of kernels bound by two different things: the traversal launches like the headline kernel
    (vector-ALU issue and L1 look-ups), ray generation / shading / film by HBM-side traffic of path state. The floor of
    a pass is the sum of both resource times for the work two instrumented passes count (traversal flops at the
    un-fused fp32 vector peak + algorithmic state bytes at 8 TB/s); frac = floor / measured."""
    core.clear()
    w = core.count_render_work(2)
    k = w["samples"]
    per = {key: w[key] / k for key in ("closest_rays", "shadow_rays", "nodes", "tris", "tri_groups")}
    flops = (4 * SLAB_FLOPS * w["nodes"] + TRI_FLOPS * w["tris"]) / 2
    state_bytes = (PT_BYTES_PER_SAMPLE * w["samples"] + PT_BYTES_PER_CLOSEST * w["closest_rays"] + PT_BYTES_PER_BOUNCE_RAY * max(0, w["closest_rays"] - w["samples"])
                   + PT_BYTES_PER_SHADOW * w["shadow_rays"]) / 2
    pmc = prof["pmc"].get("pathtrace", {})
    r = {"bound": "valu (traversal kernels) + hbm (shading, film)", "ms_per_pass": round(s_per_pass * 1e3, 4),
         "rays_per_sample": {"closest": round(per["closest_rays"], 3), "shadow": round(per["shadow_rays"], 3)},
         "nodes_per_sample": round(per["nodes"], 2), "tris_per_sample": round(per["tris"], 2),
         "traversal_flops_per_pass": int(flops), "state_bytes_per_pass": int(state_bytes),
         "useful_tflops": round(flops / s_per_pass / 1e12, 2), "state_gbs": round(state_bytes / s_per_pass / 1e9, 1)}
    floor = flops / (VALU_PEAK_NO_FMA_TFLOPS * 1e12) + state_bytes / (HBM_PEAK_GBS * 1e9)
    r.update({"floor_ms_per_pass": round(floor * 1e3, 4), "frac": round(floor / s_per_pass, 4),
              "valu_peak_no_fma_tflops": VALU_PEAK_NO_FMA_TFLOPS, "hbm_peak_gbs": HBM_PEAK_GBS})
    if pmc:
        r["kernels"] = pmc.get("kernels")
        r["hbm_bytes_per_pass"] = pmc.get("hbm_bytes_per_pass")
        r["counters_from"] = pmc.get("source")
    return r


if __name__ == "__main__":
    main()
