import importlib, os, sys, json, time
import numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("ray-tracer_amd"); scenes = importlib.import_module("ray-tracer_amd.scenes")
lib = pkg.load_library()
films = {}
for cull in ("1", "0", "1"):
    os.environ["RTX_IR_BLOCK_CULL"] = cull
    core = pkg.Core(0, lib)
    core.upload_scene(scenes.mixed_scene(128)); core.set_camera(pkg.look_at(lib, width=1920, height=1080, **scenes.mixed_camera()))
    core.set_settings(pkg.make_settings(total_spp=1 << 20, seed=1, algorithm=2, num_threads=8, vpl_rays_per_tile=16, max_bounces=5))
    core.clear(); core.render_passes(2); core.clear()
    t0 = time.perf_counter()
    for _ in range(8): core.render_passes(1)
    single = (time.perf_counter() - t0) / 8 * 1e3
    core.clear(); t0 = time.perf_counter(); core.render_passes(24); pip = (time.perf_counter() - t0) / 24 * 1e3
    f, spp, _ = core.film(); films[cull] = f.copy()
    print(json.dumps({"cull": cull, "single_ms": round(single, 3), "pipelined_ms": round(pip, 3), "film_mean": float(f.mean() / spp)}), flush=True)
    core.close()
same = (films["1"].view(np.uint32) == films["0"].view(np.uint32)).all()
print("films bit-identical:", bool(same), "max abs diff", float(np.abs(films["1"] - films["0"]).max()))
