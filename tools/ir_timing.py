import importlib, time, sys
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("ray-tracer_amd"); scenes = importlib.import_module("ray-tracer_amd.scenes")
lib = pkg.load_library(); core = pkg.Core(0, lib)
for name, sc, cam in (("cornell", scenes.cornell_box("diffuse"), scenes.cornell_camera()), ("mixed", scenes.mixed_scene(128), scenes.mixed_camera())):
    core.upload_scene(sc); core.set_camera(pkg.look_at(lib, width=1920, height=1080, **cam))
    for paths in (4, 22):
        core.set_settings(pkg.make_settings(algorithm=pkg.AL_INSTANT_RADIOSITY, total_spp=8, max_bounces=5, vpl_rays_per_tile=1, num_threads=paths))
        core.clear(); core.render_passes(1); core.clear(); core.reset_stats()
        t = time.time(); core.render_passes(4); dt = (time.time() - t) / 4
        st = core.stats()
        print(name, "light paths", paths, "ms/pass %.2f" % (dt * 1e3), {k: getattr(st, k) for k, _ in st._fields_ if "ray" in k or "vpl" in k}, flush=True)
