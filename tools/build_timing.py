import importlib, time, os, sys
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("ray-tracer_amd"); scenes = importlib.import_module("ray-tracer_amd.scenes")
lib = pkg.load_library()
core = pkg.Core(0, lib)
for name, mk in [("soup1M", lambda: scenes.soup(1_000_000)), ("terrain10M", lambda: scenes.terrain(2236))]:
    sc = mk()
    for mode in ("device", "host", "device"):
        os.environ["RTX_BVH_BUILD"] = mode
        t = time.time(); core.upload_scene(sc); dt = time.time() - t
        st = core.stats()
        print(name, mode, "upload %.2fs" % dt, {k: getattr(st, k) for k, _ in st._fields_ if "ms" in k or "bvh" in k}, flush=True)
