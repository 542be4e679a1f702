#!/usr/bin/env python3
"""Closest-hit rate when rays and hits live in host memory (rtx_trace_closest, host pointers in and
out): what the PCIe crossing costs next to the device-resident rate bench.py reports."""
import importlib, sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("ray-tracer_amd"); scenes = importlib.import_module("ray-tracer_amd.scenes")
lib = pkg.load_library(); core = pkg.Core(0, lib)
core.upload_scene(scenes.soup(1_000_000, seed=12345))
cam = pkg.look_at(lib, (0, 0, 3.5), (0, 0, 0), (0, 1, 0), 45.0, 1920, 1080); core.set_camera(cam)
n = 1920 * 1080
rays = np.empty(n, pkg.RAY_DTYPE)
ys, xs = np.mgrid[0:1080, 0:1920]
d = np.stack([(xs.ravel() - 960) / 1300.0, (540 - ys.ravel()) / 1300.0, -np.ones(n)], 1).astype(np.float32)
rays["o"] = (0, 0, 3.5); rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
core.trace_closest(rays)
t0 = time.perf_counter()
for _ in range(5): h = core.trace_closest(rays)
dt = (time.perf_counter() - t0) / 5
print("host buffers in and out, a fresh output array per call: %.2f ms per call, %.0f Mrays/s" % (dt * 1e3, n / dt / 1e6))
out = np.empty(n, pkg.HIT_DTYPE); core.trace_closest(rays, out)
t0 = time.perf_counter()
for _ in range(10): core.trace_closest(rays, out)
dt = (time.perf_counter() - t0) / 10
print("host buffers in and out, output array reused:          %.2f ms per call, %.0f Mrays/s" % (dt * 1e3, n / dt / 1e6))
assert np.array_equal(out, h)
