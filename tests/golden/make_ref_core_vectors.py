"""Generates tests/golden/ref_core_vectors.npz from the REAL reference headers
RTBase/Core.h + RTBase/Sampling.h (compiled into oracle/_ref/libref_core.so by
oracle/Makefile, which needs /root/reference). Inputs are seeded; outputs are what the
reference's own code returns: matrix inverse / lookAt / perspective / point and vector
transforms, Frame::fromVector + toLocal + toWorld, the three sampling warps with their pdfs,
Colour::Lum, the first values of MTRandom for the seeds RayTracer uses, the warps called as
f(sampler->next(), sampler->next()) on a real MTRandom (which draw lands in which parameter),
and SphericalCoordinates.

Run from the repository root:  python tests/golden/make_ref_core_vectors.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ref = importlib.import_module("oracle.pyoracle").RefCore()
rs = np.random.RandomState(2025)

out = {}
mats = []
for _ in range(16):
    m = rs.uniform(-2, 2, 16).astype(np.float32)
    mats.append(m)
# camera-like matrices too
for frm, to in (((0, 0, 3.5), (0, 0, 0)), ((0, 2.2, 7.5), (0, 0.8, 0)), ((3, 4, -5), (0.5, 0.25, 0.125))):
    mats.append(ref.look_at(frm, to, (0, 1, 0)))
    mats.append(ref.perspective(0.001, 10000.0, 16 / 9, 45.0))
mats = np.array(mats, np.float32)
out["matrices"] = mats
out["inverse"] = np.array([ref.invert(m) for m in mats], np.float32)
la_in = np.array([[0, 0, 3.5, 0, 0, 0, 0, 1, 0], [0, 2.2, 7.5, 0, 0.8, 0, 0, 1, 0], [3, 4, -5, 0.5, 0.25, 0.125, 0.1, 0.9, 0.2]], np.float32)
out["lookat_in"] = la_in
out["lookat"] = np.array([ref.look_at(r[0:3], r[3:6], r[6:9]) for r in la_in], np.float32)
p_in = np.array([[0.001, 10000.0, 16 / 9, 45.0], [0.001, 10000.0, 1.0, 40.0], [0.1, 100.0, 1.2, 60.0]], np.float32)
out["persp_in"] = p_in
out["persp"] = np.array([ref.perspective(*[float(x) for x in r]) for r in p_in], np.float32)
vecs = rs.uniform(-3, 3, (len(mats), 3)).astype(np.float32)
out["vecs"] = vecs
out["mul_point"] = np.array([ref.mul_point(m, v) for m, v in zip(mats, vecs)], np.float32)
out["mul_vec"] = np.array([ref.mul_vec(m, v) for m, v in zip(mats, vecs)], np.float32)
out["mul_point_divide"] = np.array([ref.mul_point_divide(m, v) for m, v in zip(mats, vecs)], np.float32)
normals = rs.normal(size=(64, 3)).astype(np.float32)
normals[:3] = [[0, 0, 1], [1, 0, 0], [0, -1, 0]]
fvec = rs.normal(size=(64, 3)).astype(np.float32)
out["frame_n"], out["frame_v"] = normals, fvec
out["frame"] = np.array([ref.frame(n, v) for n, v in zip(normals, fvec)], np.float32)
u = rs.uniform(0, 1, (128, 2)).astype(np.float32)
u[0] = [1e-8, 1e-8]; u[1] = [0.99999994, 0.5]
out["warp_u"] = u
for kind, name in ((0, "cosine_hemisphere"), (1, "uniform_sphere"), (2, "uniform_hemisphere")):
    out["warp_" + name] = np.array([ref.warp(kind, float(a), float(b)) for a, b in u], np.float32)
cols = rs.uniform(0, 20, (32, 3)).astype(np.float32)
out["lum_in"] = cols
out["lum"] = np.array([ref.lib.ref_lum(c.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_float))) for c in cols], np.float32)
out["mt_seeds"] = np.array([48271, 96542, 1], np.uint32)
out["mt_stream"] = np.array([ref.mt_stream(int(s), 64) for s in out["mt_seeds"]], np.float32)
# f(sampler->next(), sampler->next()) on a real MTRandom under g++: the parameter each draw lands in
out["wfs_seeds"] = np.array([48271, 96542, 1, 7, 123456789], np.uint32)
for kind, name in ((0, "cosine_hemisphere"), (1, "uniform_sphere"), (2, "uniform_hemisphere")):
    out["wfs_" + name] = np.array([ref.warp_from_sampler(kind, int(s)) for s in out["wfs_seeds"]], np.float32)
# SphericalCoordinates (Core.h:557-572)
sph = rs.uniform(0, 1, (64, 2)).astype(np.float32) * np.array([np.pi, 2 * np.pi], np.float32)
swi = rs.normal(size=(64, 3)).astype(np.float32); swi /= np.linalg.norm(swi, axis=1, keepdims=True)
swi[:4] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [-1, -1e-8, 0]]
out["sph_in"], out["sph_wi"] = sph, swi.astype(np.float32)
out["sph"] = np.array([ref.spherical(float(a), float(b), w) for (a, b), w in zip(sph, out["sph_wi"])], np.float32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_core_vectors.npz"), **out)
print("wrote", {k: v.shape for k, v in out.items()})
