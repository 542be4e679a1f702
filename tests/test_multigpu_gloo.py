"""world_size-2 test of the multi-GPU partitioning on CPU (gloo): each rank renders its shard
with the CPU oracle under the same shard settings the HIP core uses, the films are combined
by the package's own reduce (ray-tracer_amd/multigpu.py), and the result must equal the
unsharded film: bit-identical for tile sharding, equal up to float summation order for the
sample split."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("ray-tracer_amd")
    scenes = importlib.import_module("ray-tracer_amd.scenes")
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    pyoracle = importlib.import_module("oracle.pyoracle")
    orc = pyoracle.Oracle(pkg)
    sc = scenes.cornell_box("mixed")
    cam = orc.look_at((0, 0, 3.4), (0, 0, 0), (0, 1, 0), 45.0, 48, 40)
    osc = orc.scene(sc); osc.set_camera(cam)
    total = 5
    st = mg.shard_settings(pkg.make_settings(total_spp=total, seed=17), rank, world, mode)
    assert (st.shard_rank, st.shard_world) == (rank, world)
    r = osc.renderer(st, threads=2, counter_rng=True)
    n = mg.passes_for_rank(total, rank, world, mode)
    r.render_passes(n)
    film, spp, _ = r.film()
    t = torch.from_numpy(film.copy())
    mg.reduce_film(t, dist, world)
    spp = mg.reduced_spp(spp, world, mode, dist, torch, torch.device("cpu"))
    if rank == 0:
        np.save(os.path.join(out_dir, f"{mode}.npy"), t.numpy())
        np.save(os.path.join(out_dir, f"{mode}_spp.npy"), np.array([spp]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["tiles", "spp"])
def test_two_rank_partition_matches_single_rank(tmp_path, mode, pkg, orc, scenes):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 400) + (0 if mode == "tiles" else 1)
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / f"{mode}.npy")
    assert int(np.load(tmp_path / f"{mode}_spp.npy")[0]) == 5
    sc = scenes.cornell_box("mixed")
    cam = orc.look_at((0, 0, 3.4), (0, 0, 0), (0, 1, 0), 45.0, 48, 40)
    osc = orc.scene(sc); osc.set_camera(cam)
    r = osc.renderer(pkg.make_settings(total_spp=5, seed=17), threads=4, counter_rng=True)
    r.render_passes(5)
    want = r.film()[0]
    if mode == "tiles":
        assert np.array_equal(got, want)
    else:
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6)


def test_shard_helpers(pkg):
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    assert [mg.passes_for_rank(10, r, 4, "spp") for r in range(4)] == [3, 3, 2, 2]
    assert mg.passes_for_rank(10, 1, 4, "tiles") == 10
    s = mg.shard_settings(pkg.make_settings(), 0, 1, "spp")
    assert (s.shard_mode, s.shard_world) == (0, 1)
