"""The closest-hit entry point that allocates nothing (rtx_trace_closest_device_noalloc: work counters and
traversal-stack spill in a static device array) against rtx_trace_closest_device on the same rays: hit
records bit for bit, on two calls in a row, the second with a smaller input (the static array keeps the
first call's contents). Also its limits: a scene deeper than the static stack holds is refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_rays(pkg, torch, n, seed):
    rs = np.random.RandomState(seed)
    rays = np.empty(n, pkg.RAY_DTYPE)
    rays["o"] = rs.uniform(-1.2, 1.2, (n, 3)).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32)
    d[rs.rand(n) < 0.02, 0] = 0.0                     # some rays take the exact (unpruned, binary-tree) path
    rays["d"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()


def test_noalloc_closest_hit_equals_allocating_entry_point(pkg, lib, core, scenes):
    import torch
    core.upload_scene(scenes.soup(200_000, seed=71, size=0.03))
    for n, seed in ((700_001, 1), (5_003, 2)):        # second call smaller than the first
        rays = _device_rays(pkg, torch, n, seed)
        want = torch.empty(n * pkg.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        got = torch.full_like(want, 0xAB)
        torch.cuda.synchronize()
        core.trace_closest_device(rays.data_ptr(), n, want.data_ptr())
        core.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        core.trace_closest_device_noalloc(rays.data_ptr(), n, got.data_ptr())
        core.synchronize()
        # what the entry point exists for: the call takes no device memory (hipMemGetInfo sees every hipMalloc of the process)
        assert torch.cuda.mem_get_info()[0] == free_before, "the no-allocation entry point allocated device memory"
        w = want.cpu().numpy().view(pkg.HIT_DTYPE); g = got.cpu().numpy().view(pkg.HIT_DTYPE)
        assert (w["id"] != 0xFFFFFFFF).sum() > n // 10
        assert np.array_equal(g.view(np.uint8), w.view(np.uint8)), f"{(g['id'] != w['id']).sum()} of {n} hit ids differ"
    # n == 0 is a no-op, null buffers are refused
    core.trace_closest_device_noalloc(0, 0, 0)
    with pytest.raises(pkg.RtxError):
        core.trace_closest_device_noalloc(0, 10, 0)


def test_noalloc_refuses_a_tree_deeper_than_its_static_stack(pkg, lib, core, scenes):
    """scenes.deep_chains() builds a caterpillar tree that can need 44 stack entries per ray; the entry
    point must refuse it with a message instead of overrunning the static array."""
    import torch
    sc = scenes.deep_chains()
    core.upload_scene(sc)
    st = core.stats()
    need = max(3 * st.bvh4_depth + 2, st.bvh2_depth + 3)
    rays = _device_rays(pkg, torch, 1000, 3)
    hits = torch.empty(1000 * pkg.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert need > 42
    with pytest.raises(pkg.RtxError, match="traversal-stack entries"):
        core.trace_closest_device_noalloc(rays.data_ptr(), 1000, hits.data_ptr())
    core.trace_closest_device(rays.data_ptr(), 1000, hits.data_ptr())      # the allocating entry point takes it
    core.synchronize()
