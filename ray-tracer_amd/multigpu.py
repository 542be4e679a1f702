"""Multi-GPU partitioning of a render: one process per GPU, scene replicated, film
accumulators combined with a torch.distributed all-reduce (RCCL over xGMI on MI355X,
gloo in the CPU tests).

The reference has no multi-device path (its only parallelism is the std::thread tile queue
of RTBase/Renderer.h:677-782); the partitioning follows from what that queue parallelises:

* ``tiles``  — shard_mode 1: 16x16 tile t belongs to rank t % world. Ranks write disjoint
  film pixels, so the sum of the films is bit-identical to a single-GPU render (the counter
  RNG is keyed on pixel and sample, not on the device).
* ``spp``    — shard_mode 2: every rank renders the whole film for the global sample
  indices rank+1, rank+1+world, ...; the films are summed. Needed for light tracing, whose
  splats land anywhere on the film.

Adaptive sampling (calculateTileSamples, RTBase/Renderer.h:643-671, run once when SPP reaches
initSPP, :780-781) needs the variance of the COMPLETE film of the first initSPP samples:

* tiles: a tile's pixels are complete on its owner, so every rank computes the targets of its own
  tiles and the ranks exchange them (a few KB);
* spp: the variance is not linear in the per-rank sums, so the films are summed when the global
  sample count reaches initSPP, every rank computes the same targets from the sum
  (``sync_adaptive``) and carries on with its own partial film.

``render_sharded`` is the driver that puts the pieces in order; it works on any object with the
small interface of :class:`CoreTarget` (the tests drive the CPU oracle through the same code).
"""
from __future__ import annotations

import importlib

_pkg = importlib.import_module(__name__.rsplit(".", 1)[0])

MODE_NONE, MODE_TILES, MODE_SPP = 0, 1, 2


def shard_settings(settings, rank: int, world: int, mode: str):
    """Copy of ``settings`` for one rank of ``world``. mode: "tiles" or "spp"."""
    s = _pkg.Settings.from_buffer_copy(settings)
    if world <= 1:
        s.shard_mode, s.shard_rank, s.shard_world = MODE_NONE, 0, 1
    else:
        s.shard_mode = {"tiles": MODE_TILES, "spp": MODE_SPP}[mode]
        s.shard_rank, s.shard_world = rank, world
    return s


def passes_for_rank(total_spp: int, rank: int, world: int, mode: str) -> int:
    """How many local passes rank must render so that the ranks together cover the global
    samples 1..total_spp."""
    if world <= 1 or mode == "tiles":
        return total_spp
    return total_spp // world + (1 if rank < total_spp % world else 0)


class _DevArray:
    """Zero-copy view of device memory for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def film_tensor(core, torch):
    """The film accumulator of ``core`` as a torch tensor that aliases the core's memory."""
    ptr, n = core.film_device()
    return torch.as_tensor(_DevArray(ptr, n, "<f4"), device=torch.device("cuda", torch.cuda.current_device()))


def tile_samples_tensor(core, torch):
    ptr, n = core.tile_samples_device()
    return torch.as_tensor(_DevArray(ptr, n, "<i4"), device=torch.device("cuda", torch.cuda.current_device()))


def all_reduce(t, dist, op=None):
    """dist.all_reduce in place; device tensors under the gloo backend (rehearsal of the multi-rank
    control flow on a one-GPU box) go through the host."""
    op = op or dist.ReduceOp.SUM
    if t.is_cuda and dist.get_backend() == "gloo":
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)
    return t


def reduce_film(film, dist, world: int):
    """Sum the film accumulators of all ranks in place (film: any torch tensor)."""
    if world > 1:
        all_reduce(film, dist)
    return film


def reduced_spp(local_spp: int, world: int, mode: str, dist=None, torch=None, device=None) -> int:
    """Sample count of the reduced film: unchanged for tile sharding, summed for spp split."""
    if world <= 1 or mode == "tiles":
        return local_spp
    t = torch.tensor([local_spp], dtype=torch.int64, device=device)
    all_reduce(t, dist)
    return int(t.item())


class CoreTarget:
    """What ``render_sharded`` needs from a renderer, here for a device :class:`Core`.

    ``film()`` / ``tiles()`` return torch tensors that alias the core's device memory, so a
    collective on them works in place; ``fence()`` makes everything torch has enqueued
    (collectives included) complete before the core's own stream — created non-blocking, so
    not ordered against torch's streams by itself — touches the same memory, and
    ``render`` / ``recompute_targets`` return only when the core's work is complete."""

    def __init__(self, core, torch):
        self.core, self.torch = core, torch

    def render(self, n: int):
        if n > 0:
            self.core.render_passes(n)          # blocking

    def spp(self) -> int:
        return self.core.spp()

    def set_spp(self, v: int):
        self.core.set_spp(v)

    def film(self):
        return film_tensor(self.core, self.torch)

    def store_film(self, t):
        f = self.film()
        if f.data_ptr() != t.data_ptr():
            f.copy_(t)
        self.fence()

    def tiles(self):
        return tile_samples_tensor(self.core, self.torch)

    def store_tiles(self, t):
        f = self.tiles()
        if f.data_ptr() != t.data_ptr():
            f.copy_(t)
        self.fence()

    def recompute_targets(self):
        self.core.update_tile_samples()
        self.core.synchronize()

    def fence(self):
        self.torch.cuda.synchronize()


def sync_adaptive(target, dist, world: int, mode: str, rank: int = 0, wide_filter: bool = False):
    """Call on every rank after the pass in which the GLOBAL sample count reaches initSPP.

    tiles: every rank has just computed the targets of all tiles from its own film; only those of
    its own tiles (complete pixels) are right. The others are zeroed and the tables summed.
    spp: sum the films so that every rank sees the complete film of the first initSPP samples,
    recompute the targets from it with SPP = the global count (identical input, identical targets on
    every rank), then put the rank's own partial film back so that the final reduction still counts
    each sample once. Targets are in global sample indices either way.
    wide_filter (Gaussian / Mitchell, whose splats reach into the neighbouring tiles, RTBase/Imaging.h:324-339): no
    tile is complete on its owner, so the tile split sums the films like the sample split."""
    if world <= 1:
        return
    torch = importlib.import_module("torch")
    if mode == "tiles" and not wide_filter:
        tiles = target.tiles()
        owner = torch.arange(tiles.numel(), device=tiles.device) % world
        mine = torch.where(owner == rank, tiles, torch.zeros_like(tiles))
        all_reduce(mine, dist)
        target.store_tiles(mine)
        return
    film = target.film()
    local = film.clone()
    local_spp = target.spp()
    all_reduce(film, dist)
    spp = reduced_spp(local_spp, world, mode, dist, torch, film.device)
    target.store_film(film)          # also the fence: the sum is complete before the core's stream reads it
    target.set_spp(spp)
    target.recompute_targets()       # returns when the targets are written
    target.set_spp(local_spp)
    target.store_film(local)


def render_sharded(target, settings, dist, rank: int, world: int, mode: str, reduce: bool = True):
    """Starts from an empty film (``target.spp() == 0``).
    Render ``settings.total_spp`` GLOBAL samples per pixel across ``world`` ranks and (reduce)
    leave the complete film, SPP = total_spp, on every rank. ``target`` must already carry
    ``shard_settings(settings, rank, world, mode)``. With adaptive sampling the ranks meet once,
    when the global count reaches initSPP (``sync_adaptive``); otherwise the only collective is
    the final film sum."""
    if target.spp() != 0:
        raise RuntimeError(f"render_sharded starts from an empty film: SPP is {target.spp()}, clear it first")
    total = int(settings.total_spp)
    adaptive = bool(settings.adaptive_sampling) and settings.algorithm == _pkg.AL_PATH_TRACE and settings.draw_mode == _pkg.DM_ALGORITHM
    init = int(settings.init_spp)
    n_local = passes_for_rank(total, rank, world, mode)
    if world > 1 and adaptive and init < total:
        n_init = passes_for_rank(init, rank, world, mode)
        target.render(n_init)
        sync_adaptive(target, dist, world, mode, rank, wide_filter=settings.filter != _pkg.FT_BOX)
        target.render(n_local - n_init)
    else:
        target.render(n_local)
    if reduce and world > 1:
        film = target.film()
        all_reduce(film, dist)
        target.store_film(film)
        target.set_spp(total)
