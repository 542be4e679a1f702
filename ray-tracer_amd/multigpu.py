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

Adaptive sampling needs the variance of the COMPLETE film, which is not linear in the
per-rank sums: reduce the film when SPP reaches initSPP, then let every rank compute the same
tile targets from it (``sync_adaptive``).
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
    """How many local passes rank must render so that the ranks together cover total_spp."""
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


def reduce_film(film, dist, world: int):
    """Sum the film accumulators of all ranks in place (film: any torch tensor)."""
    if world > 1:
        dist.all_reduce(film, op=dist.ReduceOp.SUM)
    return film


def reduced_spp(local_spp: int, world: int, mode: str, dist=None, torch=None, device=None) -> int:
    """Sample count of the reduced film: unchanged for tile sharding, summed for spp split."""
    if world <= 1 or mode == "tiles":
        return local_spp
    t = torch.tensor([local_spp], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def sync_adaptive(core, torch, dist, world: int, mode: str):
    """After the pass in which the global sample count reaches initSPP: sum the films so
    that every rank sees the complete film, recompute the tile targets from it on every rank
    (calculateTileSamples, RTBase/Renderer.h:643-671; identical input, identical targets),
    then put the rank's own partial film back so that the final reduction still counts each
    sample once. With the spp split the targets are in global sample indices."""
    if world <= 1:
        return
    film = film_tensor(core, torch)
    local = film.clone()
    dist.all_reduce(film, op=dist.ReduceOp.SUM)
    if mode == "spp":
        spp = reduced_spp(core.spp(), world, mode, dist, torch, film.device)
        keep = core.spp()
        core.set_spp(spp)
        core.update_tile_samples()
        core.set_spp(keep)
    else:
        core.update_tile_samples()
    core.synchronize()
    torch.cuda.synchronize()
    film.copy_(local)
