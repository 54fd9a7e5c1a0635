"""x-slab sharding of one grid over the GPUs of a node + the single gather that
assembles it.

Grid points are independent, so the only exchange is the final assembly: the array is
``[channel, x, y, z]`` (C order) = Julia's column-major ``[z, y, x, channel]``
(grids.jl:126-133); rank r owns x-planes ``slab_range(nx, world, r)``, i.e. one contiguous
block per channel.  With ``nx % world == 0`` the gather is 8 ``all_gather_into_tensor``
calls (one per channel) straight into the final array -- no staging copy.  Otherwise the
slabs are padded to the largest one and copied into place after one all-gather.

One process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on ROCm;
``gloo`` in the CPU tests).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def slab_range(nx: int, world_size: int, rank: int) -> Tuple[int, int]:
    """x-planes [begin, end) owned by ``rank`` (same split as the one-shot C entry points)."""
    base, rem = divmod(nx, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def allgather_grid(full: torch.Tensor, local: torch.Tensor, group=None) -> torch.Tensor:
    """Assemble ``full[8, nx, ny, nz]`` on every rank from the ranks' slabs.

    ``local`` is this rank's ``[8, n_local, ny, nz]`` contiguous slab (``n_local`` from
    :func:`slab_range`).  Returns ``full``.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    nchan, nx = full.shape[0], full.shape[1]
    b, e = slab_range(nx, world, rank)
    assert local.shape[0] == nchan and local.shape[1] == e - b and local.shape[2:] == full.shape[2:]
    assert full.is_contiguous() and local.is_contiguous()
    if world == 1:
        if local.data_ptr() != full.data_ptr():
            full.copy_(local)
        return full
    backend = dist.get_backend(group)
    if nx % world == 0:
        for c in range(nchan):
            if backend == "gloo":
                outs = [full[c, slab_range(nx, world, r)[0]:slab_range(nx, world, r)[1]] for r in range(world)]
                dist.all_gather(outs, local[c], group=group)
            else:
                dist.all_gather_into_tensor(full[c], local[c], group=group)
        return full
    nmax = -(-nx // world)
    padded = local.new_zeros((nchan, nmax) + tuple(local.shape[2:]))
    padded[:, : e - b] = local
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    for r in range(world):
        rb, re = slab_range(nx, world, r)
        full[:, rb:re] = gathered[r][:, : re - rb]
    return full
