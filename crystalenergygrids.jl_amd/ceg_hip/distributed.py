"""x-slab sharding of one grid over the GPUs of a node + the gather that assembles it
(SURVEY §8e).

Grid points are independent, so the only exchange is the final assembly.  The array is
``[channel, x, y, z]`` (C order) = Julia's column-major ``[z, y, x, channel]``
(grids.jl:126-133); a range of x-planes is one contiguous block per channel.

Two distributions of the x-planes:

* **contiguous slabs** (``slab_range``): rank r owns one block of planes; one
  ``all_gather_into_tensor`` per channel straight into the final array when ``nx % world == 0``,
  padded all-gather + copy otherwise.  Same split as the one-shot C entry points.

* **block-cyclic chunks** (``cyclic_plan``): the planes are cut into ``nchunks`` super-blocks of
  ``world*m`` planes, rank r owning the r-th ``m`` planes of every super-block.  For one chunk the
  ranks' pieces are adjacent in the final array, so each chunk is gathered in place *while the
  next chunk is being computed* (collectives on a side stream): the exchange over xGMI hides
  behind the FP64 kernels instead of following them.

One process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo``
in the CPU tests).
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def slab_range(nx: int, world_size: int, rank: int) -> Tuple[int, int]:
    """x-planes [begin, end) owned by ``rank`` (same split as the one-shot C entry points)."""
    base, rem = divmod(nx, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def allgather_grid(full: torch.Tensor, local: torch.Tensor, group=None) -> torch.Tensor:
    """Assemble ``full[8, nx, ny, nz]`` on every rank from the ranks' contiguous slabs.

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
    if nx % world == 0:
        _gather_block(full, local, 0, nx, 0, nx // world, world, group)
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


_COALESCE_OK = True


def _gather_block(full: torch.Tensor, local: torch.Tensor, lo: int, hi: int, l0: int, m: int, world: int, group=None) -> None:
    """all-gather ``local[:, l0:l0+m]`` of every rank into ``full[:, lo:hi]`` (``hi-lo == world*m``,
    rank r's piece at ``lo + r*m``), one collective per channel, coalesced into one group launch
    where the backend allows it."""
    global _COALESCE_OK
    nchan = full.shape[0]
    assert hi - lo == world * m
    backend = dist.get_backend(group)
    if backend == "gloo":
        for c in range(nchan):
            outs = [full[c, lo + r * m: lo + (r + 1) * m] for r in range(world)]
            dist.all_gather(outs, local[c, l0:l0 + m].contiguous(), group=group)
        return

    def issue():
        for c in range(nchan):
            dist.all_gather_into_tensor(full[c, lo:hi], local[c, l0:l0 + m], group=group)

    if _COALESCE_OK:
        try:
            with dist._coalescing_manager(group=group, device=full.device, async_ops=False):
                issue()
            return
        except Exception:            # private API: fall back to plain per-channel calls
            _COALESCE_OK = False
    issue()


@dataclass
class CyclicPlan:
    """Block-cyclic distribution of ``nx`` x-planes: ``nchunks`` super-blocks of ``world*m`` planes."""
    nx: int
    world: int
    rank: int
    nchunks: int
    m: int

    @property
    def n_local(self) -> int:
        return self.nchunks * self.m

    def chunk(self, j: int) -> Tuple[int, int]:
        """global x-planes [begin, end) this rank computes in chunk j"""
        b = j * self.world * self.m + self.rank * self.m
        return b, b + self.m

    def block(self, j: int) -> Tuple[int, int]:
        """global x-planes [lo, hi) of super-block j (all ranks)"""
        return j * self.world * self.m, (j + 1) * self.world * self.m


def cyclic_plan(nx: int, world: int, rank: int, nchunks: int = 4, align: int = 4) -> Optional[CyclicPlan]:
    """Largest ``nchunks' <= nchunks`` such that ``nx == nchunks' * world * m`` with ``m`` a multiple
    of ``align`` (the kernel's tile edge); ``None`` if even one chunk does not fit (use slabs then)."""
    for k in range(nchunks, 0, -1):
        if nx % (k * world) == 0 and (nx // (k * world)) % align == 0:
            return CyclicPlan(nx, world, rank, k, nx // (k * world))
    return None


class PipelinedGather:
    """Compute chunk j+1 while chunk j is being all-gathered.

    ``launch(j, i_begin, i_end, local_plane_offset)`` must enqueue, on the current stream, the build
    of global planes [i_begin, i_end) into ``local[:, off:off+m]`` of every grid in ``locals_``.
    """

    def __init__(self, plan: CyclicPlan, fulls: List[torch.Tensor], locals_: List[torch.Tensor], group=None):
        self.plan, self.fulls, self.locals, self.group = plan, fulls, locals_, group
        dev = fulls[0].device
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None

    def run(self, launch: Callable[[int, int, int, int], None], on_compute_done: Optional[Callable[[], None]] = None) -> None:
        p = self.plan
        cur = torch.cuda.current_stream() if self.comm_stream is not None else None
        for j in range(p.nchunks):
            b, e = p.chunk(j)
            launch(j, b, e, j * p.m)
            if j == p.nchunks - 1 and on_compute_done is not None:
                on_compute_done()          # e.g. record a timing event after the last kernel
            if p.world == 1:
                continue
            lo, hi = p.block(j)
            if self.comm_stream is not None:
                ev = torch.cuda.Event()
                ev.record(cur)
                self.comm_stream.wait_event(ev)
                ctx = torch.cuda.stream(self.comm_stream)
            else:
                ctx = contextlib.nullcontext()
            with ctx:
                for full, local in zip(self.fulls, self.locals):
                    _gather_block(full, local, lo, hi, j * p.m, p.m, p.world, self.group)
        if self.comm_stream is not None and p.world > 1:
            cur.wait_stream(self.comm_stream)
