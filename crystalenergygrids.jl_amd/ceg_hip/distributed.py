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
  ranks' pieces are adjacent in the final array, so each chunk is gathered and put in place
  *while the next chunk is being computed* (collectives on a side stream, ``PipelinedGather``):
  the exchange over xGMI hides behind the FP64 kernels instead of following them.

One process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo``
in the CPU tests).
"""
from __future__ import annotations

import contextlib
import statistics
import sys
import time
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

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


def _gather_block(full: torch.Tensor, local: torch.Tensor, lo: int, hi: int, l0: int, m: int, world: int, group=None) -> None:
    """all-gather ``local[:, l0:l0+m]`` of every rank into ``full[:, lo:hi]`` (``hi-lo == world*m``,
    rank r's piece at ``lo + r*m``), one collective per channel."""
    nchan = full.shape[0]
    assert hi - lo == world * m
    backend = dist.get_backend(group)
    for c in range(nchan):
        if backend == "gloo":
            outs = [full[c, lo + r * m: lo + (r + 1) * m] for r in range(world)]
            dist.all_gather(outs, local[c, l0:l0 + m].contiguous(), group=group)
        else:
            dist.all_gather_into_tensor(full[c, lo:hi], local[c, l0:l0 + m], group=group)


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
    """Compute chunk j+1 while chunk j is being all-gathered (collectives on a side stream).

    Rank-local results live in ``locals_[g]`` of shape ``[nchunks, 8, m, ny, nz]`` (one compact
    ``[8, m, ny, nz]`` block per chunk).  ``launch(j, i_begin, i_end, blocks)`` must enqueue, on the
    current stream, the build of global planes [i_begin, i_end) into ``blocks[g] = locals_[g][j]``
    (channel stride ``m*ny*nz``, ``i_origin = i_begin``).

    Two ways to place a gathered chunk, both with public torch.distributed calls only:

    * ``"staged"`` (default): ONE ``all_gather_into_tensor`` per grid and chunk into a
      ``[world, 8, m, ny, nz]`` staging buffer, then one strided device copy into the final
      ``[8, nx, ny, nz]`` array (channel and rank axes swapped); when the grids' local blocks are slices of one
      ``joint`` tensor ``[nchunks, G, 8, m, ny, nz]`` the G grids of a chunk travel in ONE collective
      (half the collective launches of a fused build: at N = 8 a chunk's compute is ~0.2 ms, the same order as the
      launch latency of a collective);
    * ``"inplace"``: 8 ``all_gather_into_tensor`` calls per grid and chunk (one per channel)
      straight into the final array -- no copy, more (smaller) collectives;
    * ``"p2p"``: the same staging buffer filled by one grouped ``batch_isend_irecv`` per chunk -- every
      rank sends its block to each peer and receives each peer's block directly.  xGMI is a full mesh
      of point-to-point links, so the 7 transfers of a rank travel on 7 different links at once and
      no block is forwarded; an alternative to RCCL's ring all-gather to be compared on an 8-GPU node.
    """

    def __init__(self, plan: CyclicPlan, fulls: List[torch.Tensor], locals_: List[torch.Tensor], group=None,
                 mode: str = "staged", force_collectives: bool = False, joint: Optional[torch.Tensor] = None):
        assert mode in ("staged", "inplace", "p2p")
        self.plan, self.fulls, self.locals, self.group, self.mode = plan, fulls, locals_, group, mode
        self.joint = None
        if joint is not None and mode == "staged" and len(locals_) > 1:
            assert joint.is_contiguous() and tuple(joint.shape[:2]) == (plan.nchunks, len(locals_))
            for g, loc in enumerate(locals_):
                assert loc.data_ptr() == joint[0, g].data_ptr() and loc.stride(0) == joint.stride(0)
            self.joint = joint
        self.exchange = plan.world > 1 or force_collectives      # world 1: collectives only on request (tests)
        dev = fulls[0].device
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        for full, loc in zip(fulls, locals_):
            assert tuple(loc.shape) == (plan.nchunks, full.shape[0], plan.m) + tuple(full.shape[2:]) and loc[0].is_contiguous()
        self.staging = None
        if self.joint is not None and self.exchange:
            shape = (plan.world,) + tuple(self.joint.shape[1:])
            self.joint_staging = [torch.empty(shape, dtype=fulls[0].dtype, device=dev) for _ in range(2)]
        elif mode in ("staged", "p2p") and self.exchange:
            shape = (plan.world,) + tuple(locals_[0].shape[1:])
            self.staging = [[torch.empty(shape, dtype=f.dtype, device=dev) for f in fulls] for _ in range(2)]

    def _gather_chunk(self, j: int) -> None:
        p = self.plan
        lo, hi = p.block(j)
        backend = dist.get_backend(self.group)
        if self.mode == "p2p":
            ops = []
            for g, loc in enumerate(self.locals):
                block, stage = loc[j], self.staging[j % 2][g]
                stage[p.rank].copy_(block)
                for r in range(p.world):
                    if r != p.rank:
                        ops.append(dist.P2POp(dist.isend, block, r, group=self.group))
                        ops.append(dist.P2POp(dist.irecv, stage[r], r, group=self.group))
            for req in (dist.batch_isend_irecv(ops) if ops else []):
                req.wait()
            for g, full in enumerate(self.fulls):
                full[:, lo:hi].unflatten(1, (p.world, p.m)).copy_(self.staging[j % 2][g].permute(1, 0, 2, 3, 4))
            return
        if self.joint is not None:
            stage = self.joint_staging[j % 2]                         # [world, G, 8, m, ny, nz]
            if backend == "gloo":
                dist.all_gather(list(stage.unbind(0)), self.joint[j], group=self.group)
            else:
                dist.all_gather_into_tensor(stage, self.joint[j], group=self.group)
            for g, full in enumerate(self.fulls):                     # final_g[c, lo + r*m + t] = stage[r, g, c, t]
                full[:, lo:hi].unflatten(1, (p.world, p.m)).copy_(stage[:, g].permute(1, 0, 2, 3, 4))
            return
        for g, (full, loc) in enumerate(zip(self.fulls, self.locals)):
            block = loc[j]                                            # [8, m, ny, nz], contiguous
            if self.mode == "inplace":
                for c in range(full.shape[0]):
                    if backend == "gloo":
                        outs = [full[c, lo + r * p.m: lo + (r + 1) * p.m] for r in range(p.world)]
                        dist.all_gather(outs, block[c], group=self.group)
                    else:
                        dist.all_gather_into_tensor(full[c, lo:hi], block[c], group=self.group)
                continue
            stage = self.staging[j % 2][g]                            # [world, 8, m, ny, nz]
            if backend == "gloo":
                dist.all_gather(list(stage.unbind(0)), block, group=self.group)
            else:
                dist.all_gather_into_tensor(stage, block, group=self.group)
            # final[c, lo + r*m + t] = stage[r, c, t]
            full[:, lo:hi].unflatten(1, (p.world, p.m)).copy_(stage.permute(1, 0, 2, 3, 4))

    def run(self, launch: Callable[[int, int, int, List[torch.Tensor]], None],
            on_compute_done: Optional[Callable[[], None]] = None) -> None:
        p = self.plan
        cur = torch.cuda.current_stream() if self.comm_stream is not None else None
        for j in range(p.nchunks):
            b, e = p.chunk(j)
            launch(j, b, e, [loc[j] for loc in self.locals])
            if j == p.nchunks - 1 and on_compute_done is not None:
                on_compute_done()          # e.g. record a timing event after the last kernel
            if not self.exchange:
                continue
            if self.comm_stream is not None:
                ev = torch.cuda.Event()
                ev.record(cur)
                self.comm_stream.wait_event(ev)
                ctx = torch.cuda.stream(self.comm_stream)
            else:
                ctx = contextlib.nullcontext()
            with ctx:
                self._gather_chunk(j)
        if self.comm_stream is not None and self.exchange:
            cur.wait_stream(self.comm_stream)


def autotune_exchange(labels: Sequence[str], setup: Callable[[str], None], step: Callable[[], None],
                      sync: Callable[[], None], device: torch.device, timed_steps: int = 3, group=None,
                      log: Callable[[str], None] = lambda m: print(m, file=sys.stderr)) -> Tuple[Optional[str], Dict[str, Optional[float]]]:
    """Time every exchange configuration in ``labels`` on this node and return ``(best, {label: ms or None})``.

    ``setup(label)`` builds the buffers / pipeline of a candidate, ``step()`` runs one full step with it, ``sync()`` drains the
    device.  A candidate goes through three stages -- set-up, one untimed step, ``timed_steps`` timed steps (median, MAX over
    the ranks) -- and after each stage the ranks agree (all-reduce MIN of an ok flag) whether to go on: a candidate that raises
    on ANY rank is dropped on ALL of them (``None`` in the result), so no rank is left inside a collective the failed rank will
    never join and every rank ends with the same choice.  ``best`` is ``None`` when nothing completed: the caller keeps its
    default.  (A candidate that hangs instead of raising cannot be rescued from inside the process; the benchmark's default does
    not autotune for that reason.)"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0

    def agree(ok: bool) -> bool:
        if world == 1:
            return ok
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(t[0] > 0.5)

    def timed() -> float:
        ts = []
        for _ in range(timed_steps):
            sync()
            if world > 1:
                dist.barrier(group=group)
            t0 = time.perf_counter()
            step()
            sync()
            ts.append(time.perf_counter() - t0)
        t = torch.tensor([statistics.median(ts)], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return float(t[0]) * 1e3

    results: Dict[str, Optional[float]] = {}
    for label in labels:
        ok, ms = True, None
        for stage in ("setup", "first step", "timed steps"):
            try:
                if stage == "setup":
                    setup(label)
                elif stage == "first step":
                    step()
                    sync()
                else:
                    ms = timed()
            except Exception as exc:          # noqa: BLE001 -- any failure disqualifies the candidate, nothing else
                ok = False
                log(f"[autotune] candidate '{label}' failed at '{stage}' on rank {rank}: {exc!r}")
            ok = agree(ok)
            if not ok:
                break
        results[label] = ms if ok else None
    done = {k: v for k, v in results.items() if v is not None}
    return (min(done, key=done.get) if done else None), results
