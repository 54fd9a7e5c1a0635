"""N > 1 path on CPU: world_size 2 and 3 over gloo.  Each rank fills its x-slab (the oracle
stands in for the kernel launch here -- this test is about the sharding + gather, the data
path of ceg_hip.distributed), then the single all-gather assembles the grid on every rank."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, spacing, outdir):
    for p in (str(ROOT / "crystalenergygrids.jl_amd"), str(ROOT)):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ceg_hip import grids as G, workloads as W
        from ceg_hip.distributed import allgather_grid, slab_range
        from oracle import oracle as O
        w = W.fixture_workload("CIT-7", "Ar", spacing)
        nx, ny, nz = w.cset.npoints
        b, e = slab_range(nx, world, rank)
        lam, thr = G.vdw_scaling()
        g, _ = O.grid_vdw(w.probe_vdw, w.cset, lam, thr, b, e, nthreads=2)
        local = torch.from_numpy(np.ascontiguousarray(g[:, b:e]))
        full = torch.full((8, nx, ny, nz), float("nan"), dtype=torch.float32)
        allgather_grid(full, local)
        np.save(os.path.join(outdir, f"rank{rank}.npy"), full.numpy())
        # block-cyclic chunks gathered in place (the pipelined path of bench.py), when nx divides
        from ceg_hip.distributed import PipelinedGather, cyclic_plan
        cyc = cyclic_plan(nx, world, rank, nchunks=4, align=1)
        if cyc is not None:
            for mode in ("staged", "inplace", "p2p"):
                loc = torch.full((cyc.nchunks, 8, cyc.m, ny, nz), float("nan"), dtype=torch.float32)
                full2 = torch.full((8, nx, ny, nz), float("nan"), dtype=torch.float32)

                def launch(j, ib, ie, blocks):
                    gg, _ = O.grid_vdw(w.probe_vdw, w.cset, lam, thr, ib, ie, nthreads=2)
                    blocks[0].copy_(torch.from_numpy(gg[:, ib:ie]))

                PipelinedGather(cyc, [full2], [loc], mode=mode).run(launch)
                np.save(os.path.join(outdir, f"rank{rank}_cyclic_{mode}.npy"), full2.numpy())
            # two grids whose blocks are slices of one joint tensor: ONE collective per chunk (bench.py's fused build)
            joint = torch.full((cyc.nchunks, 2, 8, cyc.m, ny, nz), float("nan"), dtype=torch.float32)
            fa = torch.full((8, nx, ny, nz), float("nan"), dtype=torch.float32)
            fb = torch.full((8, nx, ny, nz), float("nan"), dtype=torch.float32)

            def launch2(j, ib, ie, blocks):
                gg, _ = O.grid_vdw(w.probe_vdw, w.cset, lam, thr, ib, ie, nthreads=2)
                blocks[0].copy_(torch.from_numpy(gg[:, ib:ie]))
                blocks[1].copy_(torch.from_numpy(-2.0 * gg[:, ib:ie]))

            pg = PipelinedGather(cyc, [fa, fb], [joint[:, 0], joint[:, 1]], mode="staged", joint=joint)
            assert pg.joint is not None
            pg.run(launch2)
            np.save(os.path.join(outdir, f"rank{rank}_cyclic_joint_a.npy"), fa.numpy())
            np.save(os.path.join(outdir, f"rank{rank}_cyclic_joint_b.npy"), fb.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,spacing", [(2, 1.5), (3, 1.5), (2, 2.5)])
def test_slab_gather_gloo(tmp_path, world, spacing, oracle):
    from ceg_hip import grids as G, workloads as W
    port = _free_port()
    mp.spawn(_worker, args=(world, port, spacing, str(tmp_path)), nprocs=world, join=True)
    w = W.fixture_workload("CIT-7", "Ar", spacing)
    lam, thr = G.vdw_scaling()
    ref, _ = oracle.grid_vdw(w.probe_vdw, w.cset, lam, thr)
    assert not np.isnan(ref).any()
    ncyc = 0
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        np.testing.assert_array_equal(got, ref)
        for mode in ("staged", "inplace", "p2p"):
            f = tmp_path / f"rank{r}_cyclic_{mode}.npy"
            if f.exists():
                np.testing.assert_array_equal(np.load(f), ref)
                ncyc += 1
        fa = tmp_path / f"rank{r}_cyclic_joint_a.npy"
        if fa.exists():
            np.testing.assert_array_equal(np.load(fa), ref)
            np.testing.assert_array_equal(np.load(tmp_path / f"rank{r}_cyclic_joint_b.npy"), -2.0 * ref)
            ncyc += 1
    assert ncyc in (0, 4 * world)


def _layout8_worker(rank, world, port, outdir):
    for p in (str(ROOT / "crystalenergygrids.jl_amd"), str(ROOT)):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ceg_hip.distributed import PipelinedGather, cyclic_plan
        nx, ny, nz = 256, 3, 5                      # the x extent of the benchmark grid, tiny planes
        cyc = cyclic_plan(nx, world, rank, nchunks=8)
        assert cyc is not None and cyc.m == nx // (8 * world)
        joint = torch.full((cyc.nchunks, 2, 8, cyc.m, ny, nz), float("nan"), dtype=torch.float32)
        fa = torch.full((8, nx, ny, nz), float("nan"), dtype=torch.float32)
        fb = torch.full((8, nx, ny, nz), float("nan"), dtype=torch.float32)

        def value(c, i, j, k):                       # what a correct build leaves at channel c, plane i, row j, column k
            return (c * 1000.0 + i) + 0.01 * j + 0.0001 * k

        jj, kk = torch.meshgrid(torch.arange(ny, dtype=torch.float32), torch.arange(nz, dtype=torch.float32), indexing="ij")

        def launch(j, ib, ie, blocks):
            assert ie - ib == cyc.m
            for c in range(8):
                for t, i in enumerate(range(ib, ie)):
                    blocks[0][c, t] = value(c, i, jj, kk)
                    blocks[1][c, t] = -value(c, i, jj, kk)

        for mode in ("staged", "p2p"):
            fa.fill_(float("nan")); fb.fill_(float("nan"))
            pg = PipelinedGather(cyc, [fa, fb], [joint[:, 0], joint[:, 1]], mode=mode, joint=joint if mode == "staged" else None)
            pg.run(launch)
            want = torch.stack([torch.stack([value(c, i, jj, kk) for i in range(nx)]) for c in range(8)])
            assert torch.equal(fa, want) and torch.equal(fb, -want), (rank, mode)
        open(os.path.join(outdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_eight_rank_layout_of_the_benchmark(tmp_path):
    """The data movement of `bench.py --gpus 8` as the driver will launch it (256 x-planes, 8 chunks of 8 x 4 planes, both grids of a
    chunk in one collective), on 8 gloo ranks with tiny planes: every rank ends with every plane of both grids in its place."""
    world = 8
    mp.spawn(_layout8_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_cyclic_plan_shapes():
    from ceg_hip.distributed import cyclic_plan
    for world in (1, 2, 4, 8):
        p = cyclic_plan(256, world, world - 1, nchunks=4)
        assert p is not None and p.nchunks == 4 and p.m == 256 // (4 * world) and p.n_local * world == 256
        covered = sorted(x for r in range(world) for j in range(p.nchunks)
                         for x in range(*cyclic_plan(256, world, r, 4).chunk(j)))
        assert covered == list(range(256))
        assert p.block(p.nchunks - 1)[1] == 256
    assert cyclic_plan(218, 8, 0) is None            # falls back to contiguous slabs
    q = cyclic_plan(64, 8, 3, nchunks=4)             # only 2 chunks keep m a multiple of the tile edge
    assert q.nchunks == 2 and q.m == 4


def _autotune_worker(rank, world, port, outdir):
    for p in (str(ROOT / "crystalenergygrids.jl_amd"), str(ROOT)):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        import time
        from ceg_hip.distributed import PipelinedGather, autotune_exchange, cyclic_plan
        nx, ny, nz = 32, 3, 5
        state = {}

        def setup(label):
            nch, mode = int(label.split()[0]), label.split(", ")[1]
            if label == "4 chunks, inplace" and rank == 1:
                raise RuntimeError("injected: this candidate fails on rank 1 only")
            if label == "2 chunks, staged":
                raise RuntimeError("injected: this candidate fails everywhere")
            cyc = cyclic_plan(nx, world, rank, nchunks=nch, align=1)
            loc = torch.zeros((cyc.nchunks, 8, cyc.m, ny, nz))
            full = torch.full((8, nx, ny, nz), float("nan"))
            state.update(cyc=cyc, loc=loc, full=full, pipe=PipelinedGather(cyc, [full], [loc], mode=mode), label=label)

        def step():
            def launch(j, ib, ie, blocks):
                if state["label"] == "2 chunks, inplace":
                    raise RuntimeError("injected: the first step fails (on every rank, before its first collective)")
                if state["label"].startswith("8"):
                    time.sleep(0.01)                      # the 8-chunk candidates are the slow ones
                blocks[0].copy_(torch.arange(ib, ie, dtype=torch.float32)[None, :, None, None].expand(8, ie - ib, ny, nz))
            state["pipe"].run(launch)

        labels = [f"{n} chunks, {m}" for n in (8, 4, 2) for m in ("staged", "inplace")]
        best, res = autotune_exchange(labels, setup, step, lambda: None, torch.device("cpu"), log=lambda m: None)
        # whatever was chosen must produce the right grid on this rank
        setup(best)
        step()
        ok = bool(torch.equal(state["full"], torch.arange(nx, dtype=torch.float32)[None, :, None, None].expand(8, nx, ny, nz)))
        with open(os.path.join(outdir, f"autotune{rank}.json"), "w") as fh:
            json.dump({"best": best, "results": res, "grid_ok": ok}, fh)
    finally:
        dist.destroy_process_group()


def test_autotune_survives_failing_candidates(tmp_path):
    """bench.py --gather auto: a candidate that raises -- during set-up on one rank only or on all ranks, or inside its first
    step -- is dropped on EVERY rank, the ranks end with the same choice and that choice works.  (A rank
    that dies INSIDE a step while its peers already wait in a collective cannot be rescued in-process: the default does not
    autotune.)"""
    import json
    world = 2
    mp.spawn(_autotune_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out = [json.loads((tmp_path / f"autotune{r}.json").read_text()) for r in range(world)]
    assert out[0]["best"] == out[1]["best"] == "4 chunks, staged"
    for o in out:
        assert o["grid_ok"]
        r = o["results"]
        assert r["4 chunks, inplace"] is None and r["2 chunks, staged"] is None and r["2 chunks, inplace"] is None
        assert r["8 chunks, staged"] > r["4 chunks, staged"] > 0 and r["8 chunks, inplace"] is not None
    assert out[0]["results"] == out[1]["results"]            # max over ranks: identical numbers everywhere
