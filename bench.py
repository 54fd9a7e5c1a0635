#!/usr/bin/env python3
"""bench.py -- grid-points/s of the grid-build hot path on MI355X.

One "step" = one full build of the BASELINE.json roofline workload (SURVEY §8d run "R"):
the CHA_1.4_3b4eeb96 fixture tiled 2x2x3 (11 664 framework atoms) probed on a 256^3 grid,
Lennard-Jones (Ar probe) VdW grid + real-space-Ewald Coulomb grid, each 8 Float32 channels per
point, computed in FP64 by the hand-written HIP kernels behind the C ABI
(crystalenergygrids.jl_amd/csrc).  Inputs (atom table, lattice images, bins) are resident in
HBM before the timed region; the timed region is kernel launch(es) + (N > 1) the RCCL
all-gather that assembles the grid on every rank.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `value` = grid points of the full grid / step time (whole job).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT / "crystalenergygrids.jl_amd"), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz (no MFMA on this path)
# Nominal flop convention of SURVEY §8d (add / mul / fma-half / div / sqrt = 1, exp = 20, erfc = 40): per in-cutoff
# pair incl. the 8-way accumulation LJ 50, Buckingham 95, real-space Ewald 140.  The pair COUNTS are counted on the
# workload (ceg_hip.workloads.count_pair_work), not assumed.  Distance: an algorithm that works from an explicit
# image list needs 3 subtractions + |d|^2 = 8 flops per pair; SURVEY's 47 is the reference's brute-force routine
# (two 3x3 mat-vecs + wrap) and is only reported alongside for continuity with round 1.
PREWARM_STEPS = 3          # untimed launches of the set-up phase (clock ramp-up), in addition to --warmup
F_DIST, F_DIST_SURVEY, F_LJ, F_BUCK, F_EWALD = 8.0, 47.0, 50.0, 95.0, 140.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=("fused", "vdw", "coulomb"), default="fused")
    ap.add_argument("--probe", default="Ar", help="probe atom of the VdW grid (Ar: LJ; Na: Buckingham + hard sphere)")
    ap.add_argument("--algo", choices=("auto", "bruteforce", "culled"), default="auto")
    ap.add_argument("--gather", choices=("auto", "staged", "inplace", "p2p"), default="staged",
                    help="how a gathered chunk is placed (N > 1).  staged (default): one all_gather_into_tensor per chunk + one placement "
                         "copy, --chunks as given -- the path every rehearsal has exercised.  auto: opt-in autotune over --chunks, half "
                         "and a quarter of it x {staged, inplace}; a candidate that raises is skipped, the default always takes part "
                         "(CEG_BENCH_NO_AUTOTUNE=1 turns auto back into staged)")
    ap.add_argument("--chunks", type=int, default=8,
                    help="x-chunks per rank pipelined with the all-gather (N > 1); 8 from the exchange model of profiles/r02_rank_emulation.txt")
    ap.add_argument("--n", "--dims", dest="n", type=int, default=255, help="dims per axis (odd); grid has (n+1)^3 points")
    ap.add_argument("--cpu-rows", type=int, default=-1, help="y-rows per x-plane the CPU baseline times, one plane per thread (-1 = auto ~12 s, 0 = skip)")
    ap.add_argument("--no-check", action="store_true", help="skip the built-in oracle spot check")
    ap.add_argument("--no-oneshot", action="store_true", help="skip the one-shot API timings (N = 1: ceg_grid_vdw / ceg_grid_coulomb / ceg_grids_multi into host arrays)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend; gloo is a rehearsal aid: several ranks may then share one GPU "
                         "(device = LOCAL_RANK mod device count), which RCCL refuses")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N = 1 only: run the N > 1 code path (RCCL group of one rank, chunked build, collectives issued) -- a rehearsal")
    return ap.parse_args()


# the sources that determine the instruction stream of the grid-build kernels (k_culled / k_bruteforce) and the tables they read:
# a PMC record stays valid across edits of the consumer kernels (ceg_mc / ceg_pairs / ceg_recip / ceg_interp / ceg_block)
GRID_KERNEL_SOURCES = ("Makefile", "ceg_api.hip", "ceg_internal.h", "ceg_kernels.hip", "ceg_math.h", "ceg_minimage.h")


def csrc_sha256() -> str:
    """Identity of the grid-kernel sources the library was built from (same recipe as scripts/pmc.sh)."""
    import hashlib
    h = hashlib.sha256()
    csrc = ROOT / "crystalenergygrids.jl_amd" / "csrc"
    for fn in GRID_KERNEL_SOURCES:
        h.update(fn.encode())
        data = (csrc / fn).read_bytes()
        if fn == "Makefile":
            # only what reaches the compiler (ARCH / HIPCC / CXXFLAGS with its continuation lines): the lists of sources and headers
            # change whenever a consumer kernel gains a file
            keep, cont = [], False
            for line in data.decode().splitlines():
                if cont or line.split("=")[0].strip() in ("ARCH", "HIPCC", "CXXFLAGS"):
                    keep.append(line)
                    cont = line.rstrip().endswith("\\")
                else:
                    cont = False
            data = "\n".join(keep).encode()
        h.update(data)
    return h.hexdigest()


def load_pmc_record(key: str):
    """PMC-derived figures of `key` = mode/probe/n/world from profiles/pmc_summary.json, with their provenance; None if the file has
    no record for this configuration.  A record taken on OTHER kernel sources is returned with only its provenance and
    "stale": true -- none of its numbers is used."""
    prof = ROOT / "profiles" / "pmc_summary.json"
    try:
        rec = json.loads(prof.read_text()).get(key)
    except Exception:
        return None
    if not rec:
        return None
    out = {"source": "profiles/pmc_summary.json", "key": key, "collected_by": rec.get("command"), "host": rec.get("host"),
           "csrc_sha256": rec.get("csrc_sha256"), "raw_summary": rec.get("source"),
           "note": "separate rocprofv3 --pmc passes of the same bench command on another run (and possibly another box); timed launches only"}
    out["stale"] = rec.get("csrc_sha256") != csrc_sha256()
    if out["stale"]:
        #  a stale record no longer blanks the headline: its instruction counts are still reported -- flagged -- because an
        # edit that does not touch the hot loops leaves them unchanged; tests/test_host_logic.py fails while the committed record is stale,
        # so a round cannot end on one
        out["note"] = ("profiles/pmc_summary.json was collected on DIFFERENT grid-kernel sources (csrc_sha256 differs): the figures derived "
                       "from it are kept but flagged stale -- re-run scripts/pmc.sh + scripts/pmc_merge.py")
    for k in ("hbm_bytes_per_launch", "valu_issue_util", "lane_util", "frac_issue", "fp64_flops_per_launch", "fp64_insts_per_launch",
              "fma_share_of_fp64_insts", "gpu_cycles_per_launch", "kernel_ms_in_profile", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
              "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS"):
        if k in rec:
            out[k] = rec[k]
    return out


def cpu_baseline(w, mode: str, rows: int):
    """Oracle (C restatement of the reference algorithm: brute force over all atoms, literal
    min-image routine, threaded over the x index like Threads.@threads in grids.jl:144) on a
    bounded sample of the same workload: one x-plane per thread, `rows` y-rows of each.  The output
    array is allocated once, outside the timed region (the reference allocates its grid before the
    loop nest too, grids.jl:139); threads = CPUs this process may use (affinity mask capped by the cgroup
    quota), and the per-thread pair-check rate is reported next to a single-thread measurement so that an
    oversubscribed or throttled host shows."""
    from oracle import oracle as O
    from ceg_hip import grids as G
    nx, ny, nz = w.cset.npoints
    out = np.empty((8, nx, ny, nz), dtype=np.float32)          # untouched pages: only the sampled rows get written
    lam_v, thr_v = G.vdw_scaling()
    lam_c, thr_c = G.coulomb_scaling()
    ngrids = int(mode in ("fused", "vdw")) + int(mode in ("fused", "coulomb"))
    mid = ny // 2

    def run(threads, j0, j1):
        planes = min(threads, nx)
        i0 = max(0, nx // 2 - planes // 2)
        t = time.perf_counter()
        if mode in ("fused", "vdw"):
            O.grid_vdw(w.probe_vdw, w.cset, lam_v, thr_v, i0, i0 + planes, j_begin=j0, j_end=j1, nthreads=threads, out=out)
        if mode in ("fused", "coulomb"):
            O.grid_coulomb(w.probe_coulomb, w.alpha, w.cset, lam_c, thr_c, i0, i0 + planes, j_begin=j0, j_end=j1, nthreads=threads, out=out)
        t = time.perf_counter() - t
        return planes * (j1 - j0) * nz, t

    threads = int(os.environ.get("CEG_BENCH_THREADS", "0")) or O.usable_cpus()
    run(1, mid, mid + 1)                                         # page in the library and the touched rows
    p1, t1 = run(1, mid, mid + 1)
    single = p1 * w.natoms * ngrids / t1                         # pair checks/s of one thread alone
    pT, tT = run(threads, mid, mid + 1)
    eff = (pT * w.natoms * ngrids / tT) / (threads * single)
    if eff < 0.5 and threads > 16 and "CEG_BENCH_THREADS" not in os.environ:
        # more runnable threads than cores behind them (a CPU share the affinity mask does not show): fall back to the
        # GPU box's documented share
        threads = 16
        pT, tT = run(threads, mid, mid + 1)
        eff = (pT * w.natoms * ngrids / tT) / (threads * single)
    if rows < 0:
        rows = int(max(1, min(ny, round(float(os.environ.get("CEG_BENCH_CPU_SECONDS", "12")) / max(tT, 1e-9)))))
    j0 = max(0, mid - rows // 2)
    j1 = min(ny, j0 + rows)
    pts, t = run(threads, j0, j1)
    rate = pts * w.natoms * ngrids / t
    return {"value": pts / t, "unit": "grid-points/s", "cores": threads, "kind": "port",
            "pair_checks_per_s": rate, "pair_checks_per_s_per_thread": rate / threads,
            "pair_checks_per_s_single_thread": single, "parallel_efficiency": rate / (threads * single),
            "sample": f"{min(threads, nx)} x-planes x {j1 - j0} y-rows x {nz} ({pts} of {nx * ny * nz} points) x {w.natoms} atoms, "
                      f"{mode}, {t:.1f} s, output array preallocated; CPU restatement of the reference algorithm "
                      "(brute force over all atoms), not the Julia package"}


def oneshot_record(w, reps: int = 3):
    """The API path the Julia shim binds (julia/CEGHip.jl create_grid_vdw / create_grid_coulomb -> ceg_grid_vdw / ceg_grid_coulomb, and
    prebuild_grids! -> ceg_grids_multi): host arrays in, the finished 8-channel grid in a HOST array out -- plan creation, kernel(s),
    the 537 MB D2H per grid and the placement in the caller's array all inside the call, so PCIe-bound and never `value`
    (reference: create_grid_vdw returns a host Array, src/grids.jl:137-157).  Wall time per call after one untimed call (the first
    call of a process page-locks the staging ring: ~0.9 s), into a freshly allocated pageable array and into a page-locked array of
    the library (ceg_host_grid_alloc: every chunk is copied straight to its place)."""
    from ceg_hip import grids as G
    nx, ny, nz = w.cset.npoints
    npts = nx * ny * nz

    def timed(fn):
        fn()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            out = fn()
            ts.append((time.perf_counter() - t) * 1e3)
            del out
        return {"ms": [round(x, 3) for x in ts], "best_ms": round(min(ts), 3), "points_per_s": npts / (min(ts) * 1e-3)}

    rec = {"unit": "ms wall per call, host arrays in and out", "grid_points": npts, "result_bytes_per_grid": 32 * npts, "reps": reps}
    pin_v = G.alloc_host_grid(w.cset)
    pin_c = G.alloc_host_grid(w.cset)
    try:
        rec["ceg_grid_vdw"] = {"pageable": timed(lambda: G.build_vdw_array(w.probe_vdw, w.cset)),
                               "page_locked": timed(lambda: G.build_vdw_array(w.probe_vdw, w.cset, out=pin_v))}
        rec["ceg_grid_coulomb"] = {"pageable": timed(lambda: G.build_coulomb_array(w.probe_coulomb, w.alpha, w.cset)),
                                   "page_locked": timed(lambda: G.build_coulomb_array(w.probe_coulomb, w.alpha, w.cset, out=pin_c))}
    finally:
        del pin_v, pin_c
    rec["ceg_grids_multi"] = {"grids": 2,
                              "pageable": timed(lambda: G.build_multi_arrays([w.probe_vdw], w.probe_coulomb, w.alpha, w.cset)),
                              "page_locked": timed(lambda: G.build_multi_arrays([w.probe_vdw], w.probe_coulomb, w.alpha, w.cset, pinned=True))}
    rec["note"] = ("what the reference-side binding calls; D2H of 537 MB per grid at ~56 GB/s = 9.6 ms bounds every figure; "
                   "compare profiles/r03_oneshot_timing.txt (VdW 13.4 / 10.5 ms, Coulomb 15.4 / 12.7 ms pageable / page-locked)")
    return rec


def main():
    args = parse_args()
    # stdout must carry exactly ONE JSON line: native libraries (RCCL prints a version banner when its first
    # communicator is created) write to fd 1 behind Python's back, so fd 1 is pointed at stderr for the whole
    # run and the result line is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The hosts of this pool support dmabuf IPC only: with the legacy IPC mode RCCL's intra-node transport (and any sharing of device
        # memory between processes) fails with `hipIpcGetMemHandle: invalid argument`.  The image exports HSA_ENABLE_IPC_MODE_LEGACY=0
        # already; setdefault keeps whatever the launcher set and only fills it in for a hand-made environment.  It must be in place
        # before the HIP runtime initialises (nothing above this line touches the GPU).
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the grid build has no CPU path)")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_exchange          # take the sharded code path
    if multi:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:           # a free port: the rehearsal must not collide with a real job
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    from ceg_hip import _abi, workloads as W
    from ceg_hip.distributed import PipelinedGather, allgather_grid, autotune_exchange, cyclic_plan, slab_range
    from ceg_hip.plan import GridPlan

    w = W.roofline_workload(args.probe, args.n)
    nx, ny, nz = w.cset.npoints
    npts = nx * ny * nz
    algo = {"auto": _abi.ALGO_AUTO, "bruteforce": _abi.ALGO_BRUTEFORCE, "culled": _abi.ALGO_CULLED}[args.algo]
    plan = GridPlan(w.cset, w.probe_vdw, w.probe_coulomb, w.alpha, device=local_rank)
    need_v = args.mode in ("fused", "vdw")
    need_c = args.mode in ("fused", "coulomb")
    plane = ny * nz
    # N > 1: block-cyclic x-chunks so that each chunk is all-gathered in place (RCCL, side stream)
    # while the next one is computed; contiguous slabs + one gather at the end if nx does not divide.
    cyc = cyclic_plan(nx, world, rank, nchunks=args.chunks) if multi else None
    if cyc is not None:
        n_local = cyc.n_local
    else:
        b, e = slab_range(nx, world, rank)
        n_local = e - b
    # full grids live on every rank (that is what the gather produces); rank-local buffers
    full_v = torch.empty((8, nx, ny, nz), dtype=torch.float32, device=dev) if need_v else None
    full_c = torch.empty((8, nx, ny, nz), dtype=torch.float32, device=dev) if need_c else None
    joint = None

    def cyclic_buffers(c):     # one compact [8, m, ny, nz] block per chunk and grid, the grids of a chunk adjacent (one collective)
        jt = torch.empty((c.nchunks, int(need_v) + int(need_c), 8, c.m, ny, nz), dtype=torch.float32, device=dev)
        return jt, (jt[:, 0] if need_v else None), (jt[:, 1 if need_v else 0] if need_c else None)

    if not multi:
        loc_v, loc_c = full_v, full_c
    elif cyc is not None:
        joint, loc_v, loc_c = cyclic_buffers(cyc)
    else:
        loc_v = torch.empty((8, n_local, ny, nz), dtype=torch.float32, device=dev) if need_v else None
        loc_c = torch.empty((8, n_local, ny, nz), dtype=torch.float32, device=dev) if need_c else None

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def launch(i_begin, i_end, i_origin, ptr_v, ptr_c, cs):
        s = torch.cuda.current_stream().cuda_stream
        if args.mode == "fused":
            plan.build_fused(ptr_v, ptr_c, cs, i_begin, i_end, i_origin, algo, s)
        elif args.mode == "vdw":
            plan.build_vdw(ptr_v, cs, i_begin, i_end, i_origin, algo, s)
        else:
            plan.build_coulomb(ptr_c, cs, i_begin, i_end, i_origin, algo, s)

    fulls = [t for t in (full_v, full_c) if t is not None]
    locs = [t for t in (loc_v, loc_c) if t is not None]
    gather_mode, autotune = args.gather, None

    def make_pipe(mode):
        return PipelinedGather(cyc, fulls, locs, mode=mode, force_collectives=args.force_exchange,
                               joint=joint if cyc is not None and multi else None)

    pipe = make_pipe("staged" if args.gather == "auto" else args.gather) if cyc is not None else None

    def launch_chunk(j, ib, ie, blocks):
        it = iter(blocks)
        pv = next(it).data_ptr() if need_v else 0
        pc = next(it).data_ptr() if need_c else 0
        launch(ib, ie, ib, pv, pc, cyc.m * plane)

    def step(k=None):
        if k is not None:
            ev0[k].record()
        if pipe is not None:
            pipe.run(launch_chunk, on_compute_done=(lambda: ev1[k].record()) if k is not None else None)
        else:
            launch(b, e, b if multi else 0, loc_v.data_ptr() if need_v else 0, loc_c.data_ptr() if need_c else 0,
                   n_local * plane)
            if k is not None:
                ev1[k].record()
            if multi:
                for full, loc in zip(fulls, locs):
                    allgather_grid(full, loc)

    want_autotune = args.gather == "auto" and os.environ.get("CEG_BENCH_NO_AUTOTUNE", "0") in ("", "0")
    if pipe is not None and want_autotune and pipe.exchange:
        # Opt-in: pick, on THIS node, the number of chunks per rank and the placement of the gathered chunks.  Fewer chunks =
        # fewer, larger collectives and launches but a longer exposed first / last chunk; which side wins depends on the
        # collective's launch latency, which only a real run on the links shows.  Every candidate runs after the clock ramp
        # (PREWARM_STEPS), is timed over 3 steps (median, max over ranks) after one step of its own, and is dropped -- on all
        # ranks -- if it raises anywhere (ceg_hip.distributed.autotune_exchange); the default (staged, --chunks) is the first
        # candidate and the fallback.
        for _ in range(PREWARM_STEPS):
            step()
        default_cfg = (cyc.nchunks, "staged")
        cands = {}
        for nch in sorted({args.chunks, max(1, args.chunks // 2), max(1, args.chunks // 4)}, reverse=True):
            cand = cyclic_plan(nx, world, rank, nchunks=nch)
            if cand is not None:
                for mode in ("staged", "inplace"):
                    cands.setdefault(f"{cand.nchunks} chunks, {mode}", (cand, mode))

        def use(cfg_plan, mode):
            nonlocal cyc, joint, loc_v, loc_c, locs, pipe, n_local
            cyc = cfg_plan
            joint, loc_v, loc_c = cyclic_buffers(cyc)
            locs = [t for t in (loc_v, loc_c) if t is not None]
            pipe = make_pipe(mode)
            n_local = cyc.n_local

        def setup_candidate(label):
            if os.environ.get("CEG_BENCH_FAIL_CANDIDATE") == label:        # test hook: a candidate that raises
                raise RuntimeError("injected failure (CEG_BENCH_FAIL_CANDIDATE)")
            use(*cands[label])

        best, autotune = autotune_exchange(list(cands), setup_candidate, step, torch.cuda.synchronize, dev)
        if best is None:
            print("[bench] autotune: no candidate completed; using the default", file=sys.stderr)
            use(cyclic_plan(nx, world, rank, nchunks=default_cfg[0]), default_cfg[1])
            gather_mode = default_cfg[1]
        else:
            use(*cands[best])
            gather_mode = cands[best][1]
    elif args.gather == "auto":
        gather_mode = "staged"
    # Last line of defence for the first run on real links: ONE guarded step of the pipelined exchange before anything is timed.  The
    # ranks agree on its outcome through the rendezvous STORE, not through a collective: a rank that raises inside step()
    # while its peers already sit in an all_gather must not enter another collective -- RCCL would pair its all_reduce with their
    # all_gather.  Protocol: every rank publishes "ok" or "fail: reason" under its own key; a rank that has launched its step waits for
    # the step's completion EVENT by polling, and looks at the peers' keys while it does.
    #   * every rank "ok"                      -> the pipelined exchange is used;
    #   * every rank failed BEFORE its first collective (the same cause everywhere: a bad argument, an unsupported call)
    #                                           -> every rank drops to the simplest path there is -- contiguous x-slabs, one all-gather per
    #                                              channel after the kernels (allgather_grid) -- and the line says so in exchange.mode;
    #   * anything else (one rank fails while others are inside a collective, a step that does not complete in time)
    #                                           -> cannot be repaired in-process: every rank prints the reason and exits non-zero so that the
    #                                              launcher starts fresh processes (never re-exec a process that has touched the GPU).
    fallback_reason = None
    if pipe is not None and pipe.exchange and world > 1:
        store = dist.distributed_c10d._get_default_store()
        key = lambda r: f"ceg_bench_first_step/{r}"          # noqa: E731
        limit = float(os.environ.get("CEG_BENCH_FIRST_STEP_TIMEOUT", "180"))
        mine = "ok"
        launched = False
        try:
            inject = os.environ.get("CEG_BENCH_FAIL_PIPELINE", "")
            if inject == "1" or inject == f"rank{rank}":            # test hooks: every rank / one rank fails before its first collective
                raise RuntimeError("injected failure of the pipelined exchange (CEG_BENCH_FAIL_PIPELINE)")
            launched = True
            step()
            if inject == f"late{rank}":                             # test hook: this rank fails AFTER its collectives were enqueued
                raise RuntimeError("injected late failure of the pipelined exchange (CEG_BENCH_FAIL_PIPELINE)")
        except Exception as exc:              # noqa: BLE001
            mine = f"fail{'-late' if launched else ''}: {exc!r}"
            print(f"[bench] pipelined exchange failed on rank {rank}: {exc!r}", file=sys.stderr)
        store.set(key(rank), mine)
        done_ev = torch.cuda.Event()
        done_ev.record()
        t_wait = time.perf_counter()
        states = {}
        while True:
            for r in range(world):
                if r not in states:
                    try:
                        if store.check([key(r)]):
                            states[r] = store.get(key(r)).decode()
                    except Exception:         # noqa: BLE001 -- a store hiccup is not a verdict
                        pass
            finished = done_ev.query()
            bad = {r: v for r, v in states.items() if v != "ok"}
            if len(states) == world and (finished or bad):
                break
            if bad and not finished and any(v.startswith("fail-late") for v in bad.values()):
                break                                               # a peer died inside the exchange: our collective will not complete
            if time.perf_counter() - t_wait > limit:
                states.setdefault(rank, mine)
                bad = bad or {rank: f"fail-late: first step not finished after {limit:.0f} s"}
                break
            time.sleep(0.002)
        bad = {r: v for r, v in states.items() if v != "ok"}
        if time.perf_counter() - t_wait > limit and not bad:
            bad = {rank: f"fail-late: first step not finished after {limit:.0f} s"}
        if bad:
            early_everywhere = len(bad) == world and all(v.startswith("fail:") for v in bad.values())
            if not early_everywhere:
                print(f"[bench] rank {rank}: the first pipelined step cannot be agreed on ({bad}); exiting so that the launcher can start fresh processes",
                      file=sys.stderr)
                sys.stderr.flush()
                os._exit(3)                                         # (no destroy_process_group: peers may be stuck inside a collective)
            fallback_reason = bad[rank]
            print("[bench] falling back to contiguous slabs + one all-gather per channel", file=sys.stderr)
            pipe, cyc, joint = None, None, None
            b, e = slab_range(nx, world, rank)
            n_local = e - b
            loc_v = torch.empty((8, n_local, ny, nz), dtype=torch.float32, device=dev) if need_v else None
            loc_c = torch.empty((8, n_local, ny, nz), dtype=torch.float32, device=dev) if need_c else None
            locs = [t for t in (loc_v, loc_c) if t is not None]
            gather_mode = "slab (fallback)"
        else:
            torch.cuda.synchronize()
    # (the first launches after the set-up phase run below the steady clock: 16-18, 14.1, 13.6 then 13.4 ms on an idle card,
    # scripts/clock_ramp.py; three untimed launches belong to the set-up, whatever --warmup says)
    for _ in range(PREWARM_STEPS):
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(bb) for a, bb in zip(ev0, ev1)])) if args.steps else float("nan")
    if multi:
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(t[0]), float(t[1])

    # built-in spot check against the oracle (a block of this rank's result), after timing
    check = None
    selfcheck_failed = False
    if not args.no_check and rank == 0:
        from oracle import oracle as O
        from oracle.compare import compare_grids
        from ceg_hip import grids as G
        # x-planes to check: N = 1 -> a block in the middle; N > 1 -> one plane out of every rank's piece of the first and
        # of the last chunk (a misplaced block of ANY rank shows), then both compared over two y-rows
        if multi and cyc is not None:
            planes = sorted({cyclic_plan(nx, world, r, nchunks=cyc.nchunks).chunk(j)[0] + (r % cyc.m) for r in range(world) for j in (0, cyc.nchunks - 1)})
        elif multi:
            planes = sorted({min(nx - 1, slab_range(nx, world, r)[0] + k) for r in range(world) for k in (0, 1)})
        else:
            T = min(O.usable_cpus(), 16, nx)
            planes = list(range(nx // 2 - T // 2, nx // 2 - T // 2 + T))
        planes = planes[:32]
        j0 = ny // 2
        j1 = min(ny, j0 + 2)
        runs = []                                  # consecutive planes -> one oracle call each
        for i in planes:
            if runs and runs[-1][1] == i:
                runs[-1][1] = i + 1
            else:
                runs.append([i, i + 1])
        worst, failures = 0.0, []
        scratch = np.empty((8, nx, ny, nz), dtype=np.float32)
        for i0, i1 in runs:
            for need, full, what in ((need_v, full_v, "vdw"), (need_c, full_c, "coulomb")):
                if not need:
                    continue
                if what == "vdw":
                    lam, thr = G.vdw_scaling()
                    ref, _ = O.grid_vdw(w.probe_vdw, w.cset, lam, thr, i0, i1, j_begin=j0, j_end=j1, out=scratch)
                else:
                    lam, thr = G.coulomb_scaling()
                    ref, _ = O.grid_coulomb(w.probe_coulomb, w.alpha, w.cset, lam, thr, i0, i1, j_begin=j0, j_end=j1, out=scratch)
                try:       # a mismatch in ANY rank's block is recorded, the line is still printed and the process exits non-zero
                    worst = max(worst, compare_grids(full[:, i0:i1, j0:j1].cpu().numpy(), ref[:, i0:i1, j0:j1], f"bench/{what} planes {i0}:{i1}"))
                except AssertionError as exc:
                    failures.append(str(exc))
        i0, i1 = 0, len(planes)
        check = {"points": (i1 - i0) * (j1 - j0) * nz, "x_planes": planes, "max_rel_err": worst, "tol": 1e-6,
                 "ok": not failures}
        if failures:
            check["failures"] = failures[:8]
            selfcheck_failed = True
            print("[bench] SELFCHECK FAILED: " + " | ".join(failures[:8]), file=sys.stderr)

    if rank == 0:
        ms = elapsed / max(args.steps, 1) * 1e3
        value = npts / (ms * 1e-3)
        ngrids = int(need_v) + int(need_c)
        # algorithmic (compulsory) HBM bytes per launch: 32 B per point per grid written + the
        # image/atom table read once (32 B position+charge, 4 B kind)  -- SURVEY §8d
        slab_pts = n_local * ny * nz
        nlaunch = cyc.nchunks if cyc is not None else 1
        alg_bytes = 32.0 * slab_pts * ngrids + 36.0 * max(plan.num_images, w.natoms)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        # counted minimum work (exact on a sample of the grid points): in-cutoff pairs and, of those, the ones whose kind
        # has a VdW rule for this probe -- Si/Al carry no Ar rule and do no VdW work
        pw = W.count_pair_work(w)
        n_in = pw["in_cutoff_per_point"]
        n_vdw = pw["lj_per_point"] + pw["buckingham_per_point"] + pw["other_vdw_per_point"]
        f_vdw = pw["lj_per_point"] * F_LJ + (pw["buckingham_per_point"] + pw["other_vdw_per_point"]) * F_BUCK
        per_point = ((n_in if need_c else n_vdw) * F_DIST + (f_vdw if need_v else 0.0) + (n_in * F_EWALD if need_c else 0.0))
        min_flops = slab_pts * per_point
        survey_flops = slab_pts * 310.0 * ((F_DIST_SURVEY + F_LJ if need_v else 0.0) + (F_DIST_SURVEY + F_EWALD if need_c else 0.0)
                                           - (F_DIST_SURVEY if (need_v and need_c) else 0.0))
        nominal_tflops = min_flops / (kern_ms * 1e-3) / 1e12
        # Executed work, convention-free: FP64 VALU instructions of one launch by class (rocprofv3 PMC passes of this very command,
        # scripts/pmc.sh -> profiles/pmc_summary.json) -> flops = (ADD + MUL + TRANS + 2 FMA) x 64 lanes x measured lane
        # utilisation.  The instruction counts of a launch are a property of (library, workload), not of the box or the run, so they
        # may be divided by the kernel time measured HERE; the record is only used when it was taken on this library (sha256 of
        # the kernel sources).  Everything that comes from that file sits under "pmc" with its provenance.
        pmc_key = f"{args.mode}/{args.probe}/{args.n}/{world}"
        pmc = load_pmc_record(pmc_key)
        if pmc is None and world > 1:
            # no profile of the N-rank run: the one-GPU record of the same workload, its per-launch totals scaled to this rank's
            # share of the x-planes (every rank runs the same kernel on 1/N of the planes)
            one = load_pmc_record(f"{args.mode}/{args.probe}/{args.n}/1")
            if one:
                share = n_local / float(nx)
                for k in ("hbm_bytes_per_launch", "fp64_flops_per_launch", "fp64_insts_per_launch", "gpu_cycles_per_launch"):
                    if k in one:
                        one[k] = one[k] * share
                one["scaled_from"] = f"{args.mode}/{args.probe}/{args.n}/1 x {share:.4f} (this rank's share of the x-planes; per STEP, i.e. over all its chunk launches)"
                one.pop("kernel_ms_in_profile", None)
                pmc = one
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        exec_flops = pmc.get("fp64_flops_per_launch") if pmc else None
        exec_tflops = exec_flops / (kern_ms * 1e-3) / 1e12 if exec_flops else None
        frac_issue = pmc.get("frac_issue") if pmc else None
        eff_clock = (pmc["gpu_cycles_per_launch"] / (pmc["kernel_ms_in_profile"] * 1e-3) / 1e9
                     if pmc and pmc.get("gpu_cycles_per_launch") and pmc.get("kernel_ms_in_profile") else None)
        transport = "RCCL" if args.backend == "nccl" else "gloo (rehearsal)"
        out = {
            "metric": "grid-points/sec", "value": value, "unit": "grid-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": PREWARM_STEPS, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": w.name, "grid_points": npts, "framework_atoms": w.natoms,
                       "lattice_images": plan.num_images, "grids_per_step": ngrids, "mode": args.mode,
                       "algo": "culled" if (algo != _abi.ALGO_BRUTEFORCE and plan.can_cull) else "bruteforce",
                       "parallelism": (f"block-cyclic x-chunks over {world} GPUs ({cyc.nchunks} chunks of {cyc.m} planes per rank), each chunk all-gathered over {transport} ({gather_mode}) while the next is computed" if cyc is not None
                                       else f"x-slab sharding over {world} GPU(s)" + (f", {transport} all-gather of slabs" if world > 1 else ""))},
            # neither HBM nor MFMA bounds this path (FP64 vector ALU does); `roofline` is the binding
            # one -- minimum-work flops / t against the FP64 vector peak -- and the HBM view sits alongside
            "roofline": {"bound": "valu_fp64", "achieved": exec_tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (exec_tflops / FP64_VALU_PEAK_TFLOPS) if exec_tflops else None, "traffic": traffic,
                         "frac_executed": (exec_tflops / FP64_VALU_PEAK_TFLOPS) if exec_tflops else None,
                         "frac_stale": bool(pmc and pmc.get("stale")),
                         "frac_issue": frac_issue,
                         "frac_nominal": nominal_tflops / FP64_VALU_PEAK_TFLOPS,
                         "kernel": f"k_culled<{args.mode}>" if plan.can_cull and algo != _abi.ALGO_BRUTEFORCE else f"k_bruteforce<{args.mode}>",
                         "kernel_ms": kern_ms, "launches_per_step": nlaunch,
                         "executed_fp64_flops_per_launch": exec_flops, "effective_clock_ghz_in_profile": eff_clock,
                         "pmc": pmc,
                         "nominal": {"achieved": nominal_tflops, "algorithmic_flops": min_flops,
                                     "counted_work": dict(pw, flops_per_point=per_point, f_dist=F_DIST, f_lj=F_LJ, f_buckingham=F_BUCK, f_ewald=F_EWALD),
                                     "frac_survey_convention": survey_flops / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                     "note": "counted minimum work in SURVEY 8d's nominal flops (LJ 50, Buckingham 95, Ewald 140 incl. exp = 20, "
                                             "erfc = 40; distance from an image list 8): a work RATE for comparing algorithms, not a hardware "
                                             "fraction -- the kernel evaluates r^2-indexed tables and executes far fewer flops per pair"},
                         "pair_checks_per_s": slab_pts * float(w.natoms) / (kern_ms * 1e-3),
                         "flops": "frac = frac_executed = executed FP64 flops of one launch (PMC: (ADD_F64 + MUL_F64 + TRANS_F64 + 2 FMA_F64) x 64 x "
                                  "lane utilisation, instruction counts of THIS library on THIS workload from profiles/pmc_summary.json) / kernel "
                                  "time measured in this run / peak; frac_issue = VALU issue utilisation x lane utilisation of the profiled "
                                  "run (share of lane-issue slots doing work); frac_nominal = counted minimum work in nominal flops / time / peak "
                                  "(round 2's headline; kept for continuity); peak = 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz "
                                  "(FP64 MFMA has the same peak on MI355X; no MFMA is used)",
                         "note": "the contract's bound enum is hbm|mfma; this path is an FP64 pairwise reduction bound by the vector ALU "
                                 "(>= 1e3 flop per compulsory HBM byte), see roofline_hbm for the byte view"},
            "roofline_hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": alg_bytes,
                             "measured_GBps": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else None,
                             "traffic_source": (pmc or {}).get("source"),
                             "note": "algorithmic bytes (32 B/point/grid written once + 36 B/image) over the kernel time; ~1 % by construction; "
                                     "measured_GBps = PMC traffic of the profiled run (WRITE_SIZE + 2 x FETCH_SIZE, profiles/pmc_summary.json) "
                                     "over this run's kernel time"},
            "selfcheck": check,
        }
        if multi:
            # gather time reported separately.  compute_ms = span of this rank's kernels
            # (max over ranks); what is left of the step is the part of the exchange that was not hidden
            out["exchange"] = {"compute_ms": kern_ms, "exposed_ms": max(0.0, ms - kern_ms),
                               "mode": gather_mode if (cyc is not None or fallback_reason) else "slab", "autotune_ms": autotune, "fallback_reason": fallback_reason,
                               "bytes_gathered_per_rank": 32.0 * npts * ngrids * (world - 1) / world, "backend": "nccl (RCCL over xGMI)" if args.backend == "nccl" else "gloo (rehearsal)"}
        if world == 1 and not args.force_exchange and not args.no_oneshot:
            # the timed buffers are no longer needed: give the one-shot pipelines the card (they keep their own device slab)
            del full_v, full_c, loc_v, loc_c, fulls, locs
            torch.cuda.empty_cache()
            try:
                out["oneshot"] = oneshot_record(w)
            except Exception as exc:          # noqa: BLE001 -- a failure here must not cost the line
                out["oneshot"] = {"error": repr(exc)}
        if world == 1 and args.cpu_rows != 0:
            out["cpu_baseline"] = cpu_baseline(w, args.mode, args.cpu_rows)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    plan.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if selfcheck_failed:
        sys.exit(1)


if __name__ == "__main__":
    main()
