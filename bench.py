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
    ap.add_argument("--gather", choices=("auto", "staged", "inplace", "p2p"), default="auto",
                    help="how a gathered chunk is placed (N > 1); auto: staged and inplace are each timed on one untimed step and the "
                         "faster one (max over ranks) is used")
    ap.add_argument("--chunks", type=int, default=8,
                    help="x-chunks per rank pipelined with the all-gather (N > 1); 8 from the exchange model of profiles/r02_rank_emulation.txt")
    ap.add_argument("--n", "--dims", dest="n", type=int, default=255, help="dims per axis (odd); grid has (n+1)^3 points")
    ap.add_argument("--cpu-rows", type=int, default=-1, help="y-rows per x-plane the CPU baseline times, one plane per thread (-1 = auto ~12 s, 0 = skip)")
    ap.add_argument("--no-check", action="store_true", help="skip the built-in oracle spot check")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend; gloo is a rehearsal aid: several ranks may then share one GPU "
                         "(device = LOCAL_RANK mod device count), which RCCL refuses")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N = 1 only: run the N > 1 code path (RCCL group of one rank, chunked build, collectives issued) -- a rehearsal")
    return ap.parse_args()


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
    from ceg_hip.distributed import PipelinedGather, allgather_grid, cyclic_plan, slab_range
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

    if pipe is not None and args.gather == "auto" and pipe.exchange:
        # pick, on THIS node, the number of chunks per rank and the placement of the gathered chunks: two untimed steps per
        # candidate (the first warms it up), max over the ranks.  Fewer chunks = fewer, larger collectives and launches but a
        # longer exposed first / last chunk; which side wins depends on the collective's launch latency, which only a real
        # run on the links shows.  Both placements are plain all_gather_into_tensor calls (no point-to-point schedule).
        autotune = {}
        step()
        tried = []
        for nch in sorted({args.chunks, max(1, args.chunks // 2), max(1, args.chunks // 4)}, reverse=True):
            cand = cyclic_plan(nx, world, rank, nchunks=nch)
            if cand is None or cand.nchunks in tried:
                continue
            tried.append(cand.nchunks)
            cyc = cand
            joint, loc_v, loc_c = cyclic_buffers(cyc)
            locs = [t for t in (loc_v, loc_c) if t is not None]
            for mode in ("staged", "inplace"):
                pipe = make_pipe(mode)
                step()
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t_a = time.perf_counter()
                step()
                torch.cuda.synchronize()
                t = torch.tensor([time.perf_counter() - t_a], dtype=torch.float64, device=dev)
                if world > 1:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                autotune[f"{cyc.nchunks} chunks, {mode}"] = float(t[0]) * 1e3
        best = min(autotune, key=autotune.get)
        cyc = cyclic_plan(nx, world, rank, nchunks=int(best.split()[0]))
        gather_mode = best.split(", ")[1]
        joint, loc_v, loc_c = cyclic_buffers(cyc)
        locs = [t for t in (loc_v, loc_c) if t is not None]
        pipe = make_pipe(gather_mode)
        n_local = cyc.n_local
    elif args.gather == "auto":
        gather_mode = "staged"
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
        worst = 0.0
        scratch = np.empty((8, nx, ny, nz), dtype=np.float32)
        for i0, i1 in runs:
            if need_v:
                lam, thr = G.vdw_scaling()
                ref, _ = O.grid_vdw(w.probe_vdw, w.cset, lam, thr, i0, i1, j_begin=j0, j_end=j1, out=scratch)
                worst = max(worst, compare_grids(full_v[:, i0:i1, j0:j1].cpu().numpy(), ref[:, i0:i1, j0:j1], f"bench/vdw planes {i0}:{i1}"))
            if need_c:
                lam, thr = G.coulomb_scaling()
                ref, _ = O.grid_coulomb(w.probe_coulomb, w.alpha, w.cset, lam, thr, i0, i1, j_begin=j0, j_end=j1, out=scratch)
                worst = max(worst, compare_grids(full_c[:, i0:i1, j0:j1].cpu().numpy(), ref[:, i0:i1, j0:j1], f"bench/coulomb planes {i0}:{i1}"))
        i0, i1 = 0, len(planes)
        check = {"points": (i1 - i0) * (j1 - j0) * nz, "x_planes": planes, "max_rel_err": worst, "tol": 1e-6}

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
        tflops = min_flops / (kern_ms * 1e-3) / 1e12
        # PMC-derived figures of the same command (separate rocprofv3 --pmc passes, scripts/pmc.sh -> profiles/pmc_summary.json)
        traffic = valu_issue = lane_util = None
        pmc_key = f"{args.mode}/{args.probe}/{args.n}/{world}"
        prof = ROOT / "profiles" / "pmc_summary.json"
        if prof.exists():
            try:
                rec = json.loads(prof.read_text()).get(pmc_key) or {}
                traffic = rec.get("hbm_bytes_per_launch")
                valu_issue = rec.get("valu_issue_util")
                lane_util = rec.get("lane_util")
            except Exception:
                pass
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
            "roofline": {"bound": "valu_fp64", "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / FP64_VALU_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": f"k_culled<{args.mode}>" if plan.can_cull and algo != _abi.ALGO_BRUTEFORCE else f"k_bruteforce<{args.mode}>",
                         "kernel_ms": kern_ms, "launches_per_step": nlaunch, "algorithmic_flops": min_flops,
                         "counted_work": dict(pw, flops_per_point=per_point, f_dist=F_DIST, f_lj=F_LJ, f_buckingham=F_BUCK, f_ewald=F_EWALD),
                         "valu_issue_util": valu_issue, "lane_util": lane_util, "pmc_key": pmc_key,
                         "frac_survey_convention": survey_flops / (kern_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                         "pair_checks_per_s": slab_pts * float(w.natoms) / (kern_ms * 1e-3),
                         "flops": "counted minimum work: sampled grid points x (in-cutoff images x 8 [distance from an image list] + images with a "
                                  "VdW rule x 50 LJ | 95 Buckingham + in-cutoff images x 140 real-space Ewald), nominal convention of SURVEY 8d "
                                  "(exp = 20, erfc = 40 flops); frac_survey_convention = round 1's 310 neighbours x (47 + 50 + 140) for continuity; "
                                  "valu_issue_util / lane_util = PMC counters of the same command (profiles/pmc_summary.json); "
                                  "peak = 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz (FP64 MFMA has the same peak on MI355X; no MFMA is used)",
                         "note": "the contract's bound enum is hbm|mfma; this path is an FP64 pairwise reduction bound by the vector ALU "
                                 "(>= 1e3 flop per compulsory HBM byte), see roofline_hbm for the byte view"},
            "roofline_hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": alg_bytes,
                             "measured_GBps": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else None,
                             "note": "algorithmic bytes (32 B/point/grid written once + 36 B/image) over the kernel time; ~1 % by construction; "
                                     "measured_GBps = PMC traffic (WRITE_SIZE + 2 x FETCH_SIZE, profiles/pmc_summary.json) over the same time"},
            "selfcheck": check,
        }
        if multi:
            # gather time reported separately.  compute_ms = span of this rank's kernels
            # (max over ranks); what is left of the step is the part of the exchange that was not hidden
            out["exchange"] = {"compute_ms": kern_ms, "exposed_ms": max(0.0, ms - kern_ms), "mode": gather_mode if cyc is not None else "slab", "autotune_ms": autotune,
                               "bytes_gathered_per_rank": 32.0 * npts * ngrids * (world - 1) / world, "backend": "nccl (RCCL over xGMI)" if args.backend == "nccl" else "gloo (rehearsal)"}
        if world == 1 and args.cpu_rows != 0:
            out["cpu_baseline"] = cpu_baseline(w, args.mode, args.cpu_rows)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    plan.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
