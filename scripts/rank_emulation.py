"""What each rank of an N-GPU run computes, run alone on ONE GPU (SURVEY 8e / BASELINE config 4 without the node):
for N = 1, 2, 4, 8 and 1 / 2 / 4 / 8 block-cyclic chunks per rank, the exact launch set of every rank r
(ceg_hip.distributed.cyclic_plan, compact [8, m, ny, nz] blocks, the bench's fused build) is timed with HIP events ->
t_rank(N, r), the compute-side strong-scaling efficiency t(1) / (N max_r t_rank) and the launch geometry
(workgroups per launch against the resident slots).  Then the exchange model: every rank receives (N-1)/N of the two
grids, one block per peer and per xGMI link (full mesh: 7 links x 76.8 GB/s per direction per GPU), chunk j's gather
overlapping chunk j+1's build.

    python scripts/rank_emulation.py [probe] [reps]
"""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import torch
from ceg_hip import workloads as W
from ceg_hip.distributed import cyclic_plan
from ceg_hip.plan import GridPlan

probe = sys.argv[1] if len(sys.argv) > 1 else "Ar"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
LINK_GBS = 76.8            # one xGMI link, one direction (153.6 GB/s bidirectional), MI355X
LINK_EFF = 0.85            # payload efficiency assumed for large transfers
NW, SLOTS = 8, 512         # waves per workgroup of the fused kernel, resident workgroups (256 CUs x 2)

dev = torch.device("cuda", 0)
w = W.roofline_workload(probe, 255)
nx, ny, nz = w.cset.npoints
plane = ny * nz
plan = GridPlan(w.cset, w.probe_vdw, w.probe_coulomb, w.alpha)
s = torch.cuda.current_stream().cuda_stream
print(f"# {w.name}: fused build, kernel times of one rank's launches run alone on one MI355X (min of {reps})")


def time_rank(cyc):
    v = torch.empty((cyc.nchunks, 8, cyc.m, ny, nz), dtype=torch.float32, device=dev)
    c = torch.empty_like(v)
    def run():
        for j in range(cyc.nchunks):
            b, e = cyc.chunk(j)
            plan.build_fused(v[j].data_ptr(), c[j].data_ptr(), cyc.m * plane, b, e, b, 0, s)
    run(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


t1 = None
rows = []
for N in (1, 2, 4, 8):
    for chunks in (1, 2, 4, 8):
        cyc0 = cyclic_plan(nx, N, 0, nchunks=chunks)
        if cyc0 is None or cyc0.nchunks != chunks:
            continue
        ts = [time_rank(cyclic_plan(nx, N, r, nchunks=chunks)) for r in range(N)]
        tmax, tmean = max(ts), sum(ts) / N
        if N == 1 and chunks == 1:
            t1 = tmax
        tiles = (cyc0.m // 4) * ((ny + 3) // 4) * ((nz + 3) // 4)
        wgs = (tiles + NW - 1) // NW
        eff = t1 / (N * tmax)
        # exchange: per chunk every rank receives one [8, m, ny, nz] block of both grids from each peer, each on its own link
        link_bytes = 2 * 32.0 * cyc0.m * plane
        te = link_bytes / (LINK_GBS * 1e9 * LINK_EFF) * 1e3 if N > 1 else 0.0
        tc = tmax / chunks
        step = tc + (chunks - 1) * max(tc, te) + te
        rows.append((N, chunks, cyc0.m, wgs, tmax, tmean, eff, te * chunks, step, t1 / step))
        print(f"N={N} chunks/rank={chunks} planes/chunk={cyc0.m:3d} workgroups/launch={wgs:5d} ({wgs / SLOTS:5.2f} x resident slots)  "
              f"t_rank max {tmax:7.3f} ms mean {tmean:7.3f} ms  compute efficiency {eff:5.3f}  |  exchange/rank {te * chunks:6.3f} ms "
              f"-> modelled step {step:6.3f} ms = {t1 / step:4.2f}x", flush=True)
plan.close()
print("# model: step = t_chunk + (chunks - 1) max(t_chunk, t_gather_chunk) + t_gather_chunk; t_gather_chunk = bytes of one peer block of both grids / "
      f"({LINK_GBS} GB/s x {LINK_EFF}); placement copies (a strided device copy of the gathered chunk, ~5 TB/s) not included")
best = {}
for r in rows:
    if r[0] not in best or r[9] > best[r[0]][9]:
        best[r[0]] = r
print("# best chunking per N: " + "; ".join(f"N={n}: {b[1]} chunks -> {b[9]:.2f}x" for n, b in sorted(best.items())))
