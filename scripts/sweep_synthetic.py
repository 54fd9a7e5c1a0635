"""SURVEY 8d sweep: fully synthetic 40 A cube, N random atoms, 128^3 grid; and the roofline run R with
the Ar / Na probes and the exact-10k truncation.  Kernel time per mode, points/s, in-cutoff pair rate."""
import os, sys, math
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import numpy as np, torch
from ceg_hip import workloads as W
from ceg_hip.plan import GridPlan

def run(w, reps=5):
    plan = GridPlan(w.cset, w.probe_vdw, w.probe_coulomb, w.alpha)
    nx, ny, nz = w.cset.npoints
    dev = torch.device("cuda", 0)
    v = torch.empty((8, nx, ny, nz), dtype=torch.float32, device=dev); c = torch.empty_like(v)
    s = torch.cuda.current_stream().cuda_stream
    n = nx * ny * nz
    vol = abs(np.linalg.det(w.probe_coulomb.mat))
    ncut = len(w.probe_coulomb.positions) / vol * 4.0 / 3.0 * math.pi * 12.0 ** 3
    for mode in ("vdw", "coulomb", "fused"):
        def launch():
            if mode == "fused": plan.build_fused(v.data_ptr(), c.data_ptr(), n, 0, nx, 0, 0, s)
            elif mode == "vdw": plan.build_vdw(v.data_ptr(), n, 0, nx, 0, 0, s)
            else: plan.build_coulomb(c.data_ptr(), n, 0, nx, 0, 0, s)
        launch(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): launch()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{w.name:95s} {mode:8s} atoms {w.natoms:6d} images {plan.num_images:6d} n_cut {ncut:6.0f}  {ms:8.3f} ms  {n/ms*1e3:.3e} pts/s  "
              f"{n*ncut/ms*1e3:.3e} in-cutoff pairs/s", flush=True)
    plan.close()

for N in (250, 1000, 2000, 4000, 8000):
    run(W.synthetic_workload(N, 127))
run(W.roofline_workload("Ar", 255))
run(W.roofline_workload("Ar", 255, truncate=10000))
run(W.roofline_workload("Na", 255))
