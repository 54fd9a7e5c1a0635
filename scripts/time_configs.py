"""Kernel timings of the BASELINE.json configs other than the bench workload.  Prints one line per measurement."""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import numpy as np, torch
from ceg_hip import workloads as W, grids as G, _abi
from ceg_hip.plan import GridPlan

def time_plan(w, mode, reps=5):
    t0 = time.perf_counter()
    plan = GridPlan(w.cset, w.probe_vdw, w.probe_coulomb, w.alpha)
    t_plan = time.perf_counter() - t0
    nx, ny, nz = w.cset.npoints
    dev = torch.device("cuda", 0)
    v = torch.empty((8, nx, ny, nz), dtype=torch.float32, device=dev)
    c = torch.empty_like(v)
    s = torch.cuda.current_stream().cuda_stream
    def launch():
        if mode == "fused": plan.build_fused(v.data_ptr(), c.data_ptr(), nx*ny*nz, 0, nx, 0, 0, s)
        elif mode == "vdw": plan.build_vdw(v.data_ptr(), nx*ny*nz, 0, nx, 0, 0, s)
        else: plan.build_coulomb(c.data_ptr(), nx*ny*nz, 0, nx, 0, 0, s)
    launch(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    n = nx*ny*nz
    print(f"{w.name:70s} {mode:8s} points {n:9d} atoms {w.natoms:6d} images {plan.num_images:6d} kernel {ms:8.3f} ms  {n/ms*1e3:.3e} pts/s  plan_create {t_plan*1e3:.1f} ms", flush=True)
    plan.close()

for fw, atom, sp in (("CHA_1.4_3b4eeb96", "Na", 0.5), ("CHA_1.4_3b4eeb96", "Ar", 0.1), ("CHA_1.4_3b4eeb96", "Na", 0.1), ("CHA_1.4_3b4eeb96", "Na", 0.15), ("CIT-7", "Na", 0.15)):
    w = W.fixture_workload(fw, atom, sp)
    for mode in ("vdw", "coulomb", "fused"):
        time_plan(w, mode)
w = W.roofline_workload("Na", 255)
for mode in ("vdw", "fused"):
    time_plan(w, mode)
# (the PCIe-inclusive one-shot entry points are timed by scripts/time_oneshot.py)
