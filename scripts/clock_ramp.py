"""Kernel time of the fused roofline launch step by step after 0 / 5 / 15 s of GPU idleness: does the clock need a ramp-up that\nbench.py's warm-up steps should cover?"""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__)); sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import numpy as np, torch
from ceg_hip import workloads as W, _abi
from ceg_hip.plan import GridPlan
from ceg_hip import grids as G
w = W.roofline_workload("Ar", 255)
nx, ny, nz = w.cset.npoints
plan = GridPlan(w.cset, w.probe_vdw, w.probe_coulomb, w.alpha)
dev = torch.device("cuda:0")
fv = torch.empty((8, nx, ny, nz), dtype=torch.float32, device=dev); fc = torch.empty_like(fv)
s = torch.cuda.current_stream().cuda_stream
lamv, thrv = G.vdw_scaling(); lamc, thrc = G.coulomb_scaling()
def step():
    plan.build_fused(fv.data_ptr(), fc.data_ptr(), nx * ny * nz, 0, nx, 0, _abi.ALGO_AUTO, s)
for idle in (0.0, 5.0, 15.0):
    torch.cuda.synchronize(); time.sleep(idle)
    ts = []
    for k in range(40):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); step(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"after {idle:4.1f} s idle: steps 1-5 " + " ".join(f"{t:.2f}" for t in ts[:5]) + f" | 6-10 mean {np.mean(ts[5:10]):.2f} | 11-20 {np.mean(ts[10:20]):.2f} | 21-40 {np.mean(ts[20:]):.2f} ms")
