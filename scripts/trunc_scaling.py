import os, sys, math
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import numpy as np, torch
from ceg_hip import workloads as W
from ceg_hip.plan import GridPlan
from ceg_hip.hostmirror.raspa import RASPASystem
from ceg_hip.hostmirror.probes import ProbeSystem

def t(w, mode="vdw", reps=5):
    plan = GridPlan(w.cset, w.probe_vdw, w.probe_coulomb, w.alpha)
    nx, ny, nz = w.cset.npoints
    dev = torch.device("cuda", 0)
    v = torch.empty((8, nx, ny, nz), dtype=torch.float32, device=dev); c = torch.empty_like(v)
    s = torch.cuda.current_stream().cuda_stream
    n = nx*ny*nz
    def launch():
        if mode == "vdw": plan.build_vdw(v.data_ptr(), n, 0, nx, 0, 0, s)
        else: plan.build_coulomb(c.data_ptr(), n, 0, nx, 0, 0, s)
    launch(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    print(f"{mode:8s} atoms {w.natoms:6d} images {plan.num_images:6d} {ms:8.3f} ms", flush=True)
    plan.close()

base = W.roofline_workload("Ar", 255)
for keep in (11664, 10000, 8000, 5832, 3000):
    w = W.roofline_workload("Ar", 255, truncate=keep) if keep < 11664 else base
    print("first", keep, end=": "); t(w, "vdw"); print("first", keep, end=": "); t(w, "coulomb")
# uniform thinning
fw = base.framework
for step in (7, 2):
    m = np.arange(len(fw.position)) % step != 0
    f2 = RASPASystem(fw.mat, fw.position[m], [s for s, k in zip(fw.atomic_symbol, m) if k], fw.atomic_mass[m], fw.atomic_charge[m], False)
    w = W.Workload("thin", f2, base.forcefield, base.cset, ProbeSystem.build(f2, base.forcefield, "Ar"), ProbeSystem.build(f2, base.forcefield), base.alpha)
    print("thin", step, end=": "); t(w, "vdw"); print("thin", step, end=": "); t(w, "coulomb")
