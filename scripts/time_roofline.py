"""Kernel time of the roofline workload (256^3 x 11 664 atoms) per mode / probe / hot-loop variant, in one process:
    python scripts/time_roofline.py [reps]
CEG_HIP_NO_EW2=1 (read at plan creation) switches the r^2-indexed Ewald tables off -> round 1's erfcx variant."""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import torch
from ceg_hip import workloads as W
from ceg_hip.plan import GridPlan

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
for probe in ("Ar", "Na"):
    w = W.roofline_workload(probe, 255)
    nx, ny, nz = w.cset.npoints
    v = torch.empty((8, nx, ny, nz), dtype=torch.float32, device=dev)
    c = torch.empty_like(v)
    for ew2 in (True, False):
        if ew2:
            os.environ.pop("CEG_HIP_NO_EW2", None)
        else:
            os.environ["CEG_HIP_NO_EW2"] = "1"
        plan = GridPlan(w.cset, w.probe_vdw, w.probe_coulomb, w.alpha)
        s = torch.cuda.current_stream().cuda_stream
        for mode in ("fused", "coulomb", "vdw"):
            if mode == "vdw" and not ew2:
                continue
            def launch():
                if mode == "fused": plan.build_fused(v.data_ptr(), c.data_ptr(), nx * ny * nz, 0, nx, 0, 0, s)
                elif mode == "vdw": plan.build_vdw(v.data_ptr(), nx * ny * nz, 0, nx, 0, 0, s)
                else: plan.build_coulomb(c.data_ptr(), nx * ny * nz, 0, nx, 0, 0, s)
            launch(); launch(); torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); launch(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            print(f"{probe} {mode:8s} {'r2-tables' if ew2 else 'erfcx    '}  min {min(ts):7.3f} ms  mean {sum(ts)/len(ts):7.3f} ms  "
                  f"{nx*ny*nz/min(ts)*1e3:.3e} pts/s", flush=True)
        plan.close()
os.environ.pop("CEG_HIP_NO_EW2", None)
