"""PCIe-inclusive rate of the one-shot entry points (host buffers in, host grid out), roofline workload."""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import numpy as np
from ceg_hip import workloads as W, grids as G
w = W.roofline_workload("Ar", 255)
for name, fn in (("ceg_grid_vdw", lambda: G.build_vdw_array(w.probe_vdw, w.cset)), ("ceg_grid_coulomb", lambda: G.build_coulomb_array(w.probe_coulomb, w.alpha, w.cset))):
    ts = []
    for rep in range(4):
        t = time.perf_counter(); g = fn(); ts.append(time.perf_counter() - t); del g
    print(f"one-shot {name}: " + ", ".join(f"{x*1e3:.1f}" for x in ts) + f" ms wall per call (16777216 points, 537 MB result into a fresh host array; "
          f"copy threads {os.environ.get('CEG_HIP_COPY_THREADS', '8')}) -> best {16777216/min(ts):.3e} pts/s", flush=True)
