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
# the same calls into a page-locked result array of the library (ceg_host_grid_alloc): every chunk is copied D2H straight to its place
out = G.alloc_host_grid(w.cset)
for name, fn in (("ceg_grid_vdw", lambda: G.build_vdw_array(w.probe_vdw, w.cset, out=out)), ("ceg_grid_coulomb", lambda: G.build_coulomb_array(w.probe_coulomb, w.alpha, w.cset, out=out))):
    ts = []
    for rep in range(4):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    print(f"one-shot {name} into a page-locked array of the library: " + ", ".join(f"{x*1e3:.1f}" for x in ts) + f" ms wall per call -> best {16777216/min(ts):.3e} pts/s", flush=True)
del out
# the same builds with the grid left in device memory (ceg_grid_*_device): no D2H; with CEG_HIP_OVERSUBSCRIBE=1 the 2 / 4 slabs
# share the one card, which exercises the peer-copy assembly (same-device copies here, xGMI on a multi-GPU node)
import torch
for name, fn in (("ceg_grid_vdw_device", lambda ng: G.build_vdw_device(w.probe_vdw, w.cset, ngpus=ng)),
                 ("ceg_grid_coulomb_device", lambda ng: G.build_coulomb_device(w.probe_coulomb, w.alpha, w.cset, ngpus=ng))):
    for ng in (1, 2, 4):
        if ng > 1 and torch.cuda.device_count() < ng:
            os.environ["CEG_HIP_OVERSUBSCRIBE"] = "1"
        ts = []
        for rep in range(4):
            t = time.perf_counter(); g = fn(ng); ts.append(time.perf_counter() - t); del g
        os.environ.pop("CEG_HIP_OVERSUBSCRIBE", None)
        print(f"one-shot {name}, {ng} slab(s){' on one card' if ng > torch.cuda.device_count() else ''}: " + ", ".join(f"{x*1e3:.1f}" for x in ts) +
              f" ms wall per call (grid stays on the GPU) -> best {16777216/min(ts):.3e} pts/s", flush=True)
