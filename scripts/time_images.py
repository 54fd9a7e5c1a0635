"""Cost of the lattice-image list of a NEW framework (image-cache miss) inside plan creation: device build (csrc/ceg_images.hip, round 4)
against the host build (CEG_HIP_IMAGES_ON_HOST=1), roofline workload (11 664 atoms -> 37 460 images) and the CHA fixture.  Every plan gets
slightly different atom positions so that the cache misses; the first plan of the process (context creation, code-object load) is not
counted.  Wall time of ceg_plan_create, and the "images" share from CEG_HIP_TRACE stamps printed by the library."""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import numpy as np
from ceg_hip import workloads as W
from ceg_hip.plan import GridPlan

def run(label, w, n=8):
    for where in ("device", "host"):
        if where == "host":
            os.environ["CEG_HIP_IMAGES_ON_HOST"] = "1"
        else:
            os.environ.pop("CEG_HIP_IMAGES_ON_HOST", None)
        ts = []
        base = np.array(w.probe_vdw.positions, dtype=np.float64)
        for k in range(n + 1):
            w.probe_vdw.positions = base + 1e-6 * (k + 1 + (100 if where == "host" else 0))     # a new framework for the cache
            w.probe_coulomb.positions = w.probe_vdw.positions
            t = time.perf_counter()
            p = GridPlan(w.cset, w.probe_vdw, w.probe_coulomb, w.alpha)
            ts.append((time.perf_counter() - t) * 1e3)
            nimg = p.num_images
            p.close()
        w.probe_vdw.positions = base
        w.probe_coulomb.positions = base
        print(f"{label}: plan creation with the image list built on the {where}: " + ", ".join(f"{x:.3f}" for x in ts[1:]) +
              f" ms (min {min(ts[1:]):.3f}); {nimg} images", flush=True)

run("roofline workload (11 664 atoms)", W.roofline_workload("Ar", 255))
run("CHA fixture @ 0.15 A (972 atoms)", W.fixture_workload("CHA_1.4_3b4eeb96", "Ar", 0.15))
