"""Quick A/B aid: kernel time of the multi-probe fused pair / VdW pair / Coulomb on the roofline workload for the library in CEG_HIP_LIB."""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import torch
from ceg_hip import workloads as W
from ceg_hip.plan import MultiGridPlan
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
ws = [W.roofline_workload(a, 255) for a in ("C_co2", "O_co2")]
w = ws[0]
nx, ny, nz = w.cset.npoints
cs = nx * ny * nz
bufs = [torch.empty((8, nx, ny, nz), dtype=torch.float32, device=dev) for _ in range(3)]
mp = MultiGridPlan(w.cset, [x.probe_vdw for x in ws], w.probe_coulomb, w.alpha)
s = torch.cuda.current_stream().cuda_stream
def timed(fn):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
ptrs = [b.data_ptr() for b in bufs[:2]]
print(os.environ.get("CEG_HIP_LIB", "in-tree"), " fused pair %.3f ms   VdW pair %.3f ms   Coulomb alone %.3f ms   probe 0 alone %.3f ms" % (
    timed(lambda: mp.build(ptrs, bufs[2].data_ptr(), cs, 0, nx, 0, s)), timed(lambda: mp.build(ptrs, 0, cs, 0, nx, 0, s)),
    timed(lambda: mp.build([0, 0], bufs[2].data_ptr(), cs, 0, nx, 0, s)), timed(lambda: mp.build([ptrs[0], 0], 0, cs, 0, nx, 0, s))), flush=True)
