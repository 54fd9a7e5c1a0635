"""Throughput of the batched interpolation kernel (row f1) on the grid the reference's tests build
(CHA fixture, 0.15 A, Ar VdW grid), random positions; CPU oracle (literal COEFF*X) beside it."""
import os, sys, time, math
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..')]
import numpy as np, torch
import ceg_hip as ceg
from ceg_hip import workloads as W, grids as G, _abi
from ceg_hip.plan import GridPlan
from ceg_hip.interp import GridInterpolator
from oracle import oracle as O
w = W.fixture_workload("CHA_1.4_3b4eeb96", "Ar", 0.15)
nx, ny, nz = w.cset.npoints
dev = torch.device("cuda", 0)
lib = _abi.load_library()
plan = GridPlan(w.cset, w.probe_vdw, None, 0.0)
d_grid = torch.empty((8, nx, ny, nz), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
plan.build_vdw(d_grid.data_ptr(), nx*ny*nz, 0, nx, 0, 0, s)
_abi.check(lib, lib.ceg_scale_grid_device(d_grid.data_ptr(), d_grid.numel(), ceg.GRID_TO_KELVIN, 0, s))
torch.cuda.synchronize()
eg = G.EnergyGrid(w.cset, (1, 1, 1), math.inf, True, d_grid.cpu().numpy())
it = GridInterpolator(eg, device_ptr=d_grid.data_ptr())
n = 1 << 24
g = torch.Generator(device=dev); g.manual_seed(1)
pts = (torch.rand((n, 3), dtype=torch.float64, device=dev, generator=g) * 60.0 - 15.0).contiguous()
out = torch.empty(n, dtype=torch.float64, device=dev)
it.on_device(pts.data_ptr(), n, out.data_ptr(), s); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): it.on_device(pts.data_ptr(), n, out.data_ptr(), s)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"GPU interpolate_grid: {n} random points on a {nx}x{ny}x{nz} grid ({d_grid.numel()*4/1e6:.0f} MB): {ms:.3f} ms -> {n/ms*1e3:.3e} points/s; "
      f"gathered bytes {n*256/ms*1e3/1e9:.0f} GB/s (64 floats/point)")
m = 200000
hp = pts[:m].cpu().numpy()
t = time.perf_counter(); ref = O.interpolate_points(eg, hp); dt = time.perf_counter() - t
print(f"CPU oracle (literal 64x64 COEFF product, {O.max_threads()} threads): {m} points in {dt*1e3:.1f} ms -> {m/dt:.3e} points/s")
got = out[:m].cpu().numpy()
blk = ref == 1e100
print("blocked pattern equal:", np.array_equal(got == 1e100, blk), " max rel err:", float(np.max(np.abs(got[~blk]-ref[~blk])/np.maximum(np.abs(ref[~blk]), 1e-3*np.median(np.abs(ref[~blk]))))))
