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
# roofline.  Algorithmic bytes per position: 8 corners x 8 channels x 4 B = 256 B read (the node-major copy makes
# them 4 contiguous 64-B pieces: the two z neighbours of an (x, y) corner pair are adjacent) + 24 B position in + 8 B energy out.
# The 270 MB grid exceeds the 256 MiB Infinity Cache and the positions are random: every piece is an HBM access; if the fabric moves
# whole 128-B lines for a 64-B piece the traffic is 2 x that.  Bound: HBM (8 TB/s), not the ALU (64 weighted terms = ~250 flops per
# position = 2.3 TFLOP/s at this rate, 3 % of the FP64 peak).
alg = n * (256 + 32) / (ms * 1e-3)
print(f"roofline k_interpolate: algorithmic {256 + 32} B/position -> {alg/1e12:.2f} TB/s = {alg/8e12:.2f} of the 8 TB/s HBM peak "
      f"(with 128-B line granularity of the 64-B pieces: {n * (512 + 32) / (ms * 1e-3) / 8e12:.2f}); FP64 work ~250 flops/position = "
      f"{n * 250 / (ms * 1e-3) / 78.6e12:.3f} of the vector peak -> HBM / gather-fabric bound")
m = 200000
hp = pts[:m].cpu().numpy()
t = time.perf_counter(); ref = O.interpolate_points(eg, hp); dt = time.perf_counter() - t
print(f"CPU oracle (literal 64x64 COEFF product, {O.max_threads()} threads): {m} points in {dt*1e3:.1f} ms -> {m/dt:.3e} points/s")
got = out[:m].cpu().numpy()
blk = ref == 1e100
print("blocked pattern equal:", np.array_equal(got == 1e100, blk), " max rel err:", float(np.max(np.abs(got[~blk]-ref[~blk])/np.maximum(np.abs(ref[~blk]), 1e-3*np.median(np.abs(ref[~blk]))))))
# the same number of positions on a regular fractional lattice of the cell, last index fastest -- the order of energy_grid
# (grids.jl:394-419: iA, iB, iC nested, iC innermost) and of ceg_hip.energy.GpuEnergySetup.energy_grid: neighbouring threads read
# neighbouring nodes of the node-major copy
cell = np.asarray(w.cset.cell.mat, dtype=np.float64)
na = 256
fr = (torch.arange(na, dtype=torch.float64, device=dev) + 0.37) / na
F = torch.stack(torch.meshgrid(fr, fr, fr, indexing="ij"), dim=-1).reshape(-1, 3)
ptsr = (F @ torch.tensor(cell.T, device=dev)).contiguous()
it.on_device(ptsr.data_ptr(), n, out.data_ptr(), s); torch.cuda.synchronize()
e0.record()
for _ in range(5): it.on_device(ptsr.data_ptr(), n, out.data_ptr(), s)
e1.record(); torch.cuda.synchronize()
msr = e0.elapsed_time(e1) / 5
print(f"GPU interpolate_grid: {n} positions on a regular {na}^3 lattice of the cell (energy_grid order): {msr:.3f} ms -> {n/msr*1e3:.3e} points/s; "
      f"positions in + energies out alone are {n * 32 / (msr * 1e-3) / 1e12:.2f} TB/s, the 270 MB grid is read ~once ({(n * 32 + d_grid.numel() * 4) / (msr * 1e-3) / 8e12:.2f} of the HBM peak)")
hp = ptsr[:m].cpu().numpy()
refr = O.interpolate_points(eg, hp)
gotr = out[:m].cpu().numpy()
blk = refr == 1e100
print("regular lattice: blocked pattern equal:", np.array_equal(gotr == 1e100, blk), " max rel err:", float(np.max(np.abs(gotr[~blk]-refr[~blk])/np.maximum(np.abs(refr[~blk]), 1e-3*np.median(np.abs(refr[~blk]))))))
