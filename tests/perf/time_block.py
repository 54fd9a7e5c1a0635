"""Row f4 timings: BlockFile(g) mask on a device-resident 256^3 value channel, and the parse_blockfile scan
(5 spheres) on the 218x204x190 CHA lattice; CPU oracle beside them.  Kernel times via HIP events on
device-resident data (ceg_block_* itself also uploads / downloads)."""
import os, sys, time, ctypes as C
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..')]
import numpy as np, torch
import ceg_hip as ceg
from ceg_hip import _abi, grids as G, workloads as W
from oracle import oracle as O
ceg.setdir_RASPA(os.path.join(here, "..", "golden", "raspa"))
lib = _abi.load_library()
dev = torch.device("cuda", 0)
# --- BlockFile(g)
n = 255
dims = np.array([n, n, n], dtype=np.int32)
v = (torch.rand((n + 1,) * 3, device=dev) * 5.2e6).float().contiguous()
out = np.empty((n + 1,) * 3, dtype=np.uint8)
for _ in range(2):
    t = time.perf_counter()
    _abi.check(lib, lib.ceg_block_from_grid(0, v.data_ptr(), 1, _abi.i32ptr(dims), 5e6, out.ctypes.data))
    dt = time.perf_counter() - t
print(f"ceg_block_from_grid 256^3 (value on device, mask to host): {dt*1e3:.2f} ms end to end, {out.mean()*100:.1f} % blocked")
import tempfile, math
cset = ceg.GridCoordinatesSetup.from_cell(ceg.load_framework_RASPA("CHA_1.4_3b4eeb96", "BoulfelfelSholl2021").mat, 0.15)
g = G.EnergyGrid(cset, (1, 1, 1), math.inf, True, None)
hv = v.cpu().numpy()
t = time.perf_counter(); ref = O.block_from_grid(G.EnergyGrid(W.grid_setup_with_dims(np.diag([40.0] * 3), (n, n, n)), (1, 1, 1), math.inf, True, hv[None])); dt = time.perf_counter() - t
print(f"CPU oracle BlockFile(g) 256^3 (1 thread, like the reference): {dt*1e3:.1f} ms; equal: {np.array_equal(ref, out.astype(bool))}")
# --- parse_blockfile scan
with tempfile.NamedTemporaryFile("w", suffix=".block", delete=False) as f:
    f.write("5\n0.05 0.5 0.95 2.5\n0.5 0.5 0.5 4.0\n0.99 0.01 0.5 1.2\n0.3 0.7 0.1 0.9\n0.0 0.0 0.0 3.3\n")
for _ in range(2):
    t = time.perf_counter(); got = G.parse_blockfile_gpu(f.name, cset); dt = time.perf_counter() - t
npts = int(np.prod(cset.npoints))
print(f"parse_blockfile_gpu {cset.npoints} = {npts} points x 5 spheres: {dt*1e3:.2f} ms end to end")
centers, r2 = G.read_block_spheres(f.name, cset)
t = time.perf_counter(); ref = O.block_spheres(cset, centers, r2); dt = time.perf_counter() - t
print(f"CPU oracle scan ({O.max_threads()} threads): {dt*1e3:.1f} ms; equal: {np.array_equal(ref, got.block)}")
