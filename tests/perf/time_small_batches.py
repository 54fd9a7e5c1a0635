"""Latency of the host-array entry points of the grid consumers (rows f1-f3) at small batch sizes: ceg_interp_points, ceg_recip_energy,
ceg_pairs_energy with 1, 64 and 4096 rows (CHA fixture grid at 0.3 A; CO2; 3000 guest atoms).
    python tests/perf/time_small_batches.py"""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..'), os.path.join(here, '..')]
import numpy as np
import ceg_hip as ceg
from ceg_hip import workloads as W
from ceg_hip.energy import PairEnergies, ReciprocalEwald
from ceg_hip.interp import GridInterpolator
from ceg_hip.plan import GridPlan
from ceg_hip import grids as G
import torch

ceg.setdir_RASPA(os.path.join(here, '..', 'golden', 'raspa'))
ff = ceg.parse_forcefield_RASPA("BoulfelfelSholl2021")
rng = np.random.default_rng(0)

def timed(fn, reps):
    fn(); fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps * 1e6

# f1: a VdW grid of the CHA fixture
w = W.fixture_workload("CHA_1.4_3b4eeb96", "Ar", 0.3)
nx, ny, nz = w.cset.npoints
buf = torch.empty((8, nx, ny, nz), dtype=torch.float32, device="cuda")
p = GridPlan(w.cset, w.probe_vdw, None, 0.0)
p.build_vdw(buf.data_ptr(), nx * ny * nz, 0, nx)
torch.cuda.synchronize()
p.close()
import math
eg = G.EnergyGrid(w.cset, (1, 1, 1), math.inf, True, np.ascontiguousarray(buf.cpu().numpy()))
lines = []
gi = GridInterpolator(eg)
for n in (1, 64, 4096):
    pts = rng.uniform(0, 25.0, (n, 3))
    lines.append(f"ceg_interp_points  {n:5d} rows: {timed(lambda: gi(pts), 300):8.1f} us per call")
gi.close()
# f3: 3000 guest atoms in a 40 A cube
co2 = ceg.load_molecule_RASPA("CO2", "TraPPE", "BoulfelfelSholl2021")
base = np.asarray(co2.position).reshape(-1, 3)
ids = [ff.sdict[a] for a in co2.atomic_symbol]
centers = W._random_atoms_min_sep(1000, 40.0, 3.0, rng)
matc = np.diag([40.0] * 3)
pe = PairEnergies(ff, matc, np.linalg.inv(matc))
pe.set_atoms(np.concatenate([c + base for c in centers]), ids * 1000, np.repeat(np.arange(1000), 3))
for n in (1, 64, 4096):
    trial = rng.uniform(0, 40.0, (n, 1, 3)) + base[None]
    lines.append(f"ceg_pairs_energy   {n:5d} rows: {timed(lambda: pe.energies(trial, ids, 0), 300):8.1f} us per call")
pe.close()
print("\n".join(lines))
