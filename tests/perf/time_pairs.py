"""Throughput of the guest-guest pair kernel (row f3): 1000 CO2 molecules (3000 atoms, LJ + CoulombEwaldDirect
pair rules of the fixture force field) in a 40 A cubic MC cell, trial placements of one more CO2; CPU oracle beside it."""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..')]
import numpy as np, torch
import ceg_hip as ceg
from ceg_hip import _abi, grids as G, workloads as W
from ceg_hip.hostmirror import montecarlo as M
from ceg_hip.energy import PairEnergies
from oracle import oracle as O
ceg.setdir_RASPA(os.path.join(here, '..', 'golden', 'raspa'))
ff = ceg.parse_forcefield_RASPA("BoulfelfelSholl2021")
co2 = ceg.load_molecule_RASPA("CO2", "TraPPE", "BoulfelfelSholl2021")
base = np.asarray(co2.position).reshape(-1, 3)
edge = 40.0
nmol = 1000
centers = W._random_atoms_min_sep(nmol, edge, 3.0, np.random.default_rng(0))
mat = np.diag([edge] * 3)
ids = [ff.sdict[a] for a in co2.atomic_symbol]
charges = np.full(len(ff.sdict) + 1, np.nan)
for k, ix in enumerate(ids): charges[ix] = co2.atomic_charge[k]
mc = M.MonteCarloSetup(ff, mat, np.linalg.inv(mat), [ids], charges, [[c + base for c in centers]], ceg.EwaldFramework.empty(mat),
                       G.EnergyGrid.trivial(True), [], 0.0)
pe = PairEnergies(ff, mc.mat, mc.invmat)
pos = np.concatenate(mc.positions[0]); kinds = ids * nmol; mol = np.repeat(np.arange(nmol), 3)
pe.set_atoms(pos, kinds, mol)
lib = _abi.load_library()
dev = torch.device("cuda", 0)
n = 1 << 18
g = torch.Generator(device=dev); g.manual_seed(1)
trial = (torch.rand((n, 1, 3), dtype=torch.float64, device=dev, generator=g) * edge + torch.tensor(base, device=dev)[None]).contiguous()
out = torch.empty(n, dtype=torch.float64, device=dev)
tk = np.ascontiguousarray(np.array(ids, dtype=np.int32) - 1)
s = torch.cuda.current_stream().cuda_stream
def run(): _abi.check(lib, lib.ceg_pairs_energy_device(pe._h, trial.data_ptr(), _abi.i32ptr(tk), 3, n, 0, out.data_ptr(), s))
run(); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"GPU single_contribution_vdw: {n} trial CO2 placements x {len(pos)} guest atoms: {ms:.3f} ms -> {n/ms*1e3:.3e} placements/s "
      f"({n*3*len(pos)/ms*1e3:.3e} pair tests/s)")
# roofline.  Algorithmic FP64 work: every (trial atom, guest atom) pair is distance-tested with the reference's
# unsafe_periodic_distance2! (utils.jl:294-302): 3 subtractions + invmat*d (15) + wrap (9) + mat*f (15) + norm (5) = 47 flops; the
# ~10 % of pairs inside the cutoff add the pair rule (LJ 12 flops, CoulombEwaldDirect ~60 with exp / erfc at nominal 20 / 40).
# Bytes: the guest atoms (32 B each) are read once per workgroup from L2, 72 B per placement in, 8 B out -> compute bound.
npair = n * 3.0 * len(pos)
flops = npair * 47.0 + 0.10 * npair * 72.0
print(f"roofline k_pairs: {flops / n:.0f} algorithmic flops/placement -> {flops / (ms * 1e-3) / 1e12:.2f} TFLOP/s = "
      f"{flops / (ms * 1e-3) / 78.6e12:.3f} of the FP64 vector peak (exhaustive loop; the neighbour-cell path tests fewer pairs); "
      f"HBM {n * 80 / (ms * 1e-3) / 1e9:.1f} GB/s (negligible)")
m = 4000
hp = trial[:m].cpu().numpy()
t = time.perf_counter(); ref = O.single_contribution_vdw(mc, (0, 0), hp); dt = time.perf_counter() - t
got = out[:m].cpu().numpy()
fin = np.isfinite(ref)
print(f"CPU oracle ({O.max_threads()} threads): {m} placements in {dt*1e3:.1f} ms -> {m/dt:.3e} placements/s; "
      f"max rel err GPU vs oracle {float(np.max(np.abs(got[fin]-ref[fin])/np.maximum(np.abs(ref[fin]),1e-6))):.2e}")
