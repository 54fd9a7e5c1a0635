"""Throughput of the batched reciprocal-Ewald kernel (row f2) on the CHA fixture (1368 k-vectors),
Na (1 atom) and CO2 (3 atoms) placements; CPU oracle (literal tables + loop) beside it."""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..')]
import numpy as np, torch
import ceg_hip as ceg
from ceg_hip import _abi
from ceg_hip.energy import ReciprocalEwald
from ceg_hip.ewald import ewald_context_constants
from oracle import oracle as O
ceg.setdir_RASPA(os.path.join(here, '..', 'golden', 'raspa'))
fw = ceg.load_framework_RASPA("CHA_1.4_3b4eeb96", "BoulfelfelSholl2021")
ef = ceg.initialize_ewald(fw, (1, 1, 1))
rec = ReciprocalEwald(ef)
lib = _abi.load_library()
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
for molname in ("Na", "CO2"):
    mol = ceg.load_molecule_RASPA(molname, "TraPPE", "BoulfelfelSholl2021")
    base = torch.tensor(np.asarray(mol.position, dtype=np.float64).reshape(-1, 3), device=dev)
    na = len(base)
    n = 1 << 20
    g = torch.Generator(device=dev); g.manual_seed(1)
    pos = (torch.rand((n, 1, 3), dtype=torch.float64, device=dev, generator=g) * 60.0 - 15.0 + base[None]).contiguous()
    out = torch.empty(n, dtype=torch.float64, device=dev)
    q = np.ascontiguousarray(mol.atomic_charge, dtype=np.float64)
    enc, static = ewald_context_constants(ef, ((mol,),))
    def run():
        _abi.check(lib, lib.ceg_recip_energy_device(rec._h, pos.data_ptr(), _abi.dptr(q), na, n, enc, static, out.data_ptr(), s))
    run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    nk = len(ef.kfactors)
    print(f"GPU compute_ewald {molname}: {n} placements x {nk} k-vectors x {na} atoms: {ms:.3f} ms -> {n/ms*1e3:.3e} placements/s "
          f"({n*nk*na/ms*1e3:.3e} atom-kvec/s)")
    m = 20000
    hp = pos[:m].cpu().numpy()
    t = time.perf_counter(); ref = O.reciprocal_energies(ef, mol, hp); dt = time.perf_counter() - t
    print(f"CPU oracle ({O.max_threads()} threads): {m} placements in {dt*1e3:.1f} ms -> {m/dt:.3e} placements/s; "
          f"max rel err GPU vs oracle {float(np.max(np.abs(out[:m].cpu().numpy()-ref)/np.abs(ref))):.2e}")
