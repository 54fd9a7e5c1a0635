"""Throughput of the batched reciprocal-Ewald kernel (row f2) on the CHA fixture (1368 k-vectors),
Na (1 atom) and CO2 (3 atoms) placements; CPU oracle (literal tables + loop) beside it."""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..')]
import numpy as np, torch
import ceg_hip as ceg
from ceg_hip import _abi
from ceg_hip.energy import ReciprocalEwald
from ceg_hip.hostmirror.ewald import ewald_context_constants
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
    # roofline.  Algorithmic FP64 work per placement: per atom and k-vector the product of three table entries
    # (2 complex multiplications = 12 flops), the charge (2) and the accumulation into the structure factor (2) = 16 flops; per
    # k-vector the two energy terms 2 kf Re(conj(F) S) + kf |S|^2 = 10 flops; per atom the (kx+1) + (2ky+1) + (2kz+1) table entries
    # (sine + cosine, ~40 flops each).  Bytes: 24 B per atom in, 8 B out, tables and k-vector constants live in LDS -> compute bound.
    # (The kernel walks the k-vectors as rows (j, k) x i, so the Ey Ez q product is formed once per row segment and the step along i
    # is one complex product: it executes fewer flops per k-vector than this count, which is kept so that rounds compare.)
    ks = np.asarray(ef.kspace.ks)
    ntab = int(ks[0] + 1 + 2 * ks[1] + 1 + 2 * ks[2] + 1)
    flops = n * (na * nk * 16.0 + nk * 10.0 + na * ntab * 40.0)
    print(f"roofline k_recip {molname}: {flops / n:.0f} algorithmic flops/placement -> {flops / (ms * 1e-3) / 1e12:.2f} TFLOP/s = "
          f"{flops / (ms * 1e-3) / 78.6e12:.3f} of the FP64 vector peak; HBM {n * (24 * na + 8) / (ms * 1e-3) / 1e9:.1f} GB/s (negligible) -> "
          f"FP64 VALU issue (scripts/pmc_kernel.sh: issue utilisation 0.79 / 0.90 for Na / CO2, two thirds of the VALU instructions FP64)")
    m = 20000
    hp = pos[:m].cpu().numpy()
    t = time.perf_counter(); ref = O.reciprocal_energies(ef, mol, hp); dt = time.perf_counter() - t
    print(f"CPU oracle ({O.max_threads()} threads): {m} placements in {dt*1e3:.1f} ms -> {m/dt:.3e} placements/s; "
          f"max rel err GPU vs oracle {float(np.max(np.abs(out[:m].cpu().numpy()-ref)/np.abs(ref))):.2e}")
