"""Randomized parity run of the reciprocal-space kernel (row f2, not part of the suite): random triclinic cells, Ewald precisions
(k-space boxes from a few dozen to ~20 000 k-vectors: constants in LDS and in global memory), structure factors, rigid molecules of
1 - 16 atoms with random charges, placements far outside the cell -- ceg_recip_energy against the literal oracle.
usage: fuzz_recip.py [nconfigs] [seed]"""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..'), os.path.join(here, '..')]
import numpy as np
import ceg_hip as ceg
from ceg_hip.energy import ReciprocalEwald
from ceg_hip.hostmirror.raspa import RASPASystem
from ceg_hip.hostmirror.utils import mat_from_parameters
from oracle import oracle as O

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
done = fails = 0
stats = {"in_lds": 0, "global": 0, "max_nk": 0, "atoms16": 0}
t0 = time.time()
while done < n_cfg:
    lengths = rng.uniform(9.0, 45.0, 3)
    angles = rng.uniform(62.0, 118.0, 3) if rng.random() < 0.7 else np.array([90.0, 90.0, 90.0])
    try:
        mat = mat_from_parameters(tuple(lengths), tuple(angles))
    except Exception:
        continue
    if not np.all(np.isfinite(mat)) or np.linalg.det(mat) <= 50.0:
        continue
    precision = float(10.0 ** rng.uniform(-9.0, -3.0))
    ef = ceg.initialize_ewald(np.array(mat), (1, 1, 1), precision)
    nk = len(ef.kfactors)
    ks = ef.kspace.ks
    if nk == 0 or ks[0] + 1 + 2 * ks[1] + 1 + 2 * ks[2] + 1 > 400 or nk > 40000:
        continue
    ef.StoreRigidChargeFramework = (rng.normal(0, 30, nk) + 1j * rng.normal(0, 30, nk)) * (rng.random(nk) < 0.9)
    na = int(rng.choice([1, 2, 3, 5, 8, 16]))
    tab = 16 * na * (ks[0] + 1 + 2 * ks[1] + 1 + 2 * ks[2] + 1)
    if tab > 64 * 1024:
        continue
    q = rng.uniform(-1.2, 1.2, na)
    base = rng.uniform(-2.5, 2.5, (na, 3))
    mol = RASPASystem(np.array(mat), base, ["X"] * na, np.ones(na), q, True)
    n = int(rng.choice([1, 7, 64, 257]))
    pos = rng.uniform(-120, 160, (n, 1, 3)) + base[None]
    what = f"cfg{done} seed{seed}: L {np.round(lengths, 2)} A {np.round(angles, 1)} precision {precision:.1e} ks {ks} nk {nk} atoms {na} n {n}"
    try:
        rec = ReciprocalEwald(ef)
        got = rec.energies(mol, pos)
        rec.close()
        ref = O.reciprocal_energies(ef, mol, pos)
        assert np.all(np.isfinite(got)), "non-finite"
        err = np.abs(got - ref)
        assert np.all(err <= 1e-10 * np.abs(ref) + 1e-11 * np.abs(ref).max() + 1e-9), f"max err {err.max():.3e} at |ref| {np.abs(ref).max():.3e}"
    except AssertionError as e:
        fails += 1
        print("FAIL", what, "::", str(e)[:300], flush=True)
    stats["max_nk"] = max(stats["max_nk"], nk)
    stats["atoms16"] += int(na == 16)
    stats["in_lds" if nk * 1.15 * 24 + tab / 16 * 16 * min(8, max(1, 40960 // max(tab, 1))) < 60000 else "global"] += 1
    done += 1
    if done % 50 == 0:
        print(f"{done} configs, {fails} failures, {time.time() - t0:.0f} s, {stats}", flush=True)
print(f"done: {done} configs, {fails} failures, {stats}")
sys.exit(1 if fails else 0)
