"""Randomized run of the device-resident MC state, guest-guest term (config 5; not part of the suite): random cells (upper-triangular or
rotated, neighbour cells on / off / by size), 20 - 2500 guest atoms in molecules of 1 - 6 atoms, random sequences of accepted
displacements, insertions and removals with a host copy of the state beside them; after every few updates a small batch (workgroup kernel)
and a batch of 1100 - 2500 rows (wave kernels: k_mcw_pairs_frac or, when the cell is too small for the fast wrap, k_mcw_pairs) against
oracle_single_contribution_vdw at 1e-9.
usage: fuzz_mc_pairs.py [nconfigs] [seed]"""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..'), os.path.join(here, '..')]
import numpy as np
import ceg_hip as ceg
from ceg_hip import _abi
from ceg_hip.hostmirror.constants import COULOMBIC_CONVERSION_FACTOR
from ceg_hip.hostmirror.utils import mat_from_parameters
from oracle import oracle as O
from test_gpu_consumers import _RawMc, _rotation

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ceg.setdir_RASPA(os.path.join(here, '..', 'golden', 'raspa'))
ff = ceg.parse_forcefield_RASPA("BoulfelfelSholl2021")
rules, offsets = ff.pair_table()
lib = _abi.load_library()
pool = [ff.sdict[a] - 1 for a in ("C_co2", "O_co2", "Na", "C_ch4", "H_ch4", "Ar", "N_n2")]
done = fails = checks = 0
stats = {"general": 0, "cells_on": 0, "small_cell": 0, "wave_batches": 0, "max_atoms": 0}
t0 = time.time()

def check(got, ref, what):
    global fails
    ok = np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref)) and np.array_equal(got[np.isinf(ref)], ref[np.isinf(ref)])
    fin = np.isfinite(ref)
    if ok and fin.any():
        scale = float(np.percentile(np.abs(ref[fin]), 75))
        ok = bool(np.all(np.abs(got[fin] - ref[fin]) <= 1e-9 * np.abs(ref[fin]) + 1e-12 * scale + 1e-9))
    if not ok:
        fails += 1
        print("FAIL", what, flush=True)
    return ok

while done < n_cfg:
    small = rng.random() < 0.15
    lengths = rng.uniform(14.0, 26.0, 3) if small else rng.uniform(26.0, 80.0, 3)
    angles = rng.uniform(72.0, 108.0, 3) if rng.random() < 0.7 else np.array([90.0, 90.0, 90.0])
    try:
        mat = np.array(mat_from_parameters(tuple(lengths), tuple(angles)))
    except Exception:
        continue
    if not np.all(np.isfinite(mat)) or np.linalg.det(mat) <= 500.0:
        continue
    general = rng.random() < 0.35
    if general:
        mat = _rotation(rng) @ mat
    inv = np.linalg.inv(mat)
    for k_ in ("CEG_HIP_MC_CELLS", "CEG_HIP_MC_BIN"):
        os.environ.pop(k_, None)
    c = rng.random()
    if c < 0.3: os.environ["CEG_HIP_MC_CELLS"] = "1"; os.environ["CEG_HIP_MC_BIN"] = f"{rng.uniform(2.5, 6.0):.2f}"
    elif c < 0.6: os.environ["CEG_HIP_MC_CELLS"] = "0"
    nmol = int(rng.choice([8, 40, 200, 700]))
    mols = []
    for _ in range(nmol):
        m = int(rng.choice([1, 1, 2, 3, 3, 5, 6]))
        ks = list(rng.choice(pool, m))
        shape = rng.uniform(-1.5, 1.5, (m, 3)); shape[0] = 0.0
        mols.append((ks, (rng.uniform(-0.2, 1.2, 3) @ mat.T)[None] + shape))
    h = _RawMc(lib, mat, ff.cutoff ** 2, rules, offsets, ff.nkinds, COULOMBIC_CONVERSION_FACTOR)
    what0 = f"cfg{done} seed{seed}: L {np.round(lengths, 1)} general {general} env {dict((k, os.environ[k]) for k in ('CEG_HIP_MC_CELLS', 'CEG_HIP_MC_BIN') if k in os.environ)}"

    def flat():
        pos = np.concatenate([p for _k, p in mols])
        kinds = np.array([k for ks, _p in mols for k in ks], dtype=np.int32)
        first = np.concatenate([[0], np.cumsum([len(ks) for ks, _p in mols])]).astype(np.int32)
        mol = np.repeat(np.arange(len(mols)), [len(ks) for ks, _p in mols]).astype(np.int32)
        return pos, kinds, first, mol
    try:
        pos, kinds, first, mol = flat()
        h.set_guests(pos, kinds, first)
        stats["cells_on"] += int(h.cells() is not None)
        for step in range(24):
            op = rng.integers(0, 5)
            if op <= 2 and mols:                                         # accepted displacement
                j = int(rng.integers(len(mols)))
                ks, cur = mols[j]
                new = cur + rng.uniform(-1.0, 1.0, 3) if rng.random() < 0.7 else cur - cur[0] + (rng.uniform(-0.5, 1.5, 3) @ mat.T)
                h.accept(j, new)
                mols[j] = (ks, new.copy())
            elif op == 3:                                                # insertion
                m = int(rng.choice([1, 2, 3, 5]))
                ks = list(rng.choice(pool, m))
                p = (rng.uniform(0, 1, 3) @ mat.T)[None] + rng.uniform(-1.5, 1.5, (m, 3))
                assert h.insert(ks, p) == len(mols)
                mols.append((ks, p.copy()))
            elif len(mols) > 2:                                          # removal: the last molecule takes the index
                j = int(rng.integers(len(mols)))
                assert h.remove(j) == len(mols) - 1
                mols[j] = mols[-1]
                mols.pop()
            if step % 6 == 5:
                pos, kinds, first, mol = flat()
                j = int(rng.integers(len(mols)))
                ks, cur = mols[j]
                for n in (6, int(rng.integers(1100, 2500))):
                    trial = cur[None] + rng.uniform(-0.8, 0.8, (n, 1, 3))
                    trial[n // 2:] = (cur - cur[0])[None] + (rng.uniform(-0.3, 1.3, (n - n // 2, 3)) @ mat.T)[:, None, :]
                    got = h.trial(j, trial)
                    ref = O.single_contribution_vdw_raw(mat, inv, ff.cutoff ** 2, rules, offsets, ff.nkinds, COULOMBIC_CONVERSION_FACTOR, pos, kinds, mol,
                                                        np.concatenate([cur[None], trial]), ks, j)
                    check(got, ref, f"{what0} step {step} displacement batch {n} of molecule {j} ({len(ks)} atoms), {len(pos)} atoms")
                    checks += 1
                    stats["wave_batches"] += int(n >= 1024)
                m = int(rng.choice([1, 3, 4, 6]))
                ks = list(rng.choice(pool, m))
                shape = rng.uniform(-1.5, 1.5, (m, 3))
                n = int(rng.integers(1100, 2000))
                trial = (rng.uniform(-0.2, 1.2, (n, 3)) @ mat.T)[:, None, :] + shape[None]
                got = h.trial_insert(ks, trial)
                ref = O.single_contribution_vdw_raw(mat, inv, ff.cutoff ** 2, rules, offsets, ff.nkinds, COULOMBIC_CONVERSION_FACTOR, pos, kinds, mol, trial, ks, -1)
                check(got, ref, f"{what0} step {step} insertion batch {n} ({m} atoms), {len(pos)} atoms")
                checks += 1
                stats["wave_batches"] += 1
        stats["max_atoms"] = max(stats["max_atoms"], len(flat()[0]))
    finally:
        h.close()
    stats["general"] += int(general)
    stats["small_cell"] += int(small)
    done += 1
    if done % 10 == 0:
        print(f"{done} configs, {checks} batches checked, {fails} failures, {time.time() - t0:.0f} s, {stats}", flush=True)
print(f"done: {done} configs, {checks} batches checked, {fails} failures, {stats}")
sys.exit(1 if fails else 0)
