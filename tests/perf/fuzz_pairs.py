"""Randomized parity run of the guest-guest pair kernels (row f3 and the pairs term of config 5; not part of the suite): random cells
(upper-triangular as the reference builds them, or rotated: the general-matrix variants), every perpendicular width above or below two
cutoffs (fast / literal wrap), 1 - 6000 guest atoms of the fixture force field's kinds incl. dense clusters, rigid molecules of 1 - 16
atoms, placements inside, outside and on the faces of the cell, some exactly at cutoff distance from an atom, neighbour cells forced
on / off / by size, excluded molecule -- ceg_pairs_energy (k_pairs_frac / k_pairs) against oracle_single_contribution_vdw at 1e-9,
and the Cartesian kernel (CEG_HIP_PAIRS_FRAC=0) on every third configuration.
usage: fuzz_pairs.py [nconfigs] [seed]"""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..'), os.path.join(here, '..')]
import numpy as np
import ceg_hip as ceg
from ceg_hip import _abi
from ceg_hip.hostmirror.constants import COULOMBIC_CONVERSION_FACTOR
from ceg_hip.hostmirror.utils import mat_from_parameters
from oracle import oracle as O
from test_gpu_consumers import _pairs_gpu, _rotation

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ceg.setdir_RASPA(os.path.join(here, '..', 'golden', 'raspa'))
ff = ceg.parse_forcefield_RASPA("BoulfelfelSholl2021")
rules, offsets = ff.pair_table()
lib = _abi.load_library()
oracle = O.load() if hasattr(O, "load") else O
guest_kinds = [ff.sdict[a] - 1 for a in ("C_co2", "O_co2", "Na", "C_ch4", "H_ch4", "Ar", "N_n2", "O_o2")]
done = fails = 0
stats = {"fast": 0, "literal": 0, "general": 0, "cells_on": 0, "cutoff_hits": 0, "m>4": 0, "max_atoms": 0}
t0 = time.time()
while done < n_cfg:
    cutoff = ff.cutoff
    small = rng.random() < 0.15                                       # a cell below two cutoffs in some direction: the literal wrap
    lengths = rng.uniform(14.0, 30.0, 3) if small else rng.uniform(26.0, 90.0, 3)
    angles = rng.uniform(70.0, 110.0, 3) if rng.random() < 0.7 else np.array([90.0, 90.0, 90.0])
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
    widths = 1.0 / np.linalg.norm(inv, axis=1)
    fast = bool(np.all(cutoff / widths < 0.5 * (1 - 1e-6)))
    natoms = int(rng.choice([1, 2, 63, 64, 65, 300, 1500, 6000]))
    pos = rng.uniform(-0.2, 1.2, (natoms, 3)) @ mat.T
    if natoms >= 300 and rng.random() < 0.5:                             # a dense cluster: whole blocks inside the cutoff, the queue fills
        c = rng.uniform(0.2, 0.8, 3) @ mat.T
        k = min(natoms // 2, 200)
        pos[:k] = c + rng.uniform(-3.5, 3.5, (k, 3))
    kinds = rng.choice(guest_kinds, natoms).astype(np.int32)
    mol = (np.arange(natoms) // int(rng.integers(1, 5))).astype(np.int32)
    m = int(rng.choice([1, 2, 3, 4, 5, 7, 12, 16]))
    tk = rng.choice(guest_kinds, m).astype(np.int32)
    tbase = rng.uniform(-1.8, 1.8, (m, 3))
    n = int(rng.choice([1, 5, 64, 700]))
    trial = (rng.uniform(-0.6, 1.6, (n, 3)) @ mat.T)[:, None, :] + tbase[None]
    # some placements with their first atom exactly at cutoff distance (to rounding) from a guest atom, and some on a cell face
    nhit = min(n // 3, natoms)
    if nhit:
        u = rng.normal(size=(nhit, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
        d = rng.choice([0.0, 1e-16, -1e-16, 1e-13, -1e-13, 1e-10, -1e-10], nhit)
        first = pos[rng.integers(0, natoms, nhit)] + cutoff * (1.0 + d)[:, None] * u
        trial[:nhit] = first[:, None, :] + (tbase - tbase[0])[None]
        stats["cutoff_hits"] += nhit
    if n >= 64:
        f = rng.uniform(0, 1, (8, 3)); f[:, rng.integers(0, 3)] = rng.choice([0.0, 1.0, 1e-9, 1 - 1e-9])
        trial[-8:] = (f @ mat.T)[:, None, :] + tbase[None]
    exclude = int(rng.integers(-1, mol.max() + 1))
    env = {}
    c = rng.random()
    if c < 0.3: env["CEG_HIP_MC_CELLS"] = "1"; env["CEG_HIP_MC_BIN"] = f"{rng.uniform(2.0, 6.0):.2f}"
    elif c < 0.6: env["CEG_HIP_MC_CELLS"] = "0"
    for k_ in ("CEG_HIP_MC_CELLS", "CEG_HIP_MC_BIN", "CEG_HIP_PAIRS_FRAC"):
        os.environ.pop(k_, None)
    os.environ.update(env)
    what = f"cfg{done} seed{seed}: L {np.round(lengths, 1)} A {np.round(angles, 1)} general {general} fast {fast} atoms {natoms} m {m} n {n} exclude {exclude} env {env}"
    args = (mat, cutoff ** 2, rules, offsets, ff.nkinds, COULOMBIC_CONVERSION_FACTOR, pos, kinds, mol, trial, tk, exclude)
    try:
        ref = oracle.single_contribution_vdw_raw(mat, inv, *args[1:])
        variants = [("default", {})] + ([("Cartesian", {"CEG_HIP_PAIRS_FRAC": "0"})] if done % 3 == 0 else [])
        for name, extra in variants:
            os.environ.update(extra)
            got = _pairs_gpu(lib, *args)
            for k_ in extra:
                os.environ.pop(k_, None)
            assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{name}: NaN pattern"
            inf = np.isinf(ref)
            assert np.array_equal(np.isinf(got), inf) and np.array_equal(got[inf], ref[inf]), f"{name}: Inf pattern"
            fin = np.isfinite(ref)
            if fin.any():
                scale = float(np.percentile(np.abs(ref[fin]), 75))
                err = np.abs(got[fin] - ref[fin])
                tol = 1e-9 * np.abs(ref[fin]) + 1e-12 * scale + 1e-9
                assert (err <= tol).all(), f"{name}: worst {float(np.max(err / tol)):.2f} x tolerance, |ref| scale {scale:.3e}"
    except AssertionError as e:
        fails += 1
        print("FAIL", what, "::", str(e)[:300], flush=True)
    stats["fast" if fast else "literal"] += 1
    stats["general"] += int(general)
    stats["cells_on"] += int(env.get("CEG_HIP_MC_CELLS") == "1")
    stats["m>4"] += int(m > 4)
    stats["max_atoms"] = max(stats["max_atoms"], natoms)
    done += 1
    if done % 50 == 0:
        print(f"{done} configs, {fails} failures, {time.time() - t0:.0f} s, {stats}", flush=True)
print(f"done: {done} configs, {fails} failures, {stats}")
sys.exit(1 if fails else 0)
