"""GPU parity of the consumer kernels at the scale BASELINE config 5 runs them:
thousands of guest atoms, 10^4+ trial placements, every code path of the pair kernel (multi-pass atom
loop, queue flushes, pair table in LDS and in global memory, ceg_math.h and libm-grade rule arithmetic).
Oracle: oracle_single_contribution_vdw (energy.jl:397-427).  Run with `pytest -m gpu` on an MI355X."""
import ctypes as C

import numpy as np
import pytest

import ceg_hip as ceg
from ceg_hip import _abi, workloads as W
from ceg_hip.constants import COULOMBIC_CONVERSION_FACTOR

pytestmark = pytest.mark.gpu


def _pairs_gpu(lib, mat, cutoff2, rules, offsets, nkinds, coulombic, pos, kinds, mol, trial, tk, exclude):
    """ceg_pairs_create / set_atoms / energy through the C ABI on explicit tables (0-based kinds)."""
    matT = np.ascontiguousarray(np.asarray(mat, dtype=np.float64).T.reshape(9))
    invT = np.ascontiguousarray(np.linalg.inv(np.asarray(mat, dtype=np.float64)).T.reshape(9))
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    h = C.c_void_p()
    _abi.check(lib, lib.ceg_pairs_create(C.byref(h), 0, _abi.dptr(matT), _abi.dptr(invT), float(cutoff2), rules.ctypes.data,
                                         _abi.i32ptr(offsets), int(nkinds), float(coulombic)))
    try:
        p = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        k = np.ascontiguousarray(kinds, dtype=np.int32)
        m = np.ascontiguousarray(mol, dtype=np.int32)
        _abi.check(lib, lib.ceg_pairs_set_atoms(h, _abi.dptr(p.reshape(-1)), _abi.i32ptr(k), _abi.i32ptr(m), len(p)))
        tkk = np.ascontiguousarray(tk, dtype=np.int32)
        t = np.ascontiguousarray(trial, dtype=np.float64).reshape(-1, len(tkk), 3)
        out = np.empty(len(t), dtype=np.float64)
        _abi.check(lib, lib.ceg_pairs_energy(h, _abi.dptr(t.reshape(-1)), _abi.i32ptr(tkk), len(tkk), len(t), int(exclude),
                                             _abi.dptr(out)))
        return out
    finally:
        lib.ceg_pairs_destroy(h)


def _assert_energies(got, ref, what, rtol=1e-9):
    assert got.shape == ref.shape
    assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{what}: NaN pattern differs"
    assert np.array_equal(np.isinf(got), np.isinf(ref)) and np.array_equal(got[np.isinf(ref)], ref[np.isinf(ref)]), f"{what}: Inf pattern"
    fin = np.isfinite(ref)
    scale = float(np.percentile(np.abs(ref[fin]), 75))
    err = np.abs(got[fin] - ref[fin])
    tol = rtol * np.abs(ref[fin]) + rtol * 1e-3 * scale
    assert (err <= tol).all(), f"{what}: worst {float(np.max(err / np.maximum(np.abs(ref[fin]), 1e-3 * scale))):.3e}"
    return float(np.max(err / np.maximum(np.abs(ref[fin]), 1e-3 * scale)))


def _config5_guests(ff, rng):
    """1000 CO2 + 7 Na (3007 atoms: not a multiple of 64) in a skewed 40 A MC cell, plus a 150-atom cluster
    of oxygens 0.9 A apart (a whole 64-lane pass inside the cutoff of a trial atom: the queue must flush)."""
    co2 = ceg.load_molecule_RASPA("CO2", "TraPPE", "BoulfelfelSholl2021")
    base = np.asarray(co2.position, dtype=np.float64).reshape(-1, 3)
    ids = [ff.sdict[a] - 1 for a in co2.atomic_symbol]
    edge = 40.0
    mat = np.array([[edge, 0, 0], [3.0, edge, 0], [-2.0, 4.0, edge]]).T
    centers = W._random_atoms_min_sep(1000, edge, 3.0, rng)
    pos, kinds, mol = [], [], []
    for m, c in enumerate(centers):
        pos.append(c + base); kinds += ids; mol += [m] * 3
    na = W._random_atoms_min_sep(7, edge, 5.0, rng)
    for q, c in enumerate(na):
        pos.append(c[None]); kinds.append(ff.sdict["Na"] - 1); mol.append(1000 + q)
    cluster = W._random_atoms_min_sep(150, 6.0, 0.9, rng) + np.array([17.0, 21.0, 9.0])
    pos.append(cluster); kinds += [ff.sdict["O_co2"] - 1] * 150; mol += list(range(1007, 1157))
    return mat, np.concatenate(pos), np.array(kinds, dtype=np.int32), np.array(mol, dtype=np.int32), base, ids


@pytest.mark.parametrize("trial_mol", ["CO2", "CH4-like"])
def test_pairs_config5_scale(hip_lib, oracle, forcefield, trial_mol):
    """Row f3 at config-5 scale: 3157 guest atoms, 16384 placements of a 3- or 5-atom molecule, the excluded
    molecule in the middle of the atom list, a dense cluster, fixture force field (LJ + CoulombEwaldDirect
    pair rules, ceg_math.h arithmetic, pair table in LDS)."""
    ff = forcefield
    rng = np.random.default_rng(5)
    mat, pos, kinds, mol, base, ids = _config5_guests(ff, rng)
    assert len(pos) == 3157 and len(pos) % 64 != 0
    if trial_mol == "CO2":
        tbase, tk = base, ids
    else:          # tetrahedral 5-atom molecule with the fixture's methane kinds
        t = 1.09 / np.sqrt(3.0)
        tbase = np.array([[0, 0, 0], [t, t, t], [t, -t, -t], [-t, t, -t], [-t, -t, t]], dtype=np.float64)
        tk = [ff.sdict["C_ch4"] - 1] + [ff.sdict["H_ch4"] - 1] * 4
    n = 16384
    frac = rng.uniform(0, 1, (n, 3))
    trial = (frac @ mat.T)[:, None, :] + tbase[None]
    trial[: n // 8] = (rng.uniform(0, 6.0, (n // 8, 3)) + np.array([17.0, 21.0, 9.0]))[:, None, :] + tbase[None]   # into the cluster
    trial[n // 8: n // 4] += 3.0 * (mat[:, 0] - mat[:, 1])[None, None, :]                                       # far outside the cell
    rules, offsets = ff.pair_table()
    args = (mat, ff.cutoff ** 2, rules, offsets, ff.nkinds, COULOMBIC_CONVERSION_FACTOR, pos, kinds, mol, trial, tk, 500)
    got = _pairs_gpu(hip_lib, *args)
    ref = oracle.single_contribution_vdw_raw(args[0], np.linalg.inv(mat), *args[1:])
    assert np.isfinite(ref).sum() > n // 2 and np.abs(ref[np.isfinite(ref)]).max() > 1e3
    _assert_energies(got, ref, f"pairs {trial_mol}")
    # the excluded molecule matters: with nothing excluded the energies differ where molecule 500 is in range
    got_all = _pairs_gpu(hip_lib, *args[:-1], -1)
    assert (got_all != got).any()
    _assert_energies(got_all, oracle.single_contribution_vdw_raw(args[0], np.linalg.inv(mat), *args[1:-1], -1), "pairs, none excluded")


def _synthetic_table(nkinds, alpha, rng, cutoff):
    """Every pair: shifted LJ + CoulombEwaldDirect(alpha); a quarter of them Buckingham instead of LJ."""
    from ceg_hip.interactions import FF
    rules = np.zeros(2 * nkinds * nkinds, dtype=_abi.RULE_DTYPE)
    offsets = np.arange(0, 2 * nkinds * nkinds + 1, 2, dtype=np.int32)
    q = rng.uniform(-1.0, 1.0, nkinds)
    eps = rng.uniform(20.0, 150.0, nkinds); sig = rng.uniform(2.5, 3.6, nkinds)
    for a in range(nkinds):
        for b in range(nkinds):
            t = 2 * (a * nkinds + b)
            e, s = np.sqrt(eps[a] * eps[b]), 0.5 * (sig[a] + sig[b])
            if (a + b) % 4 == 3:
                rules[t]["kind"] = int(FF.Buckingham); rules[t]["p"] = (5.0e6, 3.6, 4.0e4)
            else:
                x6 = (s / cutoff) ** 6
                rules[t]["kind"] = int(FF.LennardJones); rules[t]["p"] = (e, s, 0.0); rules[t]["shift"] = 4 * e * x6 * (x6 - 1)
            rules[t + 1]["kind"] = int(FF.CoulombEwaldDirect); rules[t + 1]["p"] = (alpha, q[a], q[b])
    return rules, offsets


@pytest.mark.parametrize("nkinds,alpha,what", [(36, 0.265, "table in global memory"), (6, 0.5, "libm-grade arithmetic"),
                                              (36, 0.5, "global table + libm-grade")])
def test_pairs_kernel_variants(hip_lib, oracle, nkinds, alpha, what):
    """The k_pairs template variants the fixture force field does not reach: a pair table beyond the 48 KB LDS budget
    (36^2 pairs x 2 rules x 40 B = 104 KB) and alpha*cutoff > 5 (outside the erfcx polynomial's domain -> libm-grade rule
    arithmetic), each on 1000+ atoms and several workgroups, against the oracle."""
    rng = np.random.default_rng(11)
    cutoff = 12.0
    rules, offsets = _synthetic_table(nkinds, alpha, rng, cutoff)
    table_bytes = rules.nbytes + offsets.nbytes
    assert (table_bytes > 48 * 1024) == (nkinds == 36)
    edge = 31.0
    mat = np.array([[edge, 0, 0], [1.5, edge, 0], [0.5, -2.5, edge]]).T
    natoms = 1301
    pos = W._random_atoms_min_sep(natoms, edge, 1.9, rng)
    kinds = rng.integers(0, nkinds, natoms).astype(np.int32)
    mol = (np.arange(natoms) // 3).astype(np.int32)
    m = 4
    tk = rng.integers(0, nkinds, m).astype(np.int32)
    tbase = rng.uniform(-1.0, 1.0, (m, 3))
    n = 2048
    trial = rng.uniform(0, edge, (n, 1, 3)) + tbase[None]
    args = (mat, cutoff ** 2, rules, offsets, nkinds, COULOMBIC_CONVERSION_FACTOR, pos, kinds, mol, trial, tk, 217)
    got = _pairs_gpu(hip_lib, *args)
    ref = oracle.single_contribution_vdw_raw(args[0], np.linalg.inv(mat), *args[1:])
    assert np.isfinite(ref).all()
    _assert_energies(got, ref, what)
