"""GPU parity of the consumer kernels at the scale BASELINE config 5 runs them:
thousands of guest atoms, 10^4+ trial placements, every code path of the pair kernel (multi-pass atom
loop, queue flushes, pair table in LDS and in global memory, ceg_math.h and libm-grade rule arithmetic).
Oracle: oracle_single_contribution_vdw (energy.jl:397-427).  Run with `pytest -m gpu` on an MI355X."""
import ctypes as C
from pathlib import Path

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


# ------------------------------------------------------------------ BASELINE config 5: device-resident MC state
def _mc_setup(tmp_path):
    """Na + 4 CO2 in CIT-7 (2x3x3 supercell, triclinic): grids at 0.15 A built by the HIP kernels via setup_montecarlo."""
    import os
    from pathlib import Path
    from ceg_hip import montecarlo as M
    golden = Path(__file__).parent / "golden" / "raspa"
    raspa = tmp_path / "raspa"
    raspa.mkdir()
    for sub in ("forcefield", "molecules", "structures"):
        os.symlink(golden / sub, raspa / sub)
    ceg.setdir_RASPA(raspa)
    ff = "BoulfelfelSholl2021"

    def mol(name, positions):
        return ceg.load_molecule_RASPA(name, "TraPPE", ff).with_positions(positions)
    na = [[3.019388765467742, 0.8997706038543032, 26.11901621898599]]
    co2 = np.array([[11.93940309885289, 8.48657378465003, 2.135736631609201], [11.10485516124311, 7.710040763525694, 1.991767166323031],
                    [10.27030722363334, 6.933507742401357, 1.84779770103686]])
    shifts = [[0, 0, 0], [-5.6, -0.4, 6.5], [3.0, 9.0, 11.0], [-8.0, 14.0, 4.0]]
    return M, M.setup_montecarlo("CIT-7", ff, [mol("Na", na)] + [mol("CO2", co2 + np.array(s)) for s in shifts])


def _rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


def test_mc_replay_1000_moves(hip_lib, tmp_path):
    """BASELINE config 5 as the reference runs it (montecarlo.jl:563-628, simulation.jl:727-781): one trial per Markov step.
    A fixed sequence of 1000 translation / rotation moves with an energy-independent acceptance pattern is replayed on
    the device-resident state (ceg_mc_trial: ONE launch per step, ceg_mc_accept: update_mc! on the device, nothing uploaded
    between moves) and on the host mirror ceg_hip.montecarlo; every movement_energy (before, after; four terms) must agree
    to 1e-9, and so must the final positions and total structure factor."""
    from ceg_hip.energy import DeviceMonteCarlo
    try:
        M, mc = _mc_setup(tmp_path)
        M.baseline_energy(mc)                                   # host: per-molecule structure factors (mc.sums)
        dev = DeviceMonteCarlo(mc)
        mols = [(i, j) for i, kind in enumerate(mc.positions) for j in range(len(kind))]
        rng = np.random.default_rng(2024)
        worst = 0.0
        naccept = 0
        for step in range(1000):
            idx = mols[int(rng.integers(len(mols)))]
            cur = mc.positions[idx[0]][idx[1]]
            if step % 7 == 3:                                   # a jump anywhere in (and beyond) the MC cell
                new = cur + mc.mat @ rng.uniform(-1.5, 1.5, 3)
            else:
                new = cur + rng.uniform(-0.35, 0.35, 3)
            if len(cur) > 1 and step % 2 == 0:                  # rigid rotation about the centre atom
                c = new[len(cur) // 2]
                new = c + (new - c) @ _rotation(rng).T
            got = dev.trial(idx, new[None])
            before, after = M.movement_energy(mc, idx), M.movement_energy(mc, idx, new)
            for row, ref in ((got[0], before), (got[1], after)):
                r = np.array([ref.framework_vdw, ref.framework_direct, ref.inter, ref.reciprocal])
                ok = np.isfinite(r) & (np.abs(r) < 1e90)
                assert np.array_equal(row[~ok] >= 1e90, r[~ok] >= 1e90) or not (~ok).any(), (step, row, r)
                err = np.abs(row[ok] - r[ok]) / (1e-9 * np.abs(r[ok]) + 1e-7)
                worst = max(worst, float(err.max()) if ok.any() else 0.0)
                assert (err <= 1.0).all(), (step, idx, row, r)
            if step % 3 != 0:                                   # energy-independent acceptance pattern
                dev.accept(idx, new)
                M.update_mc(mc, idx, new)
                naccept += 1
        assert naccept > 600
        pos, sf = dev.state()
        ref_pos = np.concatenate([p for _i, _j, _ids, p in mc.molecules()])
        assert np.array_equal(pos, ref_pos)                     # positions are copied, not recomputed
        scale = np.abs(mc.sums[:, 0]).max()
        assert np.abs(sf - mc.sums[:, 0]).max() <= 1e-9 * scale
        # the same through a batch: 64 placements of one CO2 in one launch == the step-by-step rows
        idx = mols[2]
        cur = mc.positions[idx[0]][idx[1]]
        batch = cur[None] + rng.uniform(-1.0, 1.0, (64, 1, 3))
        rows = dev.trial(idx, batch)
        for t in (0, 17, 63):
            ref = M.movement_energy(mc, idx, batch[t])
            r = np.array([ref.framework_vdw, ref.framework_direct, ref.inter, ref.reciprocal])
            ok = np.abs(r) < 1e90
            assert np.all(np.abs(rows[1 + t][ok] - r[ok]) <= 1e-9 * np.abs(r[ok]) + 1e-7)
        # baseline_energy of the final configuration from the device state == the host mirror's
        b_dev, b_host = dev.baseline_energy(), M.baseline_energy(mc)
        for name in ("framework_vdw", "framework_direct", "inter", "reciprocal"):
            assert getattr(b_dev, name) == pytest.approx(getattr(b_host, name), rel=1e-9, abs=1e-6), name
        print(f"mc replay: 1000 moves, {naccept} accepted, worst error {worst:.2e} of the tolerance")
        dev.close()
    finally:
        ceg.setdir_RASPA(Path(__file__).parent / "golden" / "raspa")


def test_mc_insertions_and_removals(hip_lib, tmp_path):
    """GCMC swaps on the device-resident state (ceg_mc_trial_insert / ceg_mc_insert / ceg_mc_remove = movement_energy with
    ij < 0, add_one_system!, remove_one_system!; ewald.jl:704-728,775-810) interleaved with displacements, against the host
    mirror: insertion energies, the energies of every later move (they see the inserted / miss the removed molecules in the
    pair sum and in the total structure factor), final positions and structure factor."""
    from ceg_hip.energy import DeviceMonteCarlo
    try:
        M, mc = _mc_setup(tmp_path)
        M.baseline_energy(mc)
        dev = DeviceMonteCarlo(mc)
        rng = np.random.default_rng(77)
        base = mc.positions[1][0] - mc.positions[1][0][1]              # CO2 geometry about its carbon
        na = np.zeros((1, 3))

        def check(row, ref, what):
            r = np.array([ref.framework_vdw, ref.framework_direct, ref.inter, ref.reciprocal])
            ok = np.abs(r) < 1e90
            assert np.all(np.abs(row[ok] - r[ok]) <= 1e-9 * np.abs(r[ok]) + 1e-7), (what, row, r)

        nins = nrem = 0
        for step in range(240):
            kind = int(rng.integers(2))
            op = step % 4
            if op == 0:                                              # insertion trial batch + insertion of one of them
                shape = (na if kind == 0 else base @ _rotation(rng).T)
                trials = (mc.mat @ rng.uniform(0, 1, (5, 3)).T).T[:, None, :] + shape[None]
                rows = dev.trial_insert(kind, trials)
                for t in (0, 4):
                    check(rows[t], M.insertion_energy(mc, kind, trials[t]), ("insert", step, t))
                dev.insert(kind, trials[2])
                M.add_molecule(mc, kind, trials[2])
                nins += 1
            elif op == 2 and len(mc.positions[kind]) > 1:            # deletion: energy of the molecule where it is, then remove
                j = int(rng.integers(len(mc.positions[kind])))
                row = dev.trial((kind, j), np.empty((0, len(mc.ffidx[kind]), 3)))[0]
                check(row, M.movement_energy(mc, (kind, j)), ("delete", step))
                dev.remove((kind, j))
                M.remove_molecule(mc, (kind, j))
                nrem += 1
            else:                                                    # displacement
                if not mc.positions[kind]:
                    continue
                j = int(rng.integers(len(mc.positions[kind])))
                cur = mc.positions[kind][j]
                new = cur + rng.uniform(-0.4, 0.4, 3)
                got = dev.trial((kind, j), new[None])
                check(got[0], M.movement_energy(mc, (kind, j)), ("before", step))
                check(got[1], M.movement_energy(mc, (kind, j), new), ("after", step))
                if step % 3:
                    dev.accept((kind, j), new)
                    M.update_mc(mc, (kind, j), new)
        assert nins == 60 and nrem > 30
        pos, sf = dev.state()
        ref_pos = np.concatenate([p for _i, _j, _ids, p in mc.molecules()])
        assert np.array_equal(pos, ref_pos)
        assert np.abs(sf - mc.sums[:, 0]).max() <= 1e-9 * np.abs(mc.sums[:, 0]).max()
        dev.close()
    finally:
        ceg.setdir_RASPA(Path(__file__).parent / "golden" / "raspa")
