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
from ceg_hip.hostmirror.constants import COULOMBIC_CONVERSION_FACTOR

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
    from ceg_hip.hostmirror.interactions import FF
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


@pytest.mark.parametrize("cell", ["upper-triangular", "general"])
def test_pairs_fractional_kernel(hip_lib, oracle, forcefield, monkeypatch, cell):
    """k_pairs_frac (round 4: pair tests on fractional coordinates, trial atoms K at a time, 384-entry hit queue worked off in full
    batches) for molecules of 1-16 atoms (the exact-size variants 1-4 and the four-at-a-time variant), in an upper-triangular and in a
    general cell, with and without neighbour cells -- against the oracle at 1e-9 and against the Cartesian kernel
    (CEG_HIP_PAIRS_FRAC=0) and the literal wrap (CEG_HIP_PAIRS_WRAP=0), which test the same pairs."""
    ff = forcefield
    rng = np.random.default_rng(41)
    mat0, pos, kinds, mol, base, ids = _config5_guests(ff, rng)
    if cell == "general":
        rot = _rotation(rng)
        mat = rot @ mat0
        pos = pos @ rot.T
    else:
        mat = mat0
    inv = np.linalg.inv(mat)
    rules, offsets = ff.pair_table()
    pool = [ff.sdict[a] - 1 for a in ("C_co2", "O_co2", "Na", "C_ch4", "H_ch4")]
    for m, n in ((1, 4096), (2, 2048), (3, 4096), (4, 2048), (5, 2048), (9, 1024), (16, 1024)):
        tk = [pool[i % len(pool)] for i in range(m)]
        tbase = rng.uniform(-1.6, 1.6, (m, 3))
        trial = (rng.uniform(-0.5, 1.5, (n, 3)) @ mat.T)[:, None, :] + tbase[None]
        cl = (rng.uniform(0, 6.0, (n // 8, 3)) + np.array([17.0, 21.0, 9.0]))
        trial[: n // 8] = (cl @ (mat @ np.linalg.inv(mat0)).T)[:, None, :] + tbase[None]                      # into the dense cluster
        args = (mat, ff.cutoff ** 2, rules, offsets, ff.nkinds, COULOMBIC_CONVERSION_FACTOR, pos, kinds, mol, trial, tk, 500)
        ref = oracle.single_contribution_vdw_raw(mat, inv, *args[1:])
        monkeypatch.delenv("CEG_HIP_PAIRS_FRAC", raising=False)
        monkeypatch.delenv("CEG_HIP_PAIRS_WRAP", raising=False)
        got = _pairs_gpu(hip_lib, *args)
        _assert_energies(got, ref, f"fractional kernel, {m} atoms, {cell}")
        monkeypatch.setenv("CEG_HIP_PAIRS_FRAC", "0")
        cart = _pairs_gpu(hip_lib, *args)
        _assert_energies(cart, ref, f"Cartesian kernel, {m} atoms, {cell}")
        fin = np.isfinite(ref)
        assert np.array_equal(got[~fin], cart[~fin])
        assert np.all(np.abs(got[fin] - cart[fin]) <= 1e-10 * (np.abs(cart[fin]) + 1e-3 * np.percentile(np.abs(ref[fin]), 75)))
        if m in (3, 5):
            monkeypatch.setenv("CEG_HIP_PAIRS_WRAP", "0")
            _assert_energies(_pairs_gpu(hip_lib, *args), ref, f"literal wrap, {m} atoms, {cell}")
    # with neighbour cells (forced on in this 40 A cell: 2.5 A bins)
    monkeypatch.delenv("CEG_HIP_PAIRS_FRAC", raising=False)
    monkeypatch.delenv("CEG_HIP_PAIRS_WRAP", raising=False)
    monkeypatch.setenv("CEG_HIP_MC_CELLS", "1")
    monkeypatch.setenv("CEG_HIP_MC_BIN", "2.5")
    for m, n in ((3, 4096), (6, 1024)):
        tk = [pool[i % len(pool)] for i in range(m)]
        tbase = rng.uniform(-1.6, 1.6, (m, 3))
        trial = (rng.uniform(-0.5, 1.5, (n, 3)) @ mat.T)[:, None, :] + tbase[None]
        args = (mat, ff.cutoff ** 2, rules, offsets, ff.nkinds, COULOMBIC_CONVERSION_FACTOR, pos, kinds, mol, trial, tk, 500)
        ref = oracle.single_contribution_vdw_raw(mat, inv, *args[1:])
        _assert_energies(_pairs_gpu(hip_lib, *args), ref, f"fractional kernel + neighbour cells, {m} atoms, {cell}")


@pytest.mark.parametrize("cell", ["upper-triangular", "general"])
def test_pairs_cutoff_decision_is_the_references(hip_lib, oracle, forcefield, monkeypatch, cell):
    """The approximate pair distances (fractional-coordinate form, f - rint(f) with FMAs) never decide r2 < cutoff2 (energy.jl:422):
    pairs within 1e-9 of the cutoff are re-measured in the reference's operation order (utils.jl:294-302).  One guest atom, one
    trial atom per placement at cutoff * (1 + delta), delta from 0 to 1e-8 either side, in random directions, across the cell
    faces: whether the pair counts must agree with the oracle placement by placement, in all three kernel forms."""
    ff = forcefield
    rng = np.random.default_rng(43)
    mat = np.array([[40.0, 0, 0], [3.0, 40.0, 0], [-2.0, 4.0, 40.0]]).T
    if cell == "general":
        mat = _rotation(rng) @ mat
    inv = np.linalg.inv(mat)
    rules, offsets = ff.pair_table()
    kind = ff.sdict["O_co2"] - 1
    for atom_frac in ([0.31, 0.52, 0.47], [0.02, 0.97, 0.5], [1.99, -0.98, 0.01]):
        atom = mat @ np.array(atom_frac)
        deltas = np.array([0.0, 1e-16, -1e-16, 3e-16, -3e-16, 1e-14, -1e-14, 1e-12, -1e-12, 1e-10, -1e-10, 1e-8, -1e-8])
        n = 256 * len(deltas)
        u = rng.normal(size=(n, 3))
        u /= np.linalg.norm(u, axis=1)[:, None]
        d = np.tile(deltas, 256)
        trial = (atom[None] + ff.cutoff * (1.0 + d)[:, None] * u)[:, None, :]
        trial[::3] += (mat @ rng.integers(-2, 3, (3, len(trial[::3])))).T[:, None, :]      # other images of the same placement
        args = (mat, ff.cutoff ** 2, rules, offsets, ff.nkinds, COULOMBIC_CONVERSION_FACTOR, atom[None], [kind], [0], trial, [kind], -1)
        ref = oracle.single_contribution_vdw_raw(mat, inv, *args[1:])
        counted = ref != 0.0
        assert 0.3 < counted.mean() < 0.7                                  # both sides of the cutoff are populated
        for env in ({}, {"CEG_HIP_PAIRS_FRAC": "0"}, {"CEG_HIP_PAIRS_WRAP": "0"}):
            monkeypatch.delenv("CEG_HIP_PAIRS_FRAC", raising=False)
            monkeypatch.delenv("CEG_HIP_PAIRS_WRAP", raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            got = _pairs_gpu(hip_lib, *args)
            assert np.array_equal(got != 0.0, counted), f"{cell} {env}: {int(((got != 0.0) != counted).sum())} cutoff decisions differ from the oracle's"
            np.testing.assert_allclose(got, ref, rtol=1e-9, atol=0.0)


# ------------------------------------------------------------------ BASELINE config 5: device-resident MC state
def _mc_setup(tmp_path):
    """Na + 4 CO2 in CIT-7 (2x3x3 supercell, triclinic): grids at 0.15 A built by the HIP kernels via setup_montecarlo."""
    import os
    from pathlib import Path
    from ceg_hip.hostmirror import montecarlo as M
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
    between moves) and on the ORACLE's state (oracle/montecarlo.OracleMonteCarlo: movement_energy composed of the C restatements
    oracle_interpolate_grid + oracle_single_contribution_vdw + power-table structure factors + the rest sum of ewald.jl:718-737,
    nothing of the product package in it); every movement_energy (before, after; four terms) must agree to 1e-9, and so must
    the final positions and total structure factor.  The host mirror ceg_hip.hostmirror.montecarlo is checked as a second assert."""
    from ceg_hip.energy import DeviceMonteCarlo
    from oracle.montecarlo import OracleMonteCarlo
    try:
        M, mc = _mc_setup(tmp_path)
        M.baseline_energy(mc)                                   # host: per-molecule structure factors (mc.sums)
        dev = DeviceMonteCarlo(mc)
        omc = OracleMonteCarlo.from_setup(mc)
        omc.compute_ewald()
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
            for row, r, mirror in ((got[0], omc.movement_energy(idx), before), (got[1], omc.movement_energy(idx, new), after)):
                ok = np.isfinite(r) & (np.abs(r) < 1e90)
                assert np.array_equal(row[~ok] >= 1e90, r[~ok] >= 1e90) or not (~ok).any(), (step, row, r)
                err = np.abs(row[ok] - r[ok]) / (1e-9 * np.abs(r[ok]) + 1e-7)
                worst = max(worst, float(err.max()) if ok.any() else 0.0)
                assert (err <= 1.0).all(), (step, idx, row, r)
                m = np.array([mirror.framework_vdw, mirror.framework_direct, mirror.inter, mirror.reciprocal])      # second assert
                assert np.all(np.abs(row[ok] - m[ok]) <= 1e-9 * np.abs(m[ok]) + 1e-7), (step, idx, row, m)
            if step % 3 != 0:                                   # energy-independent acceptance pattern
                dev.accept(idx, new)
                omc.update(idx, new)
                M.update_mc(mc, idx, new)
                naccept += 1
        assert naccept > 600
        pos, sf = dev.state()
        assert np.array_equal(pos, omc.flat_positions())        # positions are copied, not recomputed
        osf = omc.total_structure_factor()
        scale = np.abs(osf).max()
        assert np.abs(sf - osf).max() <= 1e-9 * scale
        assert np.abs(sf - mc.sums[:, 0]).max() <= 1e-9 * scale
        # the same through a batch: 64 placements of one CO2 in one launch == the step-by-step rows
        idx = mols[2]
        cur = mc.positions[idx[0]][idx[1]]
        batch = cur[None] + rng.uniform(-1.0, 1.0, (64, 1, 3))
        rows = dev.trial(idx, batch)
        for t in (0, 17, 63):
            r = omc.movement_energy(idx, batch[t])
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
    ij < 0, add_one_system!, remove_one_system!; ewald.jl:704-728,775-810) interleaved with displacements, against the ORACLE's
    state (oracle/montecarlo.OracleMonteCarlo; the host mirror as a second assert): insertion energies, the energies of every
    later move (they see the inserted / miss the removed molecules in the pair sum and in the total structure factor), final
    positions and structure factor."""
    from ceg_hip.energy import DeviceMonteCarlo
    from oracle.montecarlo import OracleMonteCarlo
    try:
        M, mc = _mc_setup(tmp_path)
        M.baseline_energy(mc)
        dev = DeviceMonteCarlo(mc)
        omc = OracleMonteCarlo.from_setup(mc)
        omc.compute_ewald()
        rng = np.random.default_rng(77)
        base = mc.positions[1][0] - mc.positions[1][0][1]              # CO2 geometry about its carbon
        na = np.zeros((1, 3))

        def check(row, r, ref, what):
            ok = np.abs(r) < 1e90
            assert np.array_equal(row[~ok] >= 1e90, r[~ok] >= 1e90), (what, row, r)
            assert np.all(np.abs(row[ok] - r[ok]) <= 1e-9 * np.abs(r[ok]) + 1e-7), (what, row, r)
            m = np.array([ref.framework_vdw, ref.framework_direct, ref.inter, ref.reciprocal])                      # second assert: the mirror
            assert np.all(np.abs(row[ok] - m[ok]) <= 1e-9 * np.abs(m[ok]) + 1e-7), (what, row, m)

        nins = nrem = 0
        for step in range(240):
            kind = int(rng.integers(2))
            op = step % 4
            if op == 0:                                              # insertion trial batch + insertion of one of them
                shape = (na if kind == 0 else base @ _rotation(rng).T)
                trials = (mc.mat @ rng.uniform(0, 1, (5, 3)).T).T[:, None, :] + shape[None]
                rows = dev.trial_insert(kind, trials)
                for t in (0, 4):
                    check(rows[t], omc.insertion_energy(kind, trials[t]), M.insertion_energy(mc, kind, trials[t]), ("insert", step, t))
                dev.insert(kind, trials[2])
                assert omc.add(kind, trials[2]) == M.add_molecule(mc, kind, trials[2])
                nins += 1
            elif op == 2 and len(mc.positions[kind]) > 1:            # deletion: energy of the molecule where it is, then remove
                j = int(rng.integers(len(mc.positions[kind])))
                row = dev.trial((kind, j), np.empty((0, len(mc.ffidx[kind]), 3)))[0]
                check(row, omc.movement_energy((kind, j)), M.movement_energy(mc, (kind, j)), ("delete", step))
                assert dev.remove((kind, j)) == omc.remove((kind, j)) == M.remove_molecule(mc, (kind, j))     # the last molecule of the kind takes index j
                nrem += 1
            else:                                                    # displacement
                if not mc.positions[kind]:
                    continue
                j = int(rng.integers(len(mc.positions[kind])))
                cur = mc.positions[kind][j]
                new = cur + rng.uniform(-0.4, 0.4, 3)
                got = dev.trial((kind, j), new[None])
                check(got[0], omc.movement_energy((kind, j)), M.movement_energy(mc, (kind, j)), ("before", step))
                check(got[1], omc.movement_energy((kind, j), new), M.movement_energy(mc, (kind, j), new), ("after", step))
                if step % 3:
                    dev.accept((kind, j), new)
                    omc.update((kind, j), new)
                    M.update_mc(mc, (kind, j), new)
        assert nins == 60 and nrem > 30
        pos, sf = dev.state()
        assert np.array_equal(pos, omc.flat_positions())
        osf = omc.total_structure_factor()
        assert np.abs(sf - osf).max() <= 1e-9 * np.abs(osf).max()
        assert np.abs(sf - mc.sums[:, 0]).max() <= 1e-9 * np.abs(mc.sums[:, 0]).max()
        dev.close()
    finally:
        ceg.setdir_RASPA(Path(__file__).parent / "golden" / "raspa")


def test_mc_large_batches_take_the_wave_kernels(hip_lib, tmp_path, monkeypatch):
    """From 1024 rows on a trial batch runs as three wave-per-placement launches (k_mcw_frame / k_mcw_ewald / k_mcw_pairs: one grid
    corner per lane, k-space constants staged once per workgroup + row-wise k-vector walk, the pair-table rows of the molecule's
    kinds in LDS) instead of one workgroup per placement.  Displacement batches (row 0 = the molecule where it is, with its STORED
    structure factor) and insertion batches of both species, after some accepted moves and an insertion, sampled against the ORACLE's
    movement_energy (1e-9) and, row for row, against the workgroup-per-placement kernel forced on the same batch (1e-10)."""
    from ceg_hip.energy import DeviceMonteCarlo
    from oracle.montecarlo import OracleMonteCarlo
    try:
        M, mc = _mc_setup(tmp_path)
        M.baseline_energy(mc)
        dev = DeviceMonteCarlo(mc)
        omc = OracleMonteCarlo.from_setup(mc)
        omc.compute_ewald()
        rng = np.random.default_rng(404)
        base = mc.positions[1][0] - mc.positions[1][0][1]
        # move things first: accepted displacements and one insertion, so that sums[:, 1] and sums[:, ij+1] are device-updated values
        for step in range(12):
            kind = step % 2
            j = int(rng.integers(len(mc.positions[kind])))
            new = mc.positions[kind][j] + rng.uniform(-0.6, 0.6, 3)
            dev.accept((kind, j), new); omc.update((kind, j), new); M.update_mc(mc, (kind, j), new)
        extra = mc.mat @ np.array([0.41, 0.13, 0.77]) + base @ _rotation(rng).T
        dev.insert(1, extra); omc.add(1, extra); M.add_molecule(mc, 1, extra)

        def check_rows(rows, refs, what):
            for t, r in refs.items():
                ok = np.abs(r) < 1e90
                assert np.array_equal(rows[t][~ok] >= 1e90, r[~ok] >= 1e90), (what, t, rows[t], r)
                assert np.all(np.abs(rows[t][ok] - r[ok]) <= 1e-9 * np.abs(r[ok]) + 1e-7), (what, t, rows[t], r)

        def both_paths(call):
            monkeypatch.setenv("CEG_HIP_MC_WAVE_MIN", "0")
            wave = call()
            monkeypatch.setenv("CEG_HIP_MC_WAVE_MIN", "1000000000")
            group = call()                                         # one workgroup per row (more than 256 rows)
            monkeypatch.setenv("CEG_HIP_MC_SPLIT_MAX", "1000000000")
            split = call()                                         # the three terms of a row on three workgroups, as for small batches
            monkeypatch.delenv("CEG_HIP_MC_SPLIT_MAX")
            monkeypatch.delenv("CEG_HIP_MC_WAVE_MIN")
            assert np.array_equal(split, group, equal_nan=True)    # the same arithmetic in the same order
            assert np.array_equal(np.abs(wave) >= 1e90, np.abs(group) >= 1e90)
            for c in range(4):            # same terms, another order (columns 0, 1, 3) / another arithmetic (column 2: fractional coordinates, tabulated
                                          # erfc): 1e-10 of the value, floored at 1e-11 of the column's upper quartile
                ok = (np.abs(group[:, c]) < 1e90) & np.isfinite(group[:, c])
                scale = float(np.percentile(np.abs(group[ok, c]), 75)) if ok.any() else 0.0
                err = np.abs(wave[ok, c] - group[ok, c])
                bad = err > 1e-10 * np.abs(group[ok, c]) + 1e-11 * scale + 1e-300
                assert not bad.any(), (c, float(err.max()), scale, wave[ok, c][bad][:4], group[ok, c][bad][:4])
            return wave

        n = 3000
        for kind, j in ((1, 2), (0, 0)):
            cur = mc.positions[kind][j]
            trial = cur[None] + rng.uniform(-1.2, 1.2, (n, 1, 3))
            trial[1::5] = (rng.uniform(-0.5, 1.5, (len(trial[1::5]), 3)) @ mc.mat.T)[:, None, :] + (cur - cur[len(cur) // 2])[None]   # anywhere, also outside the cell
            if len(cur) > 1:
                for t in range(0, n, 3):
                    c = trial[t][1]
                    trial[t] = c + (trial[t] - c) @ _rotation(rng).T
            rows = both_paths(lambda: dev.trial((kind, j), trial))
            assert rows.shape == (n + 1, 4)
            refs = {0: omc.movement_energy((kind, j))}
            for t in (0, 1, 6, 511, 512, 1777, n - 1):
                refs[1 + t] = omc.movement_energy((kind, j), trial[t])
            check_rows(rows, refs, ("displacement", kind))
            assert (np.abs(rows[:, 0]) >= 1e90).any() and (np.abs(rows[:, 0]) < 1e90).sum() > n // 4      # blocked and open placements
            # the default routing: this batch is large enough for the wave kernels, a 64-row one is not -- same numbers either way
            np.testing.assert_allclose(dev.trial((kind, j), trial[:63])[:, 2:], rows[:64, 2:], rtol=1e-10, atol=1e-7)
        for kind in (0, 1):
            shape = np.zeros((1, 3)) if kind == 0 else base
            trial = (rng.uniform(0, 1, (n, 3)) @ mc.mat.T)[:, None, :] + shape[None]
            rows = both_paths(lambda: dev.trial_insert(kind, trial))
            assert rows.shape == (n, 4)
            check_rows(rows, {t: omc.insertion_energy(kind, trial[t]) for t in (0, 5, 640, n - 1)}, ("insertion", kind))
        # the device-pointer entry points (ceg_mc_trial_device / _insert_device): trials and rows stay on the GPU, launches on the
        # caller's stream, ordered behind an asynchronous accept and in front of the next one -- the same rows as the host entry points
        import torch
        kind, j = 1, 1
        cur = mc.positions[kind][j]
        trial = cur[None] + rng.uniform(-1.0, 1.0, (777, 1, 3))                 # (small too: the device route always takes the wave kernels)
        moved = cur + np.array([0.3, -0.2, 0.1])
        dev.accept((kind, j), moved); omc.update((kind, j), moved); M.update_mc(mc, (kind, j), moved)
        d_trial = torch.tensor(trial, dtype=torch.float64, device="cuda")
        d_rows = torch.full((len(trial) + 1, 4), float("nan"), dtype=torch.float64, device="cuda")
        side = torch.cuda.Stream()
        dev.trial_device((kind, j), d_trial.data_ptr(), len(trial), d_rows.data_ptr(), side.cuda_stream)
        back = cur + np.array([-0.1, 0.2, 0.0])
        dev.accept((kind, j), back)                                               # must not overtake the trial enqueued before it
        side.synchronize()
        rows_dev = d_rows.cpu().numpy()
        check_rows(rows_dev, {0: omc.movement_energy((kind, j)), 1: omc.movement_energy((kind, j), trial[0]), 500: omc.movement_energy((kind, j), trial[499])},
                   "device-pointer trial")
        omc.update((kind, j), back); M.update_mc(mc, (kind, j), back)
        rows_host = dev.trial((kind, j), trial)
        assert np.all(np.abs(rows_host[0] - omc.movement_energy((kind, j))) <= 1e-9 * np.abs(rows_host[0]) + 1e-7)
        d_rows_i = torch.empty((len(trial), 4), dtype=torch.float64, device="cuda")
        dev.trial_insert_device(1, d_trial.data_ptr(), len(trial), d_rows_i.data_ptr())
        torch.cuda.synchronize()
        np.testing.assert_allclose(d_rows_i.cpu().numpy(), dev.trial_insert(1, trial), rtol=1e-10, atol=1e-7)
        # argument checks of the device entry points: the reference would raise (BoundsError / ArgumentError) -- here CEG_ERR_INVALID
        lib, hnd, slot = dev._lib, dev._h, dev._slot[kind][j]
        assert lib.ceg_mc_trial_device(hnd, slot, C.c_void_p(d_trial.data_ptr()), -1, C.c_void_p(d_rows.data_ptr()), None) == -1
        assert lib.ceg_mc_trial_device(hnd, slot, C.c_void_p(d_trial.data_ptr()), 4, None, None) == -1
        assert lib.ceg_mc_trial_device(hnd, 10 ** 6, C.c_void_p(d_trial.data_ptr()), 4, C.c_void_p(d_rows.data_ptr()), None) == -1
        assert lib.ceg_mc_trial_device(None, slot, C.c_void_p(d_trial.data_ptr()), 4, C.c_void_p(d_rows.data_ptr()), None) == -1
        bad = np.array([10 ** 6], dtype=np.int32)
        assert lib.ceg_mc_trial_insert_device(hnd, _abi.i32ptr(bad), 1, C.c_void_p(d_trial.data_ptr()), 4, C.c_void_p(d_rows.data_ptr()), None) == -1
        # no trial placement at all: row 0 (the molecule where it is) alone
        d_rows.fill_(float("nan"))
        _abi.check(lib, lib.ceg_mc_trial_device(hnd, slot, None, 0, C.c_void_p(d_rows.data_ptr()), None))
        torch.cuda.synchronize()
        np.testing.assert_allclose(d_rows[0].cpu().numpy(), rows_host[0], rtol=1e-10, atol=1e-7)
        assert bool(torch.isnan(d_rows[1:]).all())
        dev.close()
    finally:
        ceg.setdir_RASPA(Path(__file__).parent / "golden" / "raspa")


def test_mc_handle_refuses_work_after_a_failed_update(hip_lib, tmp_path, monkeypatch):
    """accept / insert / remove change the host mirror before the launch is known to have succeeded; when one of them fails
    after that point (injected: CEG_HIP_MC_INJECT_FAILURE) the handle must refuse every later call -- instead of answering from
    host and device state that no longer agree -- until ceg_mc_set_guests (refresh) rebuilds both."""
    from ceg_hip.energy import DeviceMonteCarlo
    try:
        M, mc = _mc_setup(tmp_path)
        M.baseline_energy(mc)
        dev = DeviceMonteCarlo(mc)
        ref = dev.trial((1, 0), np.empty((0, 3, 3)))[0].copy()
        pos0, sf0 = dev.state()
        for what, call in (("accept", lambda: dev.accept((1, 0), mc.positions[1][0] + 0.1)),
                           ("insert", lambda: dev.insert(0, np.array([[3.0, 4.0, 5.0]]))),
                           ("remove", lambda: dev.remove((1, 1)))):
            monkeypatch.setenv("CEG_HIP_MC_INJECT_FAILURE", what)
            with pytest.raises(_abi.CegError) as ei:
                call()
            assert ei.value.code == -3, what                       # CEG_ERR_HIP
            monkeypatch.delenv("CEG_HIP_MC_INJECT_FAILURE")
            for later in (lambda: dev.trial((1, 0), np.empty((0, 3, 3))), lambda: dev.accept((1, 0), mc.positions[1][0]),
                          lambda: dev.trial_insert(0, np.zeros((1, 1, 3))), lambda: dev.state()):
                with pytest.raises(_abi.CegError) as ei:
                    later()
                assert ei.value.code == -3 and "ceg_mc_set_guests" in str(ei.value), what
            dev.refresh()                                          # the host-side mc was never touched: same state as before
            np.testing.assert_allclose(dev.trial((1, 0), np.empty((0, 3, 3)))[0], ref, rtol=1e-12, atol=1e-9)
            pos, sf = dev.state()
            assert np.array_equal(pos, pos0) and np.abs(sf - sf0).max() <= 1e-12 * np.abs(sf0).max()
        dev.close()
    finally:
        ceg.setdir_RASPA(Path(__file__).parent / "golden" / "raspa")


# ------------------------------------------------------------------ neighbour cells of the device-resident MC state
class _RawMc:
    """ceg_mc_* through the C ABI on explicit tables: no framework grids, no Ewald summation -- the guest-guest term alone."""

    def __init__(self, lib, mat, cutoff2, rules, offsets, nkinds, coulombic):
        self.lib = lib
        matT = np.ascontiguousarray(np.asarray(mat, dtype=np.float64).T.reshape(9))
        invT = np.ascontiguousarray(np.linalg.inv(np.asarray(mat, dtype=np.float64)).T.reshape(9))
        self.h = C.c_void_p()
        charge = np.zeros(nkinds)
        self._keep = (rules, np.ascontiguousarray(offsets, dtype=np.int32))
        _abi.check(lib, lib.ceg_mc_create(C.byref(self.h), 0, None, None, _abi.dptr(charge), int(nkinds), _abi.dptr(matT), _abi.dptr(invT),
                                          float(cutoff2), rules.ctypes.data, _abi.i32ptr(self._keep[1]), float(coulombic),
                                          None, None, None, None, 0, None, None))

    def cells(self):
        nb = np.zeros(3, dtype=np.int32)
        cap = C.c_int32(0)
        rc = self.lib.ceg_mc_neighbour_cells(self.h, _abi.i32ptr(nb), C.byref(cap))
        assert rc in (0, 1)
        return (tuple(int(x) for x in nb), int(cap.value)) if rc else None

    def set_guests(self, pos, kinds, first):
        p = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1)
        _abi.check(self.lib, self.lib.ceg_mc_set_guests(self.h, _abi.dptr(p), _abi.i32ptr(np.ascontiguousarray(kinds, dtype=np.int32)),
                                                        _abi.i32ptr(np.ascontiguousarray(first, dtype=np.int32)), len(first) - 1))

    def trial(self, molecule, trial):
        t = np.ascontiguousarray(trial, dtype=np.float64)
        out = np.empty((len(t) + 1, 4))
        _abi.check(self.lib, self.lib.ceg_mc_trial(self.h, int(molecule), _abi.dptr(t.reshape(-1)), len(t), _abi.dptr(out.reshape(-1))))
        return out[:, 2]

    def trial_insert(self, kinds, trial):
        t = np.ascontiguousarray(trial, dtype=np.float64)
        k = np.ascontiguousarray(kinds, dtype=np.int32)
        out = np.empty((len(t), 4))
        _abi.check(self.lib, self.lib.ceg_mc_trial_insert(self.h, _abi.i32ptr(k), len(k), _abi.dptr(t.reshape(-1)), len(t), _abi.dptr(out.reshape(-1))))
        return out[:, 2]

    def accept(self, molecule, pos):
        _abi.check(self.lib, self.lib.ceg_mc_accept(self.h, int(molecule), _abi.dptr(np.ascontiguousarray(pos, dtype=np.float64).reshape(-1))))

    def insert(self, kinds, pos):
        k = np.ascontiguousarray(kinds, dtype=np.int32)
        out = C.c_int32(-1)
        _abi.check(self.lib, self.lib.ceg_mc_insert(self.h, _abi.i32ptr(k), len(k), _abi.dptr(np.ascontiguousarray(pos, dtype=np.float64).reshape(-1)), C.byref(out)))
        return int(out.value)

    def remove(self, molecule):
        moved = C.c_int32(-1)
        _abi.check(self.lib, self.lib.ceg_mc_remove(self.h, int(molecule), C.byref(moved)))
        return int(moved.value)

    def positions(self, natoms):
        pos = np.empty((natoms, 3))
        _abi.check(self.lib, self.lib.ceg_mc_get_state(self.h, _abi.dptr(pos.reshape(-1)), None, None))
        return pos

    def close(self):
        self.lib.ceg_mc_destroy(self.h)


def test_mc_neighbour_cells(hip_lib, oracle, forcefield, monkeypatch):
    """The guest-guest sum of the device-resident MC state over neighbour cells (the reference's CellListMap branch,
    energy.jl:341-349,399-404) in an MC cell of 5-7 cutoffs: 4500+ guest atoms, displacements within and across the periodic
    boundary, jumps, insertions into one spot until its cells outgrow their capacity, removals that renumber molecules --
    every energy against the exhaustive loop of a second handle (same pair tests, another summation order: 1e-10) and
    against oracle_single_contribution_vdw on samples (1e-9), with 4 A and 2.2 A bins."""
    ff = forcefield
    rng = np.random.default_rng(11)
    co2 = ceg.load_molecule_RASPA("CO2", "TraPPE", "BoulfelfelSholl2021")
    base = np.asarray(co2.position, dtype=np.float64).reshape(-1, 3)
    ids = [ff.sdict[a] - 1 for a in co2.atomic_symbol]
    na_id = [ff.sdict["Na"] - 1]
    mat = np.array([[72.0, 0, 0], [9.0, 66.0, 0], [-7.0, 11.0, 84.0]]).T
    inv = np.linalg.inv(mat)
    rules, offsets = ff.pair_table()
    table = (mat, ff.cutoff ** 2, rules, offsets, ff.nkinds, COULOMBIC_CONVERSION_FACTOR)
    # host copy of the state: list of (kinds, positions) in device molecule order
    mols = []
    for c in (rng.uniform(0, 1, (1500, 3)) @ mat.T):
        mols.append((ids, c + base @ _rotation(rng).T))
    for c in (rng.uniform(-0.5, 1.5, (24, 3)) @ mat.T):            # some guests listed outside the unit cell
        mols.append((na_id, c[None].copy()))

    def flat():
        pos = np.concatenate([p for _k, p in mols])
        kinds = np.array([k for ks, _p in mols for k in ks], dtype=np.int32)
        first = np.concatenate([[0], np.cumsum([len(ks) for ks, _p in mols])]).astype(np.int32)
        mol = np.repeat(np.arange(len(mols)), [len(ks) for ks, _p in mols]).astype(np.int32)
        return pos, kinds, first, mol

    handles = {}
    for name, env in (("exhaustive", {"CEG_HIP_MC_CELLS": "0"}), ("cells 4 A", {}), ("cells 2.2 A", {"CEG_HIP_MC_BIN": "2.2"})):
        monkeypatch.delenv("CEG_HIP_MC_CELLS", raising=False)
        monkeypatch.delenv("CEG_HIP_MC_BIN", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        handles[name] = _RawMc(hip_lib, *table)
    monkeypatch.delenv("CEG_HIP_MC_CELLS", raising=False)
    monkeypatch.delenv("CEG_HIP_MC_BIN", raising=False)
    try:
        assert handles["exhaustive"].cells() is None
        nb4, cap0 = handles["cells 4 A"].cells()
        assert nb4 == (17, 16, 21), nb4                               # perpendicular widths / 4 A
        pos, kinds, first, mol = flat()
        for h in handles.values():
            h.set_guests(pos, kinds, first)
        hot = mat @ np.array([0.31, 0.77, 0.52])
        worst = worst_oracle = 0.0
        noracle = 0

        def compare(rows, what, tol=1e-10):
            nonlocal worst
            ref = rows["exhaustive"]
            for name, got in rows.items():
                assert np.array_equal(np.isfinite(got), np.isfinite(ref)), (what, name)
                fin = np.isfinite(ref)
                assert np.array_equal(got[~fin], ref[~fin]), (what, name)
                err = np.abs(got[fin] - ref[fin]) / (np.abs(ref[fin]) + 1e-3)
                worst = max(worst, float(err.max()) if fin.any() else 0.0)
                assert (err <= tol).all(), (what, name, float(err.max()))

        for step in range(360):
            op = step % 6
            if op in (0, 1, 3, 4):                                     # displacement batch of one molecule
                j = int(rng.integers(len(mols)))
                ks, cur = mols[j]
                trial = cur[None] + rng.uniform(-0.5, 0.5, (8, 1, 3))
                trial[1] = cur + mat @ rng.uniform(-1.5, 1.5, 3)                       # a jump, possibly far outside the cell
                trial[2] = cur - cur[len(ks) // 2] + mat @ np.array([0.9999, 0.0001, 0.5])   # next to the cell boundary
                trial[3] = cur - cur[len(ks) // 2] + hot + rng.uniform(-1, 1, 3)       # into the crowded spot
                if len(ks) > 1:
                    trial[4] = trial[4][len(ks) // 2] + (trial[4] - trial[4][len(ks) // 2]) @ _rotation(rng).T
                rows = {name: h.trial(j, trial) for name, h in handles.items()}
                compare(rows, ("trial", step))
                if step % 30 == 0:
                    p, k, _f, m = flat()
                    ref = oracle.single_contribution_vdw_raw(mat, inv, *table[1:], p, k, m, np.concatenate([cur[None], trial]), ks, j)
                    worst_oracle = max(worst_oracle, _assert_energies(rows["cells 4 A"], ref, f"cells vs oracle, step {step}"))
                    noracle += 1
                pick = int(rng.integers(8))
                if step % 4 != 1:
                    for h in handles.values():
                        h.accept(j, trial[pick])
                    mols[j] = (ks, trial[pick].copy())
            elif op == 2:                                              # insertion next to the crowded spot
                ks = ids if step % 12 == 2 else na_id
                shape = base @ _rotation(rng).T if ks is ids else np.zeros((1, 3))
                trial = (hot + rng.uniform(-1.6, 1.6, (4, 3)))[:, None, :] + shape[None]
                rows = {name: h.trial_insert(ks, trial) for name, h in handles.items()}
                compare(rows, ("insert", step))
                new = {h.insert(ks, trial[1]) for h in handles.values()}
                assert new == {len(mols)}
                mols.append((ks, trial[1].copy()))
            else:                                                      # removal: the last molecule takes the index
                j = int(rng.integers(len(mols)))
                moved = {h.remove(j) for h in handles.values()}
                assert moved == {len(mols) - 1}
                mols[j] = mols[-1]
                mols.pop()
            if step % 45 == 44:
                # a batch large enough for the wave kernels (k_mcw_pairs_frac reads the FRACTIONAL copies of the atom and cell records that
                # every update above must have kept current): against the Cartesian wave kernel on the same state
                j = int(rng.integers(len(mols)))
                ks, cur = mols[j]
                trial = (rng.uniform(0, 1, (1500, 3)) @ mat.T)[:, None, :] + (cur - cur[0])[None]
                rows = {name: h.trial(j, trial) for name, h in handles.items()}
                compare(rows, ("wave batch", step))
                monkeypatch.setenv("CEG_HIP_MC_FRAC", "0")
                compare({"exhaustive": rows["exhaustive"], **{name + ", Cartesian": h.trial(j, trial) for name, h in handles.items()}}, ("wave batch, Cartesian", step), tol=1e-9)
                monkeypatch.delenv("CEG_HIP_MC_FRAC")
        assert noracle >= 8
        _nb, cap1 = handles["cells 4 A"].cells()
        assert cap1 > cap0, (cap0, cap1)                               # the crowded spot outgrew the first capacity
        pos, kinds, first, mol = flat()
        for name, h in handles.items():
            assert np.array_equal(h.positions(len(pos)), pos), name
        # a large batch through the device buffers (not the mapped ones), every 64th row against the oracle
        j = 700
        ks, cur = mols[j]
        trial = (rng.uniform(0, 1, (20000, 3)) @ mat.T)[:, None, :] + (cur - cur[1])[None]
        rows = {name: h.trial(j, trial) for name, h in handles.items()}
        compare(rows, "batch")
        ref = oracle.single_contribution_vdw_raw(mat, inv, *table[1:], pos, kinds, mol, trial[::64], ks, j)
        _assert_energies(rows["cells 4 A"][1::64], ref, "cells vs oracle, batch")
        # the same batch through the Cartesian wave kernel (k_mcw_pairs; the default above is k_mcw_pairs_frac), and an insertion batch
        monkeypatch.setenv("CEG_HIP_MC_FRAC", "0")
        compare({"exhaustive": rows["exhaustive"], **{name + ", Cartesian kernel": h.trial(j, trial) for name, h in handles.items()}}, "batch, Cartesian kernel",
                tol=1e-9)                                          # (another arithmetic, not only another order: the parity tolerance)
        monkeypatch.delenv("CEG_HIP_MC_FRAC")
        trial_i = (rng.uniform(0, 1, (6000, 3)) @ mat.T)[:, None, :] + (cur - cur[1])[None]
        trial_i[:500] = (hot + rng.uniform(-3.0, 3.0, (500, 3)))[:, None, :] + (cur - cur[1])[None]          # into the crowded spot
        rows_i = {name: h.trial_insert(ks, trial_i) for name, h in handles.items()}
        compare(rows_i, "insertion batch")
        ref_i = oracle.single_contribution_vdw_raw(mat, inv, *table[1:], pos, kinds, mol, trial_i[::32], ks, -1)
        _assert_energies(rows_i["cells 4 A"][::32], ref_i, "cells vs oracle, insertion batch")
        # a molecule of five atoms (methane-like: the any-size variant of k_mcw_pairs_frac, four trial atoms at a time) and one of nine
        t = 1.09 / np.sqrt(3.0)
        for ks_n, shape_n in (([ff.sdict["C_ch4"] - 1] + [ff.sdict["H_ch4"] - 1] * 4,
                               np.array([[0, 0, 0], [t, t, t], [t, -t, -t], [-t, t, -t], [-t, -t, t]], dtype=np.float64)),
                              ([ids[0], ids[1], ids[2]] * 3, rng.uniform(-2.0, 2.0, (9, 3)))):
            trial_n = (rng.uniform(0, 1, (3000, 3)) @ mat.T)[:, None, :] + shape_n[None]
            trial_n[:300] = (hot + rng.uniform(-3.0, 3.0, (300, 3)))[:, None, :] + shape_n[None]
            rows_n = {name: h.trial_insert(ks_n, trial_n) for name, h in handles.items()}
            compare(rows_n, f"insertion batch, {len(ks_n)} atoms")
            ref_n = oracle.single_contribution_vdw_raw(mat, inv, *table[1:], pos, kinds, mol, trial_n[::32], ks_n, -1)
            _assert_energies(rows_n["cells 4 A"][::32], ref_n, f"cells vs oracle, insertion batch of a {len(ks_n)}-atom molecule")
        print(f"neighbour cells: worst deviation from the exhaustive loop {worst:.1e}, from the oracle {worst_oracle:.1e}; capacity {cap0} -> {cap1}")
    finally:
        for h in handles.values():
            h.close()


def test_pairs_neighbour_cells(hip_lib, oracle, forcefield, monkeypatch):
    """Row f3 with the atoms sorted by neighbour cell (ceg_pairs_set_atoms; the reference's CellListMap branch,
    energy.jl:399-404) in an MC cell of 5-7 cutoffs: 6157 guest atoms incl. a dense cluster, 8192 placements (inside, across the
    periodic boundary, far outside the cell, into the cluster), excluded molecule, 4 A and 2.2 A bins -- against the oracle
    (1e-9) and against the exhaustive loop (same pairs, another order)."""
    ff = forcefield
    rng = np.random.default_rng(23)
    _m, pos, kinds, mol, base, ids = _config5_guests(ff, rng)
    mat = np.array([[72.0, 0, 0], [9.0, 66.0, 0], [-7.0, 11.0, 84.0]]).T
    extra = rng.uniform(-0.3, 1.3, (1000, 3)) @ mat.T                   # 1000 more CO2 anywhere in (and around) the big cell
    pos = np.concatenate([pos] + [c + base for c in extra])
    kinds = np.concatenate([kinds, np.tile(ids, 1000)]).astype(np.int32)
    mol = np.concatenate([mol, np.repeat(np.arange(2000, 3000), 3)]).astype(np.int32)
    assert len(pos) == 6157
    n = 8192
    trial = (rng.uniform(0, 1, (n, 3)) @ mat.T)[:, None, :] + base[None]
    trial[: n // 8] = (rng.uniform(0, 6.0, (n // 8, 3)) + np.array([17.0, 21.0, 9.0]))[:, None, :] + base[None]       # into the cluster
    trial[n // 8: n // 4] += 2.0 * (mat[:, 0] - mat[:, 2])[None, None, :]                                           # far outside the cell
    edge = rng.uniform(0, 1, (n // 8, 3))
    edge[np.arange(n // 8), rng.integers(0, 3, n // 8)] = rng.choice([1e-7, 1 - 1e-7, 0.0, 1.0], n // 8)
    trial[n // 4: n // 4 + n // 8] = (edge @ mat.T)[:, None, :] + base[None]                                        # on the cell faces
    rules, offsets = ff.pair_table()
    args = (mat, ff.cutoff ** 2, rules, offsets, ff.nkinds, COULOMBIC_CONVERSION_FACTOR, pos, kinds, mol, trial, ids, 500)
    ref = oracle.single_contribution_vdw_raw(args[0], np.linalg.inv(mat), *args[1:])
    monkeypatch.setenv("CEG_HIP_MC_CELLS", "0")
    exhaustive = _pairs_gpu(hip_lib, *args)
    _assert_energies(exhaustive, ref, "exhaustive, large cell")
    for env in ({}, {"CEG_HIP_MC_BIN": "2.2"}):
        monkeypatch.delenv("CEG_HIP_MC_CELLS", raising=False)
        monkeypatch.delenv("CEG_HIP_MC_BIN", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = C.c_void_p()                                                 # the library picks the cells by itself here
        matT = np.ascontiguousarray(mat.T.reshape(9)); invT = np.ascontiguousarray(np.linalg.inv(mat).T.reshape(9))
        _abi.check(hip_lib, hip_lib.ceg_pairs_create(C.byref(h), 0, _abi.dptr(matT), _abi.dptr(invT), ff.cutoff ** 2, rules.ctypes.data,
                                                     _abi.i32ptr(np.ascontiguousarray(offsets, dtype=np.int32)), ff.nkinds, COULOMBIC_CONVERSION_FACTOR))
        nb = np.zeros(3, dtype=np.int32)
        assert hip_lib.ceg_pairs_neighbour_cells(h, _abi.i32ptr(nb)) == 1 and tuple(nb) == ((17, 16, 21) if not env else (32, 29, 38))
        hip_lib.ceg_pairs_destroy(h)
        got = _pairs_gpu(hip_lib, *args)
        _assert_energies(got, ref, f"cells {env}")
        fin = np.isfinite(exhaustive)
        assert np.array_equal(got[~fin], exhaustive[~fin])
        assert np.all(np.abs(got[fin] - exhaustive[fin]) <= 1e-10 * (np.abs(exhaustive[fin]) + 1e-3))
        _assert_energies(_pairs_gpu(hip_lib, *args[:-1], -1), oracle.single_contribution_vdw_raw(args[0], np.linalg.inv(mat), *args[1:-1], -1),
                         f"cells {env}, none excluded")
    # the fixtures' own cells keep the exhaustive loop
    monkeypatch.delenv("CEG_HIP_MC_BIN", raising=False)
    small = np.array([[40.0, 0, 0], [3.0, 40.0, 0], [-2.0, 4.0, 40.0]]).T
    h = C.c_void_p()
    _abi.check(hip_lib, hip_lib.ceg_pairs_create(C.byref(h), 0, _abi.dptr(np.ascontiguousarray(small.T.reshape(9))),
                                                 _abi.dptr(np.ascontiguousarray(np.linalg.inv(small).T.reshape(9))), ff.cutoff ** 2, rules.ctypes.data,
                                                 _abi.i32ptr(np.ascontiguousarray(offsets, dtype=np.int32)), ff.nkinds, COULOMBIC_CONVERSION_FACTOR))
    assert hip_lib.ceg_pairs_neighbour_cells(h, None) == 0
    hip_lib.ceg_pairs_destroy(h)


def test_incremental_ewald_context_testset(hip_lib, monkeypatch):
    """The reference's "IncrementalEwaldContext" testset, runtests.jl:61-121, on the device-resident state: the flat index a
    removal returns (the last species takes the freed index, ewald.jl:403-431), the index an addition returns, and after every
    step the total guest structure factor (hence compute_ewald) equal to that of the equivalent system built from scratch."""
    from ceg_hip import grids as G
    from ceg_hip.hostmirror import montecarlo as M
    from ceg_hip.energy import DeviceMonteCarlo
    monkeypatch.setattr(M, "retrieve_or_create_grid", lambda *a, **k: G.EnergyGrid.trivial(True))      # Ewald only: no grids
    na = ceg.load_molecule_RASPA("Na", "TraPPE", "BoulfelfelSholl2021")
    co2 = ceg.load_molecule_RASPA("CO2", "TraPPE", "BoulfelfelSholl2021")
    pos1, pos2, pos3 = np.array([[1.0, 2.5, 1.7]]), np.array([[6.2, 5.1, 3.0]]), np.array([[4.1, 3.7, 2.2]])
    pco2 = np.asarray(co2.position, dtype=np.float64).reshape(-1, 3)
    co2_2 = np.array([[5.491645446274333, 8.057854365959964, 8.669190836544463], [6.335120278303245, 7.462084936052019, 9.172986424179925],
                      [7.178595110332157, 6.866315506144074, 9.676782011815387]])
    mc = M.setup_montecarlo("CHA_1.4_3b4eeb96", "BoulfelfelSholl2021", [na.with_positions(pos1), na.with_positions(pos2), co2])
    M.compute_ewald_mc(mc)
    dev = DeviceMonteCarlo(mc)
    lib, h = dev._lib, dev._h
    kinds = [np.ascontiguousarray([k - 1 for k in ids], dtype=np.int32) for ids in mc.ffidx]      # 0: Na, 1: CO2
    state = [(0, pos1), (0, pos2), (1, pco2)]                                                      # device order (flat index ij - 1)

    def remove(ij):          # 1-based like the reference; returns the reference's return value
        moved = C.c_int32(-1)
        _abi.check(lib, lib.ceg_mc_remove(h, ij - 1, C.byref(moved)))
        state[ij - 1] = state[-1]
        state.pop()
        return moved.value + 1

    def add(i, pos):
        out = C.c_int32(-1)
        p = np.ascontiguousarray(pos, dtype=np.float64)
        _abi.check(lib, lib.ceg_mc_insert(h, _abi.i32ptr(kinds[i - 1]), len(kinds[i - 1]), _abi.dptr(p.reshape(-1)), C.byref(out)))
        state.append((i - 1, p))
        return out.value + 1

    def total_sf():
        re, im = np.empty(len(mc.ewald.kfactors)), np.empty(len(mc.ewald.kfactors))
        _abi.check(lib, lib.ceg_mc_get_state(h, None, _abi.dptr(re), _abi.dptr(im)))
        return re + 1j * im

    from oracle import hostlogic as H
    oef = H.adapt_ewald_framework(mc.ewald)

    def mol_sf(i, p):        # the oracle's power-table structure factor of one molecule (ewald.jl:352-366,660-684)
        re, im = H.molecule_sums(oef, p, [mc.charges[ix] for ix in mc.ffidx[i]])
        return re + 1j * im

    def expected_sf():
        return sum(mol_sf(i, p) for i, p in state)

    def check():
        ref = expected_sf()
        assert np.abs(total_sf() - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())

    check()
    assert remove(3) == 3 and add(2, pco2) == 3; check()                      # :81-83
    assert remove(2) == 3 and add(1, pos2) == 3; check()                      # :85-87
    assert remove(2) == 3 and add(2, pco2) == 3; check()                      # :91-93
    assert remove(2) == 3 and remove(1) == 2                                  # :95-96
    assert add(1, pos2) == 2 and add(1, pos1) == 3; check()                   # :97-99
    # move_one_system!(ctx, 3, pos3) (:103) = trial + accept on the device state
    _abi.check(lib, lib.ceg_mc_accept(h, 2, _abi.dptr(np.ascontiguousarray(pos3).reshape(-1))))
    state[2] = (state[2][0], pos3)
    check()
    assert remove(3) == 3 and add(1, pos3) == 3; check()                      # :105-107
    assert add(2, co2_2) == 4; check()                                        # :112
    # the same species listed in another order give the same energy (:113-120): the structure factor is a plain sum
    other = sum(mol_sf(i, p) for i, p in ((1, co2_2), (1, pco2), (0, pos3), (0, pos2)))
    assert np.abs(total_sf() - other).max() <= 1e-10 * max(1.0, np.abs(other).max())
    dev.close()
