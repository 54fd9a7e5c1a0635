"""The checker stands on its own feet: everything oracle/ needs besides the per-point sums -- COEFF, the
cell analysis, the EwaldFramework tables, the EwaldContext constants, the per-molecule structure factors and
``movement_energy`` -- is restated under oracle/ (oracle/hostlogic.py, oracle/ceg_oracle_mc.c, oracle/montecarlo.py) and does
not import the product package.  These CPU tests

* pin COEFF to the reference's own literal (src/constants.jl:24-89, committed as data in tests/golden/coeff.json),
* pin the oracle-only chain to the literals of test/runtests.jl (tests/golden/pins.json),
* compare the two independent restatements (oracle/ vs the host mirror in ceg_hip) with each other.
"""
import ast
import json
import math
from pathlib import Path

import numpy as np
import pytest

import ceg_hip as ceg
from ceg_hip import grids as G, workloads as W

ROOT = Path(__file__).resolve().parent.parent
PINS = json.loads((ROOT / "tests" / "golden" / "pins.json").read_text())
FFNAME = "BoulfelfelSholl2021"


def test_oracle_does_not_import_the_product_package():
    """No module under oracle/ imports ceg_hip (AST walk: comments and docstrings may mention it)."""
    for path in sorted((ROOT / "oracle").glob("*.py")):
        tree = ast.parse(path.read_text())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            for n in names:
                assert not n.split(".")[0] == "ceg_hip", f"{path.name} imports {n}"


def test_coeff_equals_the_reference_literal(oracle):
    """Three sources, one matrix: the reference's literal (golden data), the oracle's derivation (tensor product of the 1-D
    Hermite matrix) and the host mirror's derivation (rational inverse of the evaluation matrix)."""
    from oracle import hostlogic as H
    from ceg_hip.hostmirror.constants import tricubic_coeff
    lit = H.reference_coeff_literal()
    assert lit.shape == (64, 64) and np.count_nonzero(lit) == 1000
    assert np.array_equal(H.tricubic_coeff(), lit)
    assert np.array_equal(tricubic_coeff(), lit)
    # the literal is what it claims to be: COEFF @ (Hermite data of a polynomial) returns the polynomial's coefficients
    rng = np.random.default_rng(0)
    a = rng.integers(-5, 6, 64).astype(np.float64)

    def mono(e, order, x):     # d^order/dt^order t^e at x in {0, 1}
        if order > e:
            return 0.0
        c = math.prod(range(e - order + 1, e + 1)) if order else 1
        return float(c) if (x == 1 or e == order) else 0.0

    chans = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
    X = np.zeros(64)
    for ch, (ox, oy, oz) in enumerate(chans):
        for corner in range(8):
            x, y, z = corner & 1, (corner >> 1) & 1, (corner >> 2) & 1
            X[8 * ch + corner] = sum(a[i + 4 * j + 16 * k] * mono(i, ox, x) * mono(j, oy, y) * mono(k, oz, z)
                                     for i in range(4) for j in range(4) for k in range(4))
    assert np.array_equal(lit @ X, a)


def _cells():
    rng = np.random.default_rng(3)
    out = [ceg.load_framework_RASPA(n, FFNAME).mat for n in ("CHA_1.4_3b4eeb96", "CIT-7")]
    from ceg_hip.hostmirror.utils import mat_from_parameters
    # angles around the Float16 / 2 % boundary of `ortho` (88.2 and 91.8 degrees) and exact right angles
    for ang in ((90, 90, 90), (88.19, 90, 90), (88.21, 90, 91.79), (91.81, 90, 90), (91.84, 88.17, 90.03), (90, 90, 120), (60, 60, 60)):
        out.append(mat_from_parameters((11.0, 13.5, 9.25), ang))
    for _ in range(40):
        out.append(mat_from_parameters(rng.uniform(5, 40, 3), rng.uniform(75, 105, 3)))
    return out


def test_prepare_periodic_distance_computations_two_restatements(oracle):
    from oracle import hostlogic as H
    from ceg_hip.hostmirror.utils import prepare_periodic_distance_computations
    seen = set()
    for mat in _cells():
        o1, s1 = H.prepare_periodic_distance_computations(mat)
        o2, s2 = prepare_periodic_distance_computations(mat)
        assert o1 == o2, mat
        assert s1 == pytest.approx(s2, rel=1e-14)
        seen.add(o1)
    assert seen == {True, False}
    # CHA safemin 14.0757679, not ortho (Float16(94.07) = 94.06)
    o, s = H.prepare_periodic_distance_computations(ceg.load_framework_RASPA("CHA_1.4_3b4eeb96", FFNAME).mat)
    assert not o and s == pytest.approx(14.0757679, rel=1e-8)


@pytest.mark.parametrize("fwname,supercell,ks,nk", [("CHA_1.4_3b4eeb96", (1, 1, 1), (8, 8, 8), 1368), ("CIT-7", (2, 3, 3), (7, 9, 8), 1793)])
def test_initialize_ewald_two_restatements(oracle, fwname, supercell, ks, nk):
    """oracle/hostlogic.initialize_ewald (literal power tables in C, the reference's site order) against the host mirror
    (direct exponentials), and both against the numbers SURVEY appendix A derives from the fixtures."""
    from oracle import hostlogic as H
    fw = ceg.load_framework_RASPA(fwname, FFNAME)
    ef = H.initialize_ewald(fw.mat, np.asarray(fw.position, dtype=np.float64).reshape(-1, 3), fw.atomic_charge, supercell)
    assert ef.ks == ks and ef.num_kvecs == nk
    assert ef.alpha == pytest.approx(0.26505830360350674, rel=1e-15)
    mirror = ceg.initialize_ewald(fw, supercell)
    assert np.array_equal(ef.kindices, np.array(mirror.kspace.kindices, dtype=np.int32))
    np.testing.assert_allclose(ef.kfactors, mirror.kfactors, rtol=1e-13)
    assert ef.UIon == pytest.approx(mirror.UIon, rel=1e-12)
    scale = np.abs(mirror.StoreRigidChargeFramework).max()
    assert np.abs((ef.sf_re + 1j * ef.sf_im) - mirror.StoreRigidChargeFramework).max() <= 1e-11 * scale
    assert ef.net_charges_framework == pytest.approx(mirror.net_charges_framework, rel=1e-12, abs=1e-10)   # a neutral framework: summation-order noise
    # the rows tile 0..num_kvecs without gaps and the (0, 0) row starts at i = 1
    idx = 0
    for j, k, i0, i1, r in ef.kindices:
        assert r == idx and i0 == (1 if (j == 0 and k == 0) else 0) and i1 >= i0
        idx += i1 - i0 + 1
    assert idx == nk


def test_empty_framework_ewald(oracle):
    """initialize_ewald(mat) (ewald.jl:291-296): no atoms, structure factor zero."""
    from oracle import hostlogic as H
    mat = ceg.load_framework_RASPA("CIT-7", FFNAME).mat
    ef = H.initialize_ewald(mat, np.empty((0, 3)), np.empty(0), (2, 3, 3))
    assert ef.num_kvecs == 1793 and not ef.sf_re.any() and not ef.sf_im.any() and ef.net_charges_framework == 0.0


def test_reciprocal_ewald_two_co2_through_the_oracle_alone(oracle):
    """runtests.jl:53-56 with NOTHING of the host mirror between the fixture files and the number: EwaldFramework, context
    constants, structure factors and the energy loop all from oracle/ (the CIF / molecule parsers are input adapters)."""
    from oracle import hostlogic as H
    from oracle.montecarlo import OracleMonteCarlo
    pin = PINS["co2_reciprocal"]
    fw = ceg.load_framework_RASPA(pin["framework"], FFNAME)
    ef = H.initialize_ewald(fw.mat, np.asarray(fw.position, dtype=np.float64).reshape(-1, 3), fw.atomic_charge, (1, 1, 1))
    co2 = ceg.load_molecule_RASPA("CO2", "TraPPE", FFNAME)
    q = np.asarray(co2.atomic_charge, dtype=np.float64)
    charges = np.full(4, np.nan)
    charges[1:] = q
    mc = OracleMonteCarlo(ef.mat, 12.0, None, [0], 0, [[1, 2, 3]], charges, [[np.array(p) for p in pin["positions"]]], [], None, ef)
    val = mc.compute_ewald()
    assert val == pytest.approx(pin["value"], rel=pin["rtol"])
    assert val == pytest.approx(pin["value"], rel=1e-8)            # 5e-10: COULOMBIC_CONVERSION_FACTOR (third party) level
    # and the host mirror agrees with the oracle far below that
    mirror = ceg.compute_ewald(ceg.initialize_ewald(fw, (1, 1, 1)), ([co2.with_positions(p) for p in pin["positions"]],))
    assert val == pytest.approx(mirror, rel=1e-12)


def test_context_constants_two_restatements(oracle):
    from oracle import hostlogic as H
    from ceg_hip.hostmirror.ewald import ewald_context_constants
    fw = ceg.load_framework_RASPA("CIT-7", FFNAME)
    mirror = ceg.initialize_ewald(fw, (2, 3, 3))
    ef = H.adapt_ewald_framework(mirror)
    co2 = ceg.load_molecule_RASPA("CO2", "TraPPE", FFNAME)
    na = ceg.load_molecule_RASPA("Na", "TraPPE", FFNAME)
    rng = np.random.default_rng(4)
    co2s = [co2.with_positions(np.asarray(co2.position).reshape(-1, 3) + rng.uniform(0, 20, 3)) for _ in range(3)]
    nas = [na.with_positions(rng.uniform(0, 20, (1, 3))) for _ in range(2)]
    ref = ewald_context_constants(mirror, (nas, co2s))
    got = H.ewald_context_constants(ef, [(na.atomic_charge, nas[0].position, 2), (co2.atomic_charge, co2s[0].position, 3)])
    assert got[0] == pytest.approx(ref[0], rel=1e-13) and got[1] == pytest.approx(ref[1], rel=1e-13)


def _small_mc(oracle, tmp_path, spacing=0.75):
    """Na + 3 CO2 in CIT-7 with REAL (coarse) grids built by the oracle's brute-force loop nest."""
    from ceg_hip.hostmirror import montecarlo as M
    from ceg_hip.hostmirror.utils import find_supercell
    import ceg_hip.hostmirror.montecarlo as MM

    def build(grid_path, syst_framework, ff, gridstep, atom_or_ef, mat, new, cutoff, ngpus=1):
        iscoulomb = isinstance(atom_or_ef, ceg.EwaldFramework)
        if not iscoulomb and not ff.needsvdwgrid(atom_or_ef):
            return G.EnergyGrid.trivial(True)
        w = W.fixture_workload("CIT-7", "Ar" if iscoulomb else atom_or_ef, spacing)
        if iscoulomb:
            lam, thr = G.coulomb_scaling()
            g, _ = oracle.grid_coulomb(w.probe_coulomb, atom_or_ef.alpha, w.cset, lam, thr)
        else:
            lam, thr = G.vdw_scaling()
            g, _ = oracle.grid_vdw(w.probe_vdw, w.cset, lam, thr)
        gk = (g.astype(np.float64) * ceg.GRID_TO_KELVIN).astype(np.float32)        # parse_grid, grids.jl:78
        return G.EnergyGrid(w.cset, tuple(find_supercell(syst_framework.mat, 12.0)), 1e-6 if iscoulomb else math.inf, True, gk)

    saved = MM.retrieve_or_create_grid
    MM.retrieve_or_create_grid = build
    try:
        def mol(name, positions):
            return ceg.load_molecule_RASPA(name, "TraPPE", FFNAME).with_positions(positions)
        na = [[3.019388765467742, 0.8997706038543032, 26.11901621898599]]
        co2 = np.array([[11.93940309885289, 8.48657378465003, 2.135736631609201], [11.10485516124311, 7.710040763525694, 1.991767166323031],
                        [10.27030722363334, 6.933507742401357, 1.84779770103686]])
        shifts = [[0, 0, 0], [-5.6, -0.4, 6.5], [3.0, 9.0, 11.0]]
        return M, M.setup_montecarlo("CIT-7", FFNAME, [mol("Na", na)] + [mol("CO2", co2 + np.array(s)) for s in shifts], gridstep=spacing)
    finally:
        MM.retrieve_or_create_grid = saved


def test_oracle_movement_energy_equals_the_pinned_mirror(oracle, tmp_path):
    """oracle/montecarlo.OracleMonteCarlo (C restatements: literal COEFF*X interpolation, literal pair loop, power-table structure
    factors, the reference's summation order) against the host mirror ceg_hip.hostmirror.montecarlo that test_montecarlo_pins.py pins to
    runtests.jl:186-267 -- over displacements, rotations, insertions and removals, term by term."""
    from oracle.montecarlo import OracleMonteCarlo
    M, mc = _small_mc(oracle, tmp_path)
    base = M.baseline_energy(mc)
    omc = OracleMonteCarlo.from_setup(mc)
    rec = omc.compute_ewald()
    assert rec == pytest.approx(base.reciprocal, rel=1e-11)
    rng = np.random.default_rng(8)

    def check(got, ref, what):
        r = np.array([ref.framework_vdw, ref.framework_direct, ref.inter, ref.reciprocal])
        blocked = np.abs(r) >= 1e90
        assert np.array_equal(np.abs(got) >= 1e90, blocked), what
        assert np.all(np.abs(got[~blocked] - r[~blocked]) <= 1e-10 * np.abs(r[~blocked]) + 1e-8), (what, got, r)

    co2base = mc.positions[1][0] - mc.positions[1][0][1]
    for step in range(60):
        kind = int(rng.integers(2))
        if step % 10 == 4:                                              # insertion
            shape = np.zeros((1, 3)) if kind == 0 else co2base
            trial = mc.mat @ rng.uniform(0, 1, 3) + shape
            check(omc.insertion_energy(kind, trial), M.insertion_energy(mc, kind, trial), ("insert", step))
            assert omc.add(kind, trial) == M.add_molecule(mc, kind, trial)
        elif step % 10 == 9 and len(mc.positions[kind]) > 1:            # removal
            j = int(rng.integers(len(mc.positions[kind])))
            check(omc.movement_energy((kind, j)), M.movement_energy(mc, (kind, j)), ("delete", step))
            assert omc.remove((kind, j)) == M.remove_molecule(mc, (kind, j))
        else:
            j = int(rng.integers(len(mc.positions[kind])))
            cur = mc.positions[kind][j]
            new = cur + (mc.mat @ rng.uniform(-1, 1, 3) if step % 5 == 0 else rng.uniform(-0.4, 0.4, 3))
            check(omc.movement_energy((kind, j)), M.movement_energy(mc, (kind, j)), ("before", step))
            check(omc.movement_energy((kind, j), new), M.movement_energy(mc, (kind, j), new), ("after", step))
            if step % 3:
                omc.update((kind, j), new)
                M.update_mc(mc, (kind, j), new)
    assert np.array_equal(omc.flat_positions(), np.concatenate([p for _i, _j, _ids, p in mc.molecules()]))
    scale = np.abs(mc.sums[:, 0]).max()
    assert np.abs(omc.total_structure_factor() - mc.sums[:, 0]).max() <= 1e-11 * scale
    # the incremental total equals a fresh compute_ewald of the final configuration
    inc = omc.total_structure_factor().copy()
    omc.compute_ewald()
    assert np.abs(omc.total_structure_factor() - inc).max() <= 1e-11 * scale
