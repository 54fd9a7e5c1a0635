"""Row f3 (guest-guest energies) and the consumers of the grids: the reference's Monte-Carlo
energy literals of ``test/runtests.jl:186-267`` (CIT-7, 2x3x3 supercell, triclinic), rebuilt with
the host mirror ``ceg_hip.hostmirror.montecarlo`` on top of grids whose needed corners come from the CPU
oracle.  Each literal exercises the VdW grids (Ar: LJ; Na: Buckingham + hard sphere; CO2: O/C LJ),
the Coulomb grid, reciprocal Ewald for several molecules, guest-guest pair terms and the tail
correction at once; rtol is 1e-3 in the reference, the observed agreement is recorded per assert."""
import math

import numpy as np
import pytest

import ceg_hip as ceg
from ceg_hip import grids as G
from ceg_hip.hostmirror import montecarlo as M
from ceg_hip.hostmirror.probes import ProbeSystem

from test_reference_pins import FFNAME, interpolate_with_oracle


class OracleGrid(G.EnergyGrid):
    """An EnergyGrid whose values are produced on demand (8 corners per interpolation) by the oracle."""


@pytest.fixture()
def oracle_grids(oracle, forcefield, monkeypatch):
    """setup_montecarlo with grids that are never built: each interpolation asks the oracle for its 8 corners."""
    from ceg_hip.hostmirror.utils import find_supercell
    probes = {}

    def fake_retrieve(grid_path, syst_framework, ff, gridstep, atom_or_ef, mat, new, cutoff, ngpus=1):
        iscoulomb = isinstance(atom_or_ef, ceg.EwaldFramework)
        if not iscoulomb and not ff.needsvdwgrid(atom_or_ef):
            return G.EnergyGrid.trivial(True)
        cset = ceg.GridCoordinatesSetup.from_cell(syst_framework.mat, gridstep)
        g = OracleGrid(cset, tuple(find_supercell(syst_framework.mat, 12.0)), 1e-6 if iscoulomb else math.inf, True, None)
        g.fw = syst_framework
        g.key = ("coulomb", atom_or_ef.alpha) if iscoulomb else ("vdw", atom_or_ef)
        return g

    def fake_interpolate(g, point):
        if g.ewald_precision == -math.inf:
            return 0.0
        kind, what = g.key
        pk = (id(g.fw), kind if kind == "coulomb" else what)
        if pk not in probes:
            # (the framework object is kept in the entry: id() of a collected object can be handed out again, and "Mini" would then be
            #  served the ProbeSystem of "MiniRef" -- seen once as a one-off failure of test_one_atom_frameworks_in_supercells)
            probes[pk] = (g.fw, ProbeSystem.build(g.fw, forcefield) if kind == "coulomb" else ProbeSystem.build(g.fw, forcefield, what))
        if kind == "vdw":
            return interpolate_with_oracle(oracle, g.csetup, probes[pk][1], point)
        return interpolate_with_oracle(oracle, g.csetup, probes[pk][1], point, what)

    monkeypatch.setattr(M, "retrieve_or_create_grid", fake_retrieve)
    monkeypatch.setattr(M, "interpolate_grid", fake_interpolate)
    return None


def _mol(name, positions):
    m = ceg.load_molecule_RASPA(name, "TraPPE", FFNAME)
    return m.with_positions(positions)


def test_two_argon_in_cit7(oracle_grids):
    """runtests.jl:192-199"""
    mc = M.setup_montecarlo("CIT-7", FFNAME, [_mol("Ar", [[-7.7365250811304911, 31.5070011601372251, 1.5285305931479920]]),
                                               _mol("Ar", [[10.7586599791867421, -2.3259182727570948, 20.5642722996513001]])])
    base = float(M.baseline_energy(mc))
    mov1 = float(M.movement_energy(mc, (0, 0)))
    mov2 = float(M.movement_energy(mc, (0, 1)))
    assert base == pytest.approx(-1789.77383582, rel=1e-3)
    assert base - mov1 - mov2 == pytest.approx(0.28797384, rel=1e-3)
    assert base == pytest.approx(-1789.77383582, rel=1e-7)                 # 2e-8, like the CIT-7 point pin
    assert base - mov1 - mov2 == pytest.approx(0.28797384, rel=1e-7)


def test_sodium_in_cit7(oracle_grids):
    """runtests.jl:222-241"""
    solo = [[-4.728415488310421, 32.03533696753957, 2.943765448968882]]
    nxt = [[-5.036, 31.876, 3.117]]
    mc = M.setup_montecarlo("CIT-7", FFNAME, [_mol("Na", solo)])
    assert mc.tailcorrection == pytest.approx(-70.44772635984882, rel=1e-8)          # runtests.jl:224 (isapprox default)
    base = float(M.baseline_energy(mc))
    assert base == pytest.approx(-21375.116833457894, rel=1e-3)
    mc2 = M.setup_montecarlo("CIT-7", FFNAME, [_mol("Na", nxt)])
    assert mc2.tailcorrection == mc.tailcorrection
    base2 = float(M.baseline_energy(mc2))
    assert base2 == pytest.approx(-21795.8765195143, rel=1e-3)
    diff = float(M.movement_energy(mc, (0, 0), nxt) - M.movement_energy(mc, (0, 0)))
    assert base2 == pytest.approx(base + diff, rel=1e-8)                               # runtests.jl:234
    # Observed: 1.2e-4 and 3.0e-4 (2.5 K, 6.6 K).  The Ar literals above are met to 2e-8 and the tail
    # correction to 1e-8, so the charged-guest literals are taken to be RASPA2-era numbers that the
    # reference itself only reproduces to its rtol (the difference is position dependent, i.e. it sits in
    # the interpolated Buckingham / Coulomb terms, not in a constant).  Regression values = this repo.
    assert base == pytest.approx(-21375.116833457894, rel=2e-4)
    assert base2 == pytest.approx(-21795.8765195143, rel=4e-4)
    assert base == pytest.approx(-21372.581099, rel=1e-9)
    assert base2 == pytest.approx(-21789.326070, rel=1e-9)
    duo = M.setup_montecarlo("CIT-7", FFNAME, [_mol("Na", [[1.001641978413878, 8.638263743446769, 17.23737576131632]]),
                                                _mol("Na", [[4.163424680441308, 16.12704796355876, 17.20994387427006]])])
    bduo = float(M.baseline_energy(duo))
    assert bduo == pytest.approx(-25957.746610866408, rel=1e-3)
    assert bduo == pytest.approx(-25957.746610866408, rel=2e-4)            # observed 1.0e-4
    assert bduo == pytest.approx(-25955.127617, rel=1e-9)


def test_sodium_and_two_co2_in_cit7(oracle_grids):
    """runtests.jl:243-266"""
    na = [[3.019388765467742, 0.8997706038543032, 26.11901621898599]]
    co2_1 = [[11.93940309885289, 8.48657378465003, 2.135736631609201], [11.10485516124311, 7.710040763525694, 1.991767166323031],
             [10.27030722363334, 6.933507742401357, 1.84779770103686]]
    co2_2 = [[5.491645446274333, 8.057854365959964, 8.669190836544463], [6.335120278303245, 7.462084936052019, 9.172986424179925],
             [7.178595110332157, 6.866315506144074, 9.676782011815387]]
    mc = M.setup_montecarlo("CIT-7", FFNAME, [_mol("Na", na), _mol("CO2", co2_1), _mol("CO2", co2_2)])
    base = float(M.baseline_energy(mc))
    assert base == pytest.approx(-28329.113561030445, rel=1e-3)
    newna = [[0.9, 0.1, 1.5]]
    diff_na = float(M.movement_energy(mc, (0, 0), newna) - M.movement_energy(mc, (0, 0)))
    assert diff_na == pytest.approx(5440.529635958557, rel=1e-3)
    mc_na = M.setup_montecarlo("CIT-7", FFNAME, [_mol("Na", newna), _mol("CO2", co2_1), _mol("CO2", co2_2)])
    assert base + diff_na == pytest.approx(float(M.baseline_energy(mc_na)), rel=1e-8)   # runtests.jl:261
    newco2 = [[1.3, 2.9, 1.149], [1.3, 2.9, 0.0], [1.3, 2.9, -1.149]]
    diff_co2 = float(M.movement_energy(mc, (1, 1), newco2) - M.movement_energy(mc, (1, 1)))
    mc_co2 = M.setup_montecarlo("CIT-7", FFNAME, [_mol("Na", na), _mol("CO2", co2_1), _mol("CO2", newco2)])
    assert base + diff_co2 == pytest.approx(float(M.baseline_energy(mc_co2)), rel=1e-4)  # runtests.jl:266
    assert base == pytest.approx(-28329.113561030445, rel=3e-4)            # observed 2.4e-4
    assert diff_na == pytest.approx(5440.529635958557, rel=2e-5)           # observed 8.8e-6
    assert base == pytest.approx(-28322.179659, rel=1e-9)
    assert diff_na == pytest.approx(5440.481803253, rel=1e-9)


def test_one_atom_frameworks_in_supercells(oracle_grids):
    """runtests.jl:203-221 -- frameworks of one / a few atoms in the CIT-7 cell ("Mini", "Petit": the grids live
    on the unit cell, the ProbeSystem and the Ewald sums on its 2x3x3 supercell, the framework carries a net
    charge) against the same atoms written out as an explicit 1x1x1 supercell ("MiniRef", "PetitRef")."""
    na_mini = [[-1.401612509676063, 14.86235802394228, 15.37932058231622]]
    mini = float(M.baseline_energy(M.setup_montecarlo("Mini", FFNAME, [_mol("Na", na_mini)])))
    miniref = float(M.baseline_energy(M.setup_montecarlo("MiniRef", FFNAME, [_mol("Na", na_mini)])))
    assert mini == pytest.approx(miniref, rel=1e-3)                         # runtests.jl:209
    assert mini == pytest.approx(-248304.58180794, rel=1e-3)                # :210
    na_petit = [[18.77838182689036, 14.73031622108175, 3.669308624409278]]
    petit = float(M.baseline_energy(M.setup_montecarlo("Petit", FFNAME, [_mol("Na", na_petit)])))
    petitref = float(M.baseline_energy(M.setup_montecarlo("PetitRef", FFNAME, [_mol("Na", na_petit)])))
    assert petit == pytest.approx(petitref, rel=1e-4)                       # :217
    assert petit == pytest.approx(262204.85720076, rel=1e-3)                # :218
    # observed: 6.2e-6 / 6.7e-6 from the literals, 1.8e-8 / 5.0e-7 between the two descriptions of the same crystal
    assert mini == pytest.approx(-248304.58180794, rel=2e-5) and petit == pytest.approx(262204.85720076, rel=2e-5)
    assert mini == pytest.approx(miniref, rel=1e-7) and petit == pytest.approx(petitref, rel=2e-6)
    assert mini == pytest.approx(-248306.13377495, rel=1e-9) and petit == pytest.approx(262203.11093656, rel=1e-9)


def test_sodium_in_supercit7_minus_one(oracle_grids):
    """runtests.jl:200-202 -- the CIT-7 2x3x3 supercell written out as one P1 cell with one atom removed
    (1079 atoms, charged framework)."""
    mc = M.setup_montecarlo("SuperCIT7m1", FFNAME, [_mol("Na", [[4.935357501688667, 23.53557287745349, 25.71480449842175]])])
    base = float(M.baseline_energy(mc))
    assert base == pytest.approx(-8996.975999017683, rel=1e-3)
    assert base == pytest.approx(-8996.975999017683, rel=1e-4)             # observed 3.8e-5
    assert base == pytest.approx(-8997.318017862597, rel=1e-9)


def test_tail_correction_species_counts(forcefield):
    """runtests.jl:283-300 -- TailCorrection for 2 Na + 3 CO2 in CIT-7 (2x3x3), built directly and incrementally."""
    fw = ceg.load_framework_RASPA("CIT-7", FFNAME)
    cell = fw.mat * np.array([2.0, 3.0, 3.0])[None, :]
    lam = 2 * math.pi / float(np.linalg.det(cell))
    ffidx = [[forcefield.sdict["Na"]], [forcefield.sdict[a] for a in ("O_co2", "C_co2", "O_co2")]]
    framework_atoms = [0, 720, 0, 0, 360] + [0] * 15                     # runtests.jl:283
    assert len(framework_atoms) == len(forcefield.sdict)
    v0, fwk, cross = M.tail_correction(forcefield, ffidx, framework_atoms, lam, [0, 0])
    assert v0 == 0.0                                                     # :287
    counts = [0, 0]
    val = v0
    for i, num in ((0, 2), (1, 3)):                                      # :288-289
        val += M.modify_species_dryrun(fwk, cross, counts, i, num)
        counts[i] += num
    assert val == pytest.approx(-322.46442841047406, rel=1e-8)           # :290
    v1, _, _ = M.tail_correction(forcefield, ffidx, framework_atoms, lam, [2, 3])
    assert v1 == pytest.approx(val, rel=1e-12)                           # :293
    v2, fwk2, cross2 = M.tail_correction(forcefield, ffidx, framework_atoms, lam, [1, 5])
    counts = [1, 5]
    for i, num in ((1, -1), (0, 1), (1, -1)):                            # :296-298
        v2 += M.modify_species_dryrun(fwk2, cross2, counts, i, num)
        counts[i] += num
    assert v2 == pytest.approx(val, rel=1e-10)                           # :299


def test_restart_co2_in_cha_na(oracle_grids):
    """runtests.jl:452-455 -- one CO2 at the positions of the reference's restart fixture
    (test/CHA_1.4_3b4eeb96_Na_11812.restart, lines 48-50) in the Na-exchanged CHA framework."""
    pos = [[14.901841423007, 23.433903291107, 4.454129422603], [14.148462432033, 23.671252375058, 3.619691582913],
           [13.395083441058, 23.908601459010, 2.785253743223]]
    mc = M.setup_montecarlo("CHA_1.4_3b4eeb96_Na_11812", FFNAME, [_mol("CO2", pos)])
    base = float(M.baseline_energy(mc))
    assert base == pytest.approx(-14128.888030042883, rel=1e-3)
    assert base == pytest.approx(-14128.888030042883, rel=1e-6)            # observed 2.7e-7
    assert base == pytest.approx(-14128.88426348615, rel=1e-9)


def test_montecarlo_setup_testset(oracle_grids):
    """The reference's "MonteCarloSetup" testset, runtests.jl:126-161: which terms of baseline_energy / movement_energy depend on
    an uncharged / a charged guest's position, movement_energy at the current position == with the position given, the
    baseline of the moved system == baseline - before + after, and a kind listed with zero molecules + add_one_system!."""
    co2 = ceg.load_molecule_RASPA("CO2", "TraPPE", FFNAME)
    pos1, pos2 = [[0.0, 2.0, 4.0]], [[1.0, 3.0, 1.0]]
    fw = "CHA_1.4_3b4eeb96"
    ar1 = M.setup_montecarlo(fw, FFNAME, [co2, _mol("Ar", pos1)])
    ar2 = M.setup_montecarlo(fw, FFNAME, [co2, _mol("Ar", pos2)])
    b1, b2 = M.baseline_energy(ar1), M.baseline_energy(ar2)
    mov1 = M.movement_energy(ar1, (1, 0))
    assert b1.reciprocal == b2.reciprocal and b1.framework_direct == b2.framework_direct             # :140-141
    assert b1.inter != b2.inter and b1.framework_vdw != b2.framework_vdw                             # :142-143
    assert mov1.reciprocal == 0.0 and mov1.framework_direct == 0.0                                   # :144
    again = M.movement_energy(ar1, (1, 0), pos1)
    assert (mov1.framework_vdw, mov1.framework_direct, mov1.inter, mov1.reciprocal) == \
           (again.framework_vdw, again.framework_direct, again.inter, again.reciprocal)               # :145
    assert float(b2) == pytest.approx(float(b1) - float(mov1) + float(M.movement_energy(ar1, (1, 0), pos2)), rel=1e-9)   # :146

    na1 = M.setup_montecarlo(fw, FFNAME, [co2, _mol("Na", pos1)])
    na2 = M.setup_montecarlo(fw, FFNAME, [co2, _mol("Na", pos2)])
    bn1 = M.baseline_energy(na1)
    m11, m12 = M.movement_energy(na1, (1, 0)), M.movement_energy(na1, (1, 0), pos2)
    a = M.movement_energy(na1, (1, 0), pos1)
    assert float(m11) == pytest.approx(float(a), rel=1e-12, abs=1e-9)                                # :152 (== in the reference)
    assert float(M.baseline_energy(na2)) == pytest.approx(float(bn1) - float(m11) + float(m12), rel=1e-9)          # :153

    # handling of empty systems (:155-160)
    empty = M.setup_montecarlo("CIT-7", FFNAME, [(co2, 0)])
    assert float(M.baseline_energy(empty)) == 0.0
    one = M.setup_montecarlo("CIT-7", FFNAME, [co2.with_positions(empty.models[0])])
    assert M.add_molecule(empty, 0) == 0
    be, bo = M.baseline_energy(empty), M.baseline_energy(one)
    for name in ("framework_vdw", "framework_direct", "inter", "reciprocal", "tailcorrection"):
        assert getattr(be, name) == pytest.approx(getattr(bo, name), rel=1e-12, abs=1e-9), name


def test_deletion_and_addition(oracle_grids):
    """The reference's "Deletion and addition" testset, runtests.jl:302-352: remove_one_system! / add_one_system! / update_mc! /
    movement_energy for every order in which the species can be listed -- in particular the INDEX semantics of a removal (the last
    molecule of the kind takes the freed index, montecarlo.jl:798-808) and removal before the Ewald state exists."""
    import copy
    co2_1 = [[4.221014336721, 2.235273775272, 5.118873160667], [4.915895823098, 3.103469549355, 4.829776606287],
             [5.610777309474, 3.971665323438, 4.540680051907]]
    co2_2 = [[13.732840754800, 5.744459327798, 8.119705823930], [13.418141549103, 6.398573841906, 7.229032139371],
             [13.103442343406, 7.052688356015, 6.338358454812]]
    co2_3 = [[21.473519181736, 19.719777629492, 13.809504713725], [21.888511731126, 20.756701257738, 13.539742646090],
             [22.303504280517, 21.793624885985, 13.269980578454]]
    pos_x = [[5.488965064161, 14.335087694715, 16.087580364999], [4.887655471102, 13.508430918264, 15.562922050243],
             [4.286345878043, 12.681774141812, 15.038263735487]]
    fw = "CHA_1.4_3b4eeb96_Na_11812"
    ar = ceg.load_molecule_RASPA("Ar", "TraPPE", FFNAME)
    m1, m2, m3 = _mol("CO2", co2_1), _mol("CO2", co2_2), _mol("CO2", co2_3)
    pos1, pos2 = np.array(co2_1), np.array(co2_2)

    def same(a, b):
        return float(a) == pytest.approx(float(b), rel=1e-9, abs=1e-7)

    ref = M.setup_montecarlo(fw, FFNAME, [m1, m3, ar])
    base_ref = M.baseline_energy(ref)
    move_ref = M.movement_energy(ref, (0, 0), pos2)
    M.update_mc(ref, (0, 0), pos2)
    othermove_ref = M.movement_energy(ref, (0, 1), pos_x)
    # (species as listed, kind of the CO2 0-based, index of the CO2 that is removed first 0-based)
    for mols, i, j in (([ar, m1, m2, m3], 1, 1), ([ar, m2, m1, m3], 1, 0), ([m1, m2, m3, ar], 0, 1), ([m2, m1, m3, ar], 0, 0)):
        other = 1 - j
        mc_a = M.setup_montecarlo(fw, FFNAME, mols)
        mc_b = copy.deepcopy(mc_a)
        M.remove_molecule(mc_a, (i, j))                       # before any baseline_energy: no Ewald state yet
        assert same(M.baseline_energy(mc_a), base_ref)
        assert same(M.movement_energy(mc_a, (i, other), pos2), move_ref)
        M.update_mc(mc_a, (i, other), pos2)
        assert same(M.movement_energy(mc_a, (i, j), pos_x), othermove_ref)
        assert M.add_molecule(mc_a, i, pos1) == 2             # `== 3` in the reference's 1-based count
        M.remove_molecule(mc_a, (i, other))
        assert same(M.movement_energy(mc_a, (i, other), pos2), move_ref)

        M.baseline_energy(mc_b)
        M.remove_molecule(mc_b, (i, j))
        assert same(M.movement_energy(mc_b, (i, other), pos2), move_ref)
        M.update_mc(mc_b, (i, other), pos2)
        assert same(M.movement_energy(mc_b, (i, j), pos_x), othermove_ref)
        assert M.add_molecule(mc_b, i, pos1) == 2
        M.remove_molecule(mc_b, (i, other))
        assert same(M.movement_energy(mc_b, (i, other), pos2), move_ref)


def _trio(M_):
    na = [[3.019388765467742, 0.8997706038543032, 26.11901621898599]]
    co2_1 = [[11.93940309885289, 8.48657378465003, 2.135736631609201], [11.10485516124311, 7.710040763525694, 1.991767166323031],
             [10.27030722363334, 6.933507742401357, 1.84779770103686]]
    co2_2 = [[5.491645446274333, 8.057854365959964, 8.669190836544463], [6.335120278303245, 7.462084936052019, 9.172986424179925],
             [7.178595110332157, 6.866315506144074, 9.676782011815387]]
    return M_.setup_montecarlo("CIT-7", FFNAME, [_mol("Na", na), _mol("CO2", co2_1), _mol("CO2", co2_2)])


def test_oracle_pair_energy_equals_host_mirror(oracle, monkeypatch):
    """Row f3 oracle (C restatement of single_contribution_vdw_noneighbour) against the Python mirror
    that the literals above pin; includes trial positions that wrap around the MC cell and a close
    contact."""
    monkeypatch.setattr(M, "retrieve_or_create_grid", lambda *a, **k: G.EnergyGrid.trivial(True))
    mc = _trio(M)
    rng = np.random.default_rng(3)
    for idx in ((0, 0), (1, 0), (1, 1)):
        base = mc.positions[idx[0]][idx[1]]
        trial = base[None] + np.concatenate([rng.uniform(-4, 4, (6, 1, 3)), rng.uniform(-80, 80, (6, 1, 3))])
        got = oracle.single_contribution_vdw(mc, idx, trial)
        ref = np.array([M.single_contribution_vdw(mc, idx, t) for t in trial])
        np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)          # numpy matvec vs scalar sums: last-digit differences
    # Na dropped 0.5 A from an oxygen of the first CO2: steep repulsive wall (6.5e10 K)
    contact = mc.positions[1][0][0][None, None, :] + 0.3
    assert oracle.single_contribution_vdw(mc, (0, 0), contact)[0] == pytest.approx(M.single_contribution_vdw(mc, (0, 0), contact[0]), rel=1e-10)


@pytest.mark.gpu
def test_gpu_montecarlo_energies(hip_lib, oracle, tmp_path):
    """The same literals through the product path: CIT-7 grids (Ar, Na, O_co2, C_co2 VdW + Coulomb, 0.15 A,
    2x3x3 supercell) built by the HIP kernels via setup_montecarlo, then movement energies of many trial
    placements from the GPU consumers (interpolation, pair kernel, reciprocal kernel) against the host
    mirror / oracle, and the runtests.jl literals from GPU-evaluated baselines."""
    import os
    from pathlib import Path
    from ceg_hip.energy import GpuMonteCarloEnergy
    golden = Path(__file__).parent / "golden" / "raspa"
    raspa = tmp_path / "raspa"
    raspa.mkdir()
    for sub in ("forcefield", "molecules", "structures"):
        os.symlink(golden / sub, raspa / sub)
    ceg.setdir_RASPA(raspa)
    try:
        mc = _trio(M)
        gm = GpuMonteCarloEnergy(mc)
        base = M.baseline_energy(mc)                       # host: sets mc.sums
        gbase = gm.baseline_energy()
        assert float(gbase) == pytest.approx(float(base), rel=1e-10)
        assert float(gbase) == pytest.approx(-28329.113561030445, rel=1e-3)           # runtests.jl:258
        assert float(gbase) == pytest.approx(-28322.179659, rel=1e-7)                 # the CPU-oracle value of this repo
        rng = np.random.default_rng(9)
        for idx in ((0, 0), (1, 0), (1, 1)):
            cur = mc.positions[idx[0]][idx[1]]
            trial = cur[None] + np.concatenate([np.zeros((1, 1, 3)), rng.uniform(-2, 2, (40, 1, 3)), rng.uniform(-60, 60, (23, 1, 3))])
            got = gm.movement_energies(idx, trial)
            pair_ref = oracle.single_contribution_vdw(mc, idx, trial)
            fin = np.isfinite(pair_ref)
            assert np.array_equal(np.isfinite(got[:, 2]), fin)
            assert np.all(np.abs(got[fin, 2] - pair_ref[fin]) <= 1e-10 * np.abs(pair_ref[fin]) + 1e-9)
            for t in (0, 1, 41):
                ref = M.movement_energy(mc, idx, trial[t])
                r = np.array([ref.framework_vdw, ref.framework_direct, ref.inter, ref.reciprocal])
                ok = np.isfinite(r)
                assert np.all(np.abs(got[t][ok] - r[ok]) <= 1e-9 * np.abs(r[ok]) + 1e-7), (idx, t, got[t], r)
        # runtests.jl:259-261: moving the Na
        d = gm.movement_energies((0, 0), np.array([[[0.9, 0.1, 1.5]], mc.positions[0][0]]))
        diff_na = d[0].sum() - d[1].sum()
        assert diff_na == pytest.approx(5440.529635958557, rel=1e-3)
        assert diff_na == pytest.approx(5440.481803253, rel=1e-7)
        gm.close()
        # two argon atoms (runtests.jl:192-199)
        mc2 = M.setup_montecarlo("CIT-7", FFNAME, [_mol("Ar", [[-7.7365250811304911, 31.5070011601372251, 1.5285305931479920]]),
                                                    _mol("Ar", [[10.7586599791867421, -2.3259182727570948, 20.5642722996513001]])])
        gm2 = GpuMonteCarloEnergy(mc2)
        M.baseline_energy(mc2)
        b2 = float(gm2.baseline_energy())
        assert b2 == pytest.approx(-1789.77383582, rel=1e-7)
        gm2.close()
        # one-atom framework on its 2x3x3 supercell vs the explicit supercell (runtests.jl:203-211)
        na_mini = [[-1.401612509676063, 14.86235802394228, 15.37932058231622]]
        vals = []
        for fwname in ("Mini", "MiniRef"):
            mc3 = M.setup_montecarlo(fwname, FFNAME, [_mol("Na", na_mini)])
            gm3 = GpuMonteCarloEnergy(mc3)
            M.baseline_energy(mc3)
            vals.append(float(gm3.baseline_energy()))
            gm3.close()
        assert vals[0] == pytest.approx(-248304.58180794, rel=1e-3) and vals[0] == pytest.approx(vals[1], rel=1e-3)
        assert vals[0] == pytest.approx(-248306.13377495, rel=1e-7) and vals[1] == pytest.approx(-248306.13817777, rel=1e-7)
    finally:
        ceg.setdir_RASPA(golden)


def test_incremental_ewald_sums_of_the_mirror(monkeypatch):
    """update_mc / add_molecule / remove_molecule of the host mirror (update_ewald_context!, add_one_system!, remove_one_system!:
    ewald.jl:757-810) keep `sums` equal to what compute_ewald(::IncrementalEwaldContext) (ewald.jl:630-652) rebuilds from scratch,
    and single_contribution_ewald differences equal the change of the full reciprocal energy (the relation runtests.jl:234,261 rely on)."""
    monkeypatch.setattr(M, "retrieve_or_create_grid", lambda *a, **k: G.EnergyGrid.trivial(True))
    mc = _trio(M)
    M.compute_ewald_mc(mc)
    rng = np.random.default_rng(5)
    for step in range(40):
        kind = int(rng.integers(2))
        if step % 5 == 0:
            shape = mc.positions[kind][0] - mc.positions[kind][0][0] if mc.positions[kind] else np.zeros((len(mc.ffidx[kind]), 3))
            M.add_molecule(mc, kind, rng.uniform(0, 25, 3) + shape)
        elif step % 5 == 3 and len(mc.positions[kind]) > 1:
            M.remove_molecule(mc, (kind, int(rng.integers(len(mc.positions[kind])))))
        elif mc.positions[kind]:
            j = int(rng.integers(len(mc.positions[kind])))
            new = mc.positions[kind][j] + rng.uniform(-1, 1, 3)
            e0 = M.compute_ewald_mc(_copy_mc(mc))
            d = M.single_contribution_ewald(mc, (kind, j), new) - M.single_contribution_ewald(mc, (kind, j))
            M.update_mc(mc, (kind, j), new)
            e1 = M.compute_ewald_mc(_copy_mc(mc))
            assert e1 - e0 == pytest.approx(d, rel=1e-9, abs=1e-7)
        ref = _copy_mc(mc)
        M.compute_ewald_mc(ref)
        assert ref.sums.shape == mc.sums.shape
        np.testing.assert_allclose(mc.sums, ref.sums, rtol=0, atol=1e-10 * max(1.0, np.abs(ref.sums).max()))


def _copy_mc(mc):
    import copy
    c = copy.copy(mc)
    c.positions = [[p.copy() for p in kind] for kind in mc.positions]
    c.sums = None
    return c
