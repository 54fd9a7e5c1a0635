"""The oracle reproduces the committed golden samples bit for bit (guards the checker itself
against accidental edits; the samples were produced by tests/golden/make_golden.py)."""
import math
from pathlib import Path

import numpy as np
import pytest

from ceg_hip import grids as G, workloads as W

GOLDEN = Path(__file__).parent / "golden"
CASES = {"cha_0.5": ("CHA_1.4_3b4eeb96", 0.5), "cha_0.1": ("CHA_1.4_3b4eeb96", 0.1), "cit7_0.15": ("CIT-7", 0.15)}


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_matches_golden_samples(oracle, case):
    fwname, spacing = CASES[case]
    z = np.load(GOLDEN / f"samples_{case}.npz")
    for atom in ("Ar", "Na"):
        w = W.fixture_workload(fwname, atom, spacing)
        np.testing.assert_array_equal(w.cset.dims, z["dims"])
        i, j, k = z["idx"].T
        pts = np.stack([i * w.cset.size[0] / w.cset.dims[0] + w.cset.shift[0],
                        j * w.cset.size[1] / w.cset.dims[1] + w.cset.shift[1],
                        k * w.cset.size[2] / w.cset.dims[2] + w.cset.shift[2]], axis=1)
        np.testing.assert_array_equal(pts, z["points"])
        raw = oracle.points_vdw(w.probe_vdw, pts)
        np.testing.assert_array_equal(raw, z[f"raw_vdw_{atom}"])
        lam, thr = G.vdw_scaling()
        np.testing.assert_array_equal(oracle.set_gridpoints(raw, w.cset.delta, lam, thr), z[f"f32_vdw_{atom}"])
    raw = oracle.points_coulomb(w.probe_coulomb, w.alpha, pts)
    np.testing.assert_array_equal(raw, z["raw_coulomb"])
    lam, thr = G.coulomb_scaling()
    np.testing.assert_array_equal(oracle.set_gridpoints(raw, w.cset.delta, lam, thr), z["f32_coulomb"])
    # the samples do exercise the special values
    assert (z["f32_vdw_Na"][:, 0] == np.float32(2e7)).any() and (z["f32_coulomb"][:, 0] == np.float32(2e7)).any()


def test_oracle_grid_driver_equals_pointwise(oracle):
    """oracle_grid_vdw/coulomb (loop nest + _set_gridpoint!) == pointwise evaluation, and the
    [z,y,x,c] placement is right."""
    w = W.fixture_workload("CIT-7", "Na", 1.5)
    nx, ny, nz = w.cset.npoints
    lam, thr = G.vdw_scaling()
    grid, raw = oracle.grid_vdw(w.probe_vdw, w.cset, lam, thr, want_raw=True)
    assert grid.shape == (8, nx, ny, nz) and not np.isnan(raw).all()
    from ceg_hip import abc_to_xyz
    for (i, j, k) in ((0, 0, 0), (nx - 1, ny - 1, nz - 1), (3, 2, 1)):
        one = oracle.points_vdw(w.probe_vdw, abc_to_xyz(w.cset, i, j, k)[None, :])
        np.testing.assert_array_equal(raw[i, j, k], one[0])
        np.testing.assert_array_equal(grid[:, i, j, k], oracle.set_gridpoints(one, w.cset.delta, lam, thr)[0])
    lamc, thrc = G.coulomb_scaling()
    gc, rawc = oracle.grid_coulomb(w.probe_coulomb, w.alpha, w.cset, lamc, thrc, i_begin=1, i_end=3, want_raw=True)
    assert np.isnan(gc[:, 0]).all() and np.isnan(gc[:, 3:]).all() and not np.isnan(gc[0, 1:3]).any()


def test_oracle_interpolation_equals_host_mirror(oracle):
    """Row f1: the C restatement of interpolate_grid (literal COEFF*X) against the Python mirror
    that tests/test_reference_pins.py pins to the reference's literals."""
    import math
    import ceg_hip as ceg
    w = W.fixture_workload("CIT-7", "Ar", 1.0)
    lam, thr = G.vdw_scaling()
    grid, _ = oracle.grid_vdw(w.probe_vdw, w.cset, lam, thr)
    gk = (grid.astype(np.float64) * ceg.GRID_TO_KELVIN).astype(np.float32)
    rng = np.random.default_rng(0)
    pts = rng.uniform(-20, 40, (300, 3))
    for prec in (math.inf, 1e-6):                 # VdW grid (blocking rule on) / Coulomb-style grid (off)
        eg = G.EnergyGrid(w.cset, (2, 3, 3), prec, True, gk)
        a = oracle.interpolate_points(eg, pts)
        b = np.array([G.interpolate_grid(eg, p) for p in pts])
        assert np.array_equal(a == 1e100, b == 1e100)
        m = a != 1e100
        np.testing.assert_allclose(a[m], b[m], rtol=1e-11, atol=1e-9)
        assert (prec == math.inf) == bool((a == 1e100).any())


def test_oracle_reciprocal_equals_host_mirror(oracle):
    """Row f2: the literal restatement of compute_ewald (power tables by repeated multiplication,
    reference summation order; ewald.jl:109-185, 555-577) against the direct-exponential mirror
    ``ceg_hip.hostmirror.ewald.compute_ewald`` that test_reference_pins pins to the runtests.jl literals."""
    import ceg_hip as ceg
    rng = np.random.default_rng(11)
    for fwname, sc in (("CHA_1.4_3b4eeb96", (1, 1, 1)), ("CIT-7", None)):
        fw = ceg.load_framework_RASPA(fwname, "BoulfelfelSholl2021")
        ef = ceg.initialize_ewald(fw, sc)
        for molname in ("Na", "CO2"):
            mol = ceg.load_molecule_RASPA(molname, "TraPPE", "BoulfelfelSholl2021")
            base = np.asarray(mol.position, dtype=np.float64).reshape(-1, 3)
            pos = rng.uniform(-30, 50, (12, 1, 3)) + base[None]
            got = oracle.reciprocal_energies(ef, mol, pos)
            ref = np.array([ceg.compute_ewald(ef, ((mol.with_positions(p),),)) for p in pos])
            np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-10 * np.abs(ref).max())


def test_oracle_block_masks_equal_host_mirror(oracle, tmp_path):
    """Row f4: literal restatement of the parse_blockfile scan and of BlockFile(::EnergyGrid) against the
    numpy mirror (``ceg_hip.grids.parse_blockfile``, pinned by runtests.jl:269-272 in test_reference_pins)
    on the reference's block file CIT7block (a sphere straddling a periodic boundary of a triclinic cell;
    the other block files of the fixtures are empty) and on a synthetic file with five spheres in CHA."""
    import ceg_hip as ceg
    root = Path(__file__).parent / "golden" / "raspa" / "structures" / "block"
    synth = tmp_path / "five.block"
    synth.write_text("5\n0.05 0.5 0.95 2.5\n0.5 0.5 0.5 4.0\n0.99 0.01 0.5 1.2\n0.3 0.7 0.1 0.9\n0.0 0.0 0.0 3.3\n")
    for fwname, blk, sp in (("CIT7block", root / "CIT7block.block", 0.5), ("CHA_1.4_3b4eeb96", synth, 0.6), ("CIT-7", synth, 0.45)):
        fw = ceg.load_framework_RASPA(fwname, "BoulfelfelSholl2021")
        cset = ceg.GridCoordinatesSetup.from_cell(fw.mat, sp)
        ref = G.parse_blockfile(blk, cset)
        centers, r2 = G.read_block_spheres(blk, cset)
        got = oracle.block_spheres(cset, centers, r2)
        assert ref.block.any() and np.array_equal(got, ref.block)
    # BlockFile(g): synthetic value channel with isolated, edge and corner cells above 5e6
    w = W.fixture_workload("CIT-7", "Ar", 0.8)
    nx, ny, nz = w.cset.npoints
    rng = np.random.default_rng(2)
    grid = np.zeros((8, nx, ny, nz), dtype=np.float32)
    grid[0] = rng.uniform(0, 5.2e6, (nx, ny, nz)).astype(np.float32)
    grid[0, nx - 1, :, :] = 9e6                   # last plane: not a cell origin, must not block anything by itself
    g = G.EnergyGrid(w.cset, (1, 1, 1), math.inf, True, grid)
    got = oracle.block_from_grid(g)
    ref = np.zeros((nx, ny, nz), dtype=bool)
    hot = np.argwhere(grid[0, :nx - 1, :ny - 1, :nz - 1] > 5e6)
    for i, j, k in hot:
        ref[i:i + 2, j:j + 2, k:k + 2] = True
    assert hot.size and np.array_equal(got, ref)
