"""Analytic checks of the CPU oracle (independent of the reference's numbers): the radial
factors really are successive (1/r d/dr) derivatives of the pair energy, the min-image
routine returns true nearest images inside its safe radius, and the edge semantics
(clamp, Inf, NaN, stale image vector) are the ones the Julia source spells."""
import math

import mpmath as mp
import numpy as np
import pytest

from ceg_hip import _abi
from ceg_hip.hostmirror.utils import mat_from_parameters, prepare_periodic_distance_computations

mp.mp.dps = 40


def _rules(*specs):
    t = np.zeros(len(specs), dtype=_abi.RULE_DTYPE)
    for i, (kind, p, shift) in enumerate(specs):
        t[i]["kind"] = kind
        t[i]["p"][:len(p)] = p
        t[i]["shift"] = shift
    return t


def _radial_chain(f, r):
    """value, f'/r, (f'/r)'/r, ((f'/r)'/r)'/r at r (mpmath)."""
    g1 = lambda x: mp.diff(f, x) / x
    g2 = lambda x: mp.diff(g1, x) / x
    g3 = lambda x: mp.diff(g2, x) / x
    return [f(r), g1(r), g2(r), g3(r)]


@pytest.mark.parametrize("r", [1.7, 2.9, 3.535, 6.0, 11.9])
def test_lennard_jones_derivatives(oracle, r):
    """interactions.jl:434-441.  NOTE the reference's third factor is 384 eps x6 (5 - 28 x6) / r^8
    (`/(r4*r4)`, :440) whereas ((1/r) d/dr)^3 of the LJ energy has r^6 in the denominator; the
    Buckingham branch (:457) is consistent.  Parity target is the reference as written, so the
    oracle (and the kernels) keep r^8; this test pins exactly that deviation."""
    eps, sig, shift = 107.69, 3.15, -0.1408864
    f = lambda x: 4 * eps * ((sig / x) ** 12 - (sig / x) ** 6) - shift
    want = _radial_chain(f, mp.mpf(r))
    want[3] = want[3] / mp.mpf(r) ** 2
    got = oracle.derivatives_grid(_rules((3, [eps, sig], shift)), r * r)
    for g, w in zip(got, want):
        assert g == pytest.approx(float(w), rel=1e-11)


@pytest.mark.parametrize("r", [1.6, 2.3, 4.0, 9.5])
def test_buckingham_derivatives(oracle, r):
    """interactions.jl:447-457"""
    A, B, C = 5.581e7, 3.985, 9.167e5
    f = lambda x: A * mp.exp(-B * x) - C / x ** 6
    want = _radial_chain(f, mp.mpf(r))
    got = oracle.derivatives_grid(_rules((4, [A, B, C], 0.0)), r * r)
    for g, w in zip(got, want):
        assert g == pytest.approx(float(w), rel=1e-10)


@pytest.mark.parametrize("r", [1.01, 2.0, 5.5, 11.99])
def test_ewald_derivatives(oracle, r):
    """ewald.jl:299-312"""
    alpha, q = 0.26505830360350674, -1.1427
    f = lambda x: q * mp.erfc(alpha * x) / x
    want = _radial_chain(f, mp.mpf(r))
    got = oracle.derivatives_ewald(alpha, q, r * r)
    for g, w in zip(got, want):
        assert g == pytest.approx(float(w), rel=1e-11)


def test_rule_sum_hard_sphere_and_zero_kinds(oracle):
    """interactions.jl:444-446,458-461,599-610"""
    rules = _rules((0, [1.5, 0.0], 0.0), (1, [0.265, 0.9, -0.9], 0.0), (4, [5.581e7, 3.985, 9.167e5], 0.0))
    inside = oracle.derivatives_grid(rules, 1.4 ** 2)
    assert inside[0] == math.inf and np.all(np.isfinite(inside[1:]))
    outside = oracle.derivatives_grid(rules, 1.6 ** 2)
    alone = oracle.derivatives_grid(rules[2:], 1.6 ** 2)
    np.testing.assert_array_equal(outside, alone)
    np.testing.assert_array_equal(oracle.derivatives_grid(_rules((8, [], 0.0)), 4.0), np.zeros(4))
    np.testing.assert_array_equal(oracle.derivatives_grid(_rules((1, [0.2, 1, 1], 5.0)), 4.0), np.zeros(4))   # shift not applied
    for bad in (2, 5, 6, 7):
        with pytest.raises(RuntimeError):
            oracle.derivatives_grid(_rules((bad, [1.0, 1.0], 0.0)), 4.0)


def test_lj_at_zero_distance(oracle):
    """r2 = 0: value +Inf, factors -Inf/+Inf/-Inf (then -Inf*0 = NaN in the accumulation)."""
    got = oracle.derivatives_grid(_rules((3, [100.0, 3.0], 0.0)), 0.0)
    assert got[0] == math.inf and got[1] == -math.inf and got[2] == math.inf and got[3] == -math.inf


# ------------------------------------------------------------------ min-image
def _true_min_image(d, mat, reach=3):
    best, vec = math.inf, None
    for a in range(-reach, reach + 1):
        for b in range(-reach, reach + 1):
            for c in range(-reach, reach + 1):
                v = d + mat @ np.array([a, b, c], dtype=float)
                n = float(v @ v)
                if n < best:
                    best, vec = n, v
    return best, vec


@pytest.mark.parametrize("angles", [(90, 90, 90), (94.07, 94.07, 94.07), (92.82, 107.2, 103.26), (75.0, 110.0, 60.0)])
def test_min_image_inside_safe_radius(oracle, angles):
    """Whenever the routine answers within safemin the result must be the true nearest image
    (utils.jl:233); beyond it the answer is one of the 7 images it may inspect."""
    mat = mat_from_parameters((26.0, 28.0, 31.0), angles)
    inv = np.linalg.inv(mat)
    ortho, safemin = prepare_periodic_distance_computations(mat)
    rng = np.random.default_rng(5)
    hits = 0
    for _ in range(400):
        d = mat @ rng.uniform(-2, 2, 3)
        d2, vec = oracle.periodic_distance2(d, mat, inv, ortho, safemin ** 2)
        f = inv @ vec
        assert np.allclose(f - inv @ d, np.round(f - inv @ d), atol=1e-9)          # a lattice image of d
        if d2 <= safemin ** 2 and not ortho:
            best, _ = _true_min_image(d, mat)
            assert d2 == pytest.approx(best, rel=1e-12)
            assert float(vec @ vec) == pytest.approx(d2, rel=1e-12)
            hits += 1
    assert ortho or hits > 50


def test_min_image_stale_vector_quirk(oracle):
    """utils.jl:234-245: when the search finds nothing closer, d2 is that of the wrapped image but
    `buffer` is left at (wrapped - c).  Reproduced bug-for-bug."""
    mat = mat_from_parameters((30.0, 30.0, 30.0), (60.0, 60.0, 60.0))            # strongly skewed: small safemin
    inv = np.linalg.inv(mat)
    ortho, safemin = prepare_periodic_distance_computations(mat)
    assert not ortho
    rng = np.random.default_rng(11)
    seen = 0
    for _ in range(2000):
        f = rng.uniform(-0.5, 0.5, 3)
        d = mat @ f
        if float(d @ d) <= safemin ** 2:
            continue
        best, _ = _true_min_image(d, mat)
        if best < float(d @ d) * (1 - 1e-12):
            continue                                  # a closer image exists; not the fall-through case
        closer = [float((d + s * mat[:, ax]) @ (d + s * mat[:, ax])) < float(d @ d) for ax in range(3) for s in (1, -1)]
        assert not any(closer)
        d2, vec = oracle.periodic_distance2(d, mat, inv, ortho, safemin ** 2)
        assert d2 == pytest.approx(float(d @ d), rel=1e-12)
        np.testing.assert_allclose(vec, d - mat[:, 2], rtol=0, atol=1e-9)
        seen += 1
    assert seen > 20


def test_ortho_shortcut_skips_search(oracle):
    mat = mat_from_parameters((25.0, 25.0, 25.0), (91.5, 91.5, 91.5))
    inv = np.linalg.inv(mat)
    ortho, safemin = prepare_periodic_distance_computations(mat)
    assert ortho
    d = mat @ np.array([0.49, 0.49, -0.49])
    d2, vec = oracle.periodic_distance2(d, mat, inv, ortho, safemin ** 2)
    np.testing.assert_allclose(vec, d, atol=1e-9)                     # wrapped image returned even if not nearest
    assert d2 == pytest.approx(float(d @ d), rel=1e-12)


# ------------------------------------------------------------------ store semantics
def test_set_gridpoint_scaling_and_clamp(oracle):
    """grids.jl:118-135"""
    delta = np.array([0.25, 0.3, 0.35])
    lam, thr = 0.8314, 1.2e7
    raw = np.array([
        [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0],
        [2e7, 3e7, -5e9, 7.0, 1.0, 2.0, 3.0, 4.0],              # clamped
        [math.inf, math.nan, -math.inf, 0.5, 9.0, 9.0, 9.0, 9.0],  # Inf value, NaN passes through clamp
        [math.nan, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0],           # NaN > thr is false: stored as is
        [thr, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0],                # not strictly greater: no clamp
    ])
    out = oracle.set_gridpoints(raw, delta, lam, thr)
    d1, d2, d3 = delta
    want0 = np.array([1.0 * lam, 2 * d1 * lam, 3 * d2 * lam, 4 * d3 * lam, 5 * (d1 * d2) * lam, 6 * (d1 * d3) * lam,
                      7 * (d2 * d3) * lam, 8 * (d1 * d2 * d3) * lam], dtype=np.float32)
    np.testing.assert_array_equal(out[0], want0)
    np.testing.assert_array_equal(out[1], np.array([2 * thr * lam, thr * d1 * lam, -thr * d2 * lam, 7 * d3 * lam, 0, 0, 0, 0], dtype=np.float32))
    assert out[2][0] == np.float32(2 * thr * lam) and np.isnan(out[2][1]) and out[2][2] == np.float32(-thr * d2 * lam)
    assert np.all(out[2][4:] == 0)
    assert np.isnan(out[3][0]) and out[3][1] == np.float32(d1 * lam)
    assert out[4][0] == np.float32(thr * lam) and out[4][4] == np.float32((d1 * d2) * lam)


def test_coulomb_value_is_inf_within_one_angstrom(oracle, forcefield):
    """probes.jl:116"""
    import ceg_hip as ceg
    from ceg_hip.hostmirror.probes import ProbeSystem
    fw = ceg.load_framework_RASPA("CIT-7", "BoulfelfelSholl2021")
    pc = ProbeSystem.build(fw, forcefield)
    alpha = 0.26505830360350674
    atom = pc.positions[17]
    near, far = atom + [0.6, 0.0, 0.0], atom + [0.0, 1.2, 0.0]
    out = oracle.points_coulomb(pc, alpha, np.array([near, far]))
    assert out[0, 0] == math.inf and np.all(np.isfinite(out[0, 1:]))
    assert np.all(np.isfinite(out[1]))
    on = oracle.points_coulomb(pc, alpha, np.array([atom]))
    assert on[0, 0] == math.inf and np.isnan(on[0, 1:4]).any()


def test_abc_to_xyz_evaluation_order(oracle):
    """coordinates.jl:72-76: (i*size)/dims + shift -- the oracle, the host mirror and the formula agree bitwise."""
    import ctypes as C
    import ceg_hip as ceg
    dims = np.array([217, 203, 189], dtype=np.int32)
    size = np.array([32.40512513, 30.4679001, 28.22271121])
    shift = np.array([-4.02812513, -2.16246457, 0.0])
    cs = ceg.GridCoordinatesSetup(None, 0.15, dims, size, shift, size, size / dims)
    pos = np.empty(3)
    dp = C.POINTER(C.c_double)
    for (i, j, k) in ((0, 0, 0), (217, 203, 189), (13, 77, 101), (216, 1, 188)):
        oracle.lib().oracle_abc_to_xyz(dims.ctypes.data_as(C.POINTER(C.c_int32)), size.ctypes.data_as(dp),
                                       shift.ctypes.data_as(dp), i, j, k, pos.ctypes.data_as(dp))
        want = np.array([(i * size[0]) / 217 + shift[0], (j * size[1]) / 203 + shift[1], (k * size[2]) / 189 + shift[2]])
        np.testing.assert_array_equal(pos, want)
        np.testing.assert_array_equal(ceg.abc_to_xyz(cs, i, j, k), want)
