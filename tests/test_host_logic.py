"""Host-side logic around the hot path (no GPU): ABI exports, geometry rules, force-field
flattening, ``.grid`` file format, interpolation, error behaviour."""
import ctypes
import math
import re
import struct
from pathlib import Path

import numpy as np
import pytest

import ceg_hip as ceg
from ceg_hip import _abi, grids as G, workloads as W
from ceg_hip.hostmirror.constants import tricubic_coeff
from ceg_hip.hostmirror.interactions import FF, InteractionRule, InteractionRuleSum, make_rule
from ceg_hip.hostmirror.probes import ProbeSystem
from ceg_hip.hostmirror.utils import find_supercell, get_atom_name, mat_from_parameters, prepare_periodic_distance_computations

ROOT = Path(__file__).resolve().parent.parent
FFNAME = "BoulfelfelSholl2021"


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    """include/ceg_hip.h <-> libceg_hip.so <-> ctypes prototypes (no compute call)."""
    header = (ROOT / "include" / "ceg_hip.h").read_text()
    declared = set(re.findall(r"CEG_API\s+[\w\s\*]+?\b(ceg_\w+)\s*\(", header))
    assert declared == set(_abi.PROTOTYPES), declared ^ set(_abi.PROTOTYPES)
    lib = _abi.load_library()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.ceg_abi_version() == 1
    assert lib.ceg_last_error() is not None


def test_rule_struct_layout():
    assert _abi.RULE_DTYPE.itemsize == 40
    assert _abi.RULE_DTYPE.fields["p"][1] == 8 and _abi.RULE_DTYPE.fields["shift"][1] == 32


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU the build entry points must fail loudly (this container has none)."""
    lib = _abi.load_library()
    if lib.ceg_device_count() > 0:
        pytest.skip("a HIP device is visible")
    w = W.fixture_workload("CIT-7", "Ar", 2.0)
    with pytest.raises(_abi.CegError) as ei:
        G.build_vdw_array(w.probe_vdw, w.cset)
    assert ei.value.code == -2 and "no HIP device" in str(ei.value)
    with pytest.raises(_abi.CegError):
        G.build_coulomb_array(w.probe_coulomb, w.alpha, w.cset)
    # the device-resident variants: NULL output rejected before anything else, then the same "no device" error
    dims, size, shift, delta = G._grid_args(w.cset)
    pos = np.ascontiguousarray(w.probe_coulomb.positions, dtype=np.float64)
    q = np.ascontiguousarray(w.probe_coulomb.charges, dtype=np.float64)
    mat, inv = G._matT(w.probe_coulomb.mat), G._matT(w.probe_coulomb.invmat)
    common = (_abi.dptr(pos), _abi.dptr(q), len(q), _abi.dptr(mat), _abi.dptr(inv), 0, 1e9, 144.0, w.alpha,
              _abi.i32ptr(dims), _abi.dptr(size), _abi.dptr(shift), _abi.dptr(delta), 1.0, 1e7)
    assert lib.ceg_grid_coulomb_device(*common, None, 0, 1) == -1 and b"d_grid" in lib.ceg_last_error()
    assert lib.ceg_grid_coulomb_device(*common, 4096, 0, 1) == -2 and b"no HIP device" in lib.ceg_last_error()


def test_consumer_entry_points_reject_bad_arguments_and_have_no_cpu_path():
    """The f1-f4 entry points validate their arguments before touching a device (CEG_ERR_INVALID /
    CEG_ERR_UNSUPPORTED, message through ceg_last_error) and, with valid arguments on a box without a
    GPU, fail with CEG_ERR_NO_DEVICE -- never a CPU result."""
    import ctypes as C
    lib = _abi.load_library()
    h = C.c_void_p()
    ks = np.array([2, 2, 2], dtype=np.int32)
    inv = np.eye(3).reshape(-1) / 30.0
    ijk = np.array([[1, 0, 0], [0, 1, -2]], dtype=np.int32)
    kf = np.ones(2); re_ = np.zeros(2); im_ = np.zeros(2)
    # k-vector outside the (kx, ky, kz) box
    bad = np.array([[3, 0, 0], [0, 1, -2]], dtype=np.int32)
    assert lib.ceg_recip_create(C.byref(h), 0, _abi.i32ptr(bad.reshape(-1)), _abi.dptr(kf), _abi.dptr(re_), _abi.dptr(im_), 2,
                                _abi.i32ptr(ks), _abi.dptr(inv)) == -1
    assert b"k-vector" in lib.ceg_last_error()
    # k-space box larger than the LDS tables
    big = np.array([200, 200, 200], dtype=np.int32)
    assert lib.ceg_recip_create(C.byref(h), 0, _abi.i32ptr(ijk.reshape(-1)), _abi.dptr(kf), _abi.dptr(re_), _abi.dptr(im_), 2,
                                _abi.i32ptr(big), _abi.dptr(inv)) == -5
    # pair table: decreasing offsets, undefined interaction
    rules = np.zeros(1, dtype=_abi.RULE_DTYPE)
    rules[0]["kind"] = 7
    mat = np.eye(3).reshape(-1) * 30.0
    off_bad = np.array([0, 1, 0, 1, 1], dtype=np.int32)
    assert lib.ceg_pairs_create(C.byref(h), 0, _abi.dptr(mat), _abi.dptr(inv), 144.0, rules.ctypes.data, _abi.i32ptr(off_bad), 2,
                                1.0) == -1
    off = np.array([0, 1, 1, 1, 1], dtype=np.int32)
    assert lib.ceg_pairs_create(C.byref(h), 0, _abi.dptr(mat), _abi.dptr(inv), 144.0, rules.ctypes.data, _abi.i32ptr(off), 2,
                                1.0) == -4
    assert b"Undefined" in lib.ceg_last_error()
    # Monte-Carlo state: undefined interaction in the pair table, k-space tables announced but missing, calls on a NULL handle
    charge2 = np.zeros(2)
    assert lib.ceg_mc_create(C.byref(h), 0, None, None, _abi.dptr(charge2), 2, _abi.dptr(mat), _abi.dptr(inv), 144.0, rules.ctypes.data,
                             _abi.i32ptr(off), 1.0, None, None, None, None, 0, None, None) == -4
    assert lib.ceg_mc_create(C.byref(h), 0, None, None, _abi.dptr(charge2), 2, _abi.dptr(mat), _abi.dptr(inv), 144.0, rules.ctypes.data,
                             _abi.i32ptr(off), 1.0, None, None, None, None, 5, None, None) == -1
    assert lib.ceg_mc_trial(None, 0, None, 0, None) == -1 and lib.ceg_mc_accept(None, 0, None) == -1
    assert lib.ceg_mc_insert(None, None, 0, None, None) == -1 and lib.ceg_mc_remove(None, 0, None) == -1
    assert lib.ceg_mc_destroy(None) == 0
    # blocking masks: null pointers / empty dims
    dims = np.array([3, 3, 3], dtype=np.int32)
    assert lib.ceg_block_from_grid(0, None, 0, _abi.i32ptr(dims), 5e6, None) == -1
    assert lib.ceg_release_cached_buffers() == 0
    if lib.ceg_device_count() > 0:
        return
    # valid arguments, no device: loud failure
    rc = lib.ceg_recip_create(C.byref(h), 0, _abi.i32ptr(ijk.reshape(-1)), _abi.dptr(kf), _abi.dptr(re_), _abi.dptr(im_), 2,
                              _abi.i32ptr(ks), _abi.dptr(inv))
    assert rc == -2 and b"no HIP device" in lib.ceg_last_error() and not h.value
    rules[0]["kind"] = 3
    assert lib.ceg_pairs_create(C.byref(h), 0, _abi.dptr(mat), _abi.dptr(inv), 144.0, rules.ctypes.data, _abi.i32ptr(off), 2, 1.0) == -2
    value = np.zeros((4, 4, 4), dtype=np.float32)
    out = np.zeros((4, 4, 4), dtype=np.uint8)
    assert lib.ceg_block_from_grid(0, value.ctypes.data, 0, _abi.i32ptr(dims), 5e6, out.ctypes.data) == -2
    # device-resident Monte-Carlo state: no device, no state (and never a CPU evaluation)
    charge = np.zeros(2)
    assert lib.ceg_mc_create(C.byref(h), 0, None, None, _abi.dptr(charge), 2, _abi.dptr(mat), _abi.dptr(inv), 144.0, rules.ctypes.data,
                             _abi.i32ptr(off), 1.0, None, None, None, None, 0, None, None) == -2
    assert b"no HIP device" in lib.ceg_last_error() and not h.value


# ------------------------------------------------------------------ geometry
def test_reciprocal_row_layout_host_side():
    """ceg_recip_layout (host only): the k-vectors of the CHA and CIT-7 fixtures regrouped into rows (j, k) x i, cut into segments and
    dealt to 64 lanes.  Every k-vector gets its own slot; the k-vectors of a segment sit in consecutive slots of one lane with
    consecutive i; a round is as long as its longest segment; few slots are padding; a shuffled list gives the same slot counts."""
    import ctypes as C
    lib = _abi.load_library()
    rng = np.random.default_rng(3)
    for name, sc, min_eff in (("CHA_1.4_3b4eeb96", (1, 1, 1), 0.85), ("CIT-7", None, 0.85)):
        fw = ceg.load_framework_RASPA(name, "BoulfelfelSholl2021")
        ef = ceg.initialize_ewald(fw, sc)
        ks = np.asarray(ef.kspace.ks, dtype=np.int32)
        for shuffle in (False, True):
            ijk = np.asarray(ef.kvec_ijk, dtype=np.int32)
            if shuffle:
                ijk = ijk[rng.permutation(len(ijk))]
            ijk = np.ascontiguousarray(ijk)
            nk = len(ijk)
            nr, ns = C.c_int32(), C.c_int32()
            assert lib.ceg_recip_layout(_abi.i32ptr(ijk.reshape(-1)), nk, _abi.i32ptr(ks), C.byref(nr), C.byref(ns), None, None) == 0
            slot = np.empty(nk, dtype=np.int64)
            desc = np.empty(nr.value * 64, dtype=np.int32)
            assert lib.ceg_recip_layout(_abi.i32ptr(ijk.reshape(-1)), nk, _abi.i32ptr(ks), C.byref(nr), C.byref(ns),
                                        slot.ctypes.data, desc.ctypes.data) == 0
            assert len(np.unique(slot)) == nk and slot.min() >= 0 and slot.max() < ns.value * 64
            assert nk / (ns.value * 64) >= min_eff, (name, nk, ns.value)
            L = desc >> 27
            assert np.all(L.reshape(nr.value, 64) == L.reshape(nr.value, 64)[:, :1]) and L.reshape(nr.value, 64)[:, 0].sum() == ns.value
            assert np.all(np.diff(L.reshape(nr.value, 64)[:, 0]) <= 0)          # longest segments first
            first = np.concatenate([[0], np.cumsum(L.reshape(nr.value, 64)[:, 0])])
            # every k-vector: its (round, lane) descriptor names its row, and its slot is the segment start + (i - i0)
            s_idx, lane = slot // 64, slot % 64
            rnd = np.searchsorted(first, s_idx, side="right") - 1
            d = desc.reshape(nr.value, 64)[rnd, lane]
            i0, jj, kk = d & 0x1ff, (d >> 9) & 0x1ff, (d >> 18) & 0x1ff
            assert np.array_equal(jj - ks[1], ijk[:, 1]) and np.array_equal(kk - ks[2], ijk[:, 2])
            assert np.array_equal(s_idx - first[rnd], ijk[:, 0] - i0) and np.all(ijk[:, 0] >= i0)
            if not shuffle:
                counts = (nr.value, ns.value)
            else:
                assert (nr.value, ns.value) == counts
    # validation happens on the host as well
    bad = np.array([[3, 0, 0]], dtype=np.int32)
    nr, ns = C.c_int32(), C.c_int32()
    assert lib.ceg_recip_layout(_abi.i32ptr(bad.reshape(-1)), 1, _abi.i32ptr(np.array([2, 2, 2], dtype=np.int32)), C.byref(nr), C.byref(ns), None, None) == -1
    assert lib.ceg_recip_layout(None, 0, _abi.i32ptr(np.array([2, 2, 2], dtype=np.int32)), C.byref(nr), C.byref(ns), None, None) == 0 and nr.value == 0


def test_grid_coordinates_setup_cha():
    """coordinates.jl:32-41 on the CHA fixture (numbers of SURVEY appendix A)."""
    fw = ceg.load_framework_RASPA("CHA_1.4_3b4eeb96", FFNAME)
    assert len(fw) == 972
    for spacing, dims in ((0.5, (65, 61, 57)), (0.3, (109, 101, 95)), (0.15, (217, 203, 189)), (0.1, (325, 305, 283))):
        cs = ceg.GridCoordinatesSetup.from_cell(fw.mat, spacing)
        assert tuple(cs.dims) == dims and cs.dims.dtype == np.int32
        assert np.all(cs.dims % 2 == 1)
    np.testing.assert_allclose(cs.size, [32.40512513, 30.46790010, 28.22271121], rtol=1e-9)
    np.testing.assert_allclose(cs.shift, [-4.02812513, -2.16246457, 0.0], rtol=1e-8, atol=1e-12)
    np.testing.assert_array_equal(cs.delta, cs.size / cs.dims)
    assert find_supercell(fw.mat, 12.0) == (1, 1, 1)
    ortho, safemin = prepare_periodic_distance_computations(fw.mat)
    assert not ortho and safemin == pytest.approx(14.0757679, rel=1e-8)


def test_ortho_flag_uses_float16_tolerance():
    """utils.jl:148 -- 2 % of 90 degrees evaluated on the Float16-rounded angle."""
    def ortho(angle):
        return prepare_periodic_distance_computations(mat_from_parameters((20.0, 21.0, 22.0), (90.0, 90.0, angle)))[0]
    assert ortho(90.0) and ortho(91.5) and ortho(88.3)
    assert not ortho(91.9) and not ortho(94.07) and not ortho(88.0)


def test_supercell_tiling_order():
    """probes.jl:37-53: index = ix*n*ny*nz + iy*n*nz + iz*n + i."""
    ff = ceg.parse_forcefield_RASPA(FFNAME)
    fw = ceg.load_framework_RASPA("CIT-7", FFNAME)
    p = ProbeSystem.build(fw, ff, "Ar")
    n, (nx, ny, nz) = len(fw), p.num_supercell
    assert (nx, ny, nz) == (2, 3, 3)
    a, b, c = fw.mat[:, 0], fw.mat[:, 1], fw.mat[:, 2]
    for ix, iy, iz, i in ((0, 0, 0, 5), (1, 0, 0, 0), (0, 2, 1, 17), (1, 2, 2, 59)):
        idx = ix * n * ny * nz + iy * n * nz + iz * n + i
        np.testing.assert_allclose(p.positions[idx], fw.position[i] + ix * a + iy * b + iz * c, rtol=0, atol=1e-12)
    np.testing.assert_allclose(p.mat, np.column_stack((2 * a, 3 * b, 3 * c)))
    np.testing.assert_allclose(p.mat @ p.invmat, np.eye(3), atol=1e-12)


def test_get_atom_name():
    assert get_atom_name("Oz_2") == "Oz" and get_atom_name("C_co2") == "C_co2"
    assert get_atom_name("Na") == "Na" and get_atom_name("Si12") == "Si" and get_atom_name("O_co2_7") == "O_co2"


# ------------------------------------------------------------------ force field
def test_forcefield_fixture_rules():
    """raspa.jl:611-702 on the fixture."""
    ff = ceg.parse_forcefield_RASPA(FFNAME)
    assert ff.sdict["UNIT"] == 1 and ff.sdict["Oz"] == 2 and ff.sdict["Siz"] == 5 and ff.sdict["Na"] == 8 and ff.sdict["Ar"] == 20
    assert ff.cutoff == 12.0
    r = ff["Ar", "Oz"]
    assert isinstance(r, InteractionRule) and r.kind == FF.LennardJones and r.params == [107.69, 3.15]
    assert r.shift == pytest.approx(4 * 107.69 * ((3.15 / 12) ** 12 - (3.15 / 12) ** 6), rel=1e-12) and not r.tailcorrection
    assert ff["Ar", "Siz"].kind == FF.NoInteraction and ff["Ar", "Alz"].kind == FF.NoInteraction
    assert ff["Ar", "Na"].params == [262.0, 2.396]
    s = ff["Na", "Oa"]
    assert isinstance(s, InteractionRuleSum)
    assert [x.kind for x in s.rules] == [FF.HardSphere, FF.CoulombEwaldDirect, FF.Buckingham]
    assert s.rules[0].params == [1.5, 0.0] and s.rules[2].params == [5.581e7, 3.985, 9.167e5]
    assert s.rules[1].params[0] == pytest.approx(0.26505830360350674, rel=1e-15)
    assert all(x.shift == 0.0 for x in s.rules)                       # general rule: truncated
    assert ff["Na", "Siz"].kind == FF.CoulombEwaldDirect
    assert ff.needsvdwgrid("Na") and ff.needsvdwgrid("Ar")
    # Lorentz-Berthelot mixing of two LJ species (forcefields.jl:72-76)
    m = ff["C_co2", "N_n2"]
    lj = [x for x in ([m] if isinstance(m, InteractionRule) else m.rules) if x.kind == FF.LennardJones][0]
    assert lj.params[0] == pytest.approx(math.sqrt(28.129 * 36.4)) and lj.params[1] == pytest.approx((2.757 + 3.32) / 2)


def test_rule_table_flattening():
    ff = ceg.parse_forcefield_RASPA(FFNAME)
    rules, off = ff.rule_table(ff.sdict["Na"])
    assert off.dtype == np.int32 and len(off) == ff.nkinds + 1 and off[0] == 0 and off[-1] == len(rules)
    k = ff.sdict["Oz"]
    run = rules[off[k - 1]:off[k]]
    assert [int(x) for x in run["kind"]] == [0, 1, 4]
    np.testing.assert_array_equal(run["p"][2], [5.581e7, 3.985, 9.167e5])
    rules_ar, off_ar = ff.rule_table(ff.sdict["Ar"])
    k = ff.sdict["Siz"]
    assert [int(x) for x in rules_ar[off_ar[k - 1]:off_ar[k]]["kind"]] == [8]


def test_vdw_grid_rule_errors_mirror_reference():
    """interactions.jl:442-443,462-467: the shim raises before the device call."""
    from ceg_hip.hostmirror.interactions import check_vdw_grid_rule, UndefinedInteractionError
    with pytest.raises(UndefinedInteractionError):
        check_vdw_grid_rule(make_rule(FF.UndefinedInteraction))
    with pytest.raises(RuntimeError, match="Monomial"):
        check_vdw_grid_rule(make_rule(FF.Monomial, 1.0, 2.0))
    with pytest.raises(RuntimeError, match="Coulomb"):
        check_vdw_grid_rule(make_rule(FF.Coulomb, 1.0, 1.0))
    check_vdw_grid_rule(make_rule(FF.NoInteraction))


# ------------------------------------------------------------------ .grid files
def _fake_grid(cset, seed=0):
    nx, ny, nz = cset.npoints
    rng = np.random.default_rng(seed)
    return rng.standard_normal((8, nx, ny, nz)).astype(np.float32)


def test_grid_file_layout_roundtrip(tmp_path):
    """grids.jl:108-116,151-155,178-183 -> parse_grid grids.jl:61-94."""
    fw = ceg.load_framework_RASPA("CIT-7", FFNAME)
    cset = ceg.GridCoordinatesSetup.from_cell(fw.mat, 1.0)
    grid = _fake_grid(cset)
    nbytes = grid.size * 4
    f_v, f_c = tmp_path / "v.grid", tmp_path / "c.grid"
    G.write_grid_file(f_v, cset, (2, 3, 3), grid)
    G.write_grid_file(f_c, cset, (2, 3, 3), grid, 1e-6)
    assert f_v.stat().st_size == 128 + nbytes + 72          # header 128 B, trailer 72 B
    assert f_c.stat().st_size == 136 + nbytes + 72
    raw = f_v.read_bytes()
    assert struct.unpack("<d", raw[:8])[0] == 1.0
    assert struct.unpack("<3i", raw[8:20]) == tuple(int(d) for d in cset.dims)
    assert struct.unpack("<3d", raw[20:44]) == tuple(cset.size)
    assert struct.unpack("<3i", raw[116:128]) == (2, 3, 3)
    np.testing.assert_array_equal(np.frombuffer(raw[128:128 + nbytes], dtype="<f4"), grid.reshape(-1))
    np.testing.assert_array_equal(np.frombuffer(raw[-72:], dtype="<f8").reshape(3, 3).T, fw.mat)
    for path, isc in ((f_v, False), (f_c, True)):
        g = ceg.parse_grid(path, isc)
        assert g.num_unitcell == (2, 3, 3) and g.higherorder
        assert g.ewald_precision == (1e-6 if isc else math.inf)
        np.testing.assert_array_equal(g.csetup.dims, cset.dims)
        np.testing.assert_array_equal(g.grid, (grid.astype(np.float64) * ceg.GRID_TO_KELVIN).astype(np.float32))
        np.testing.assert_array_equal(g.csetup.cell.mat, fw.mat)


def test_array_memory_order_matches_julia():
    """numpy (8,nx,ny,nz) C-order == Julia Array{Cfloat,4}(nz,ny,nx,8) column-major."""
    nx, ny, nz = 5, 4, 3
    a = np.arange(8 * nx * ny * nz, dtype=np.float32).reshape(8, nx, ny, nz)
    c, i, j, k = 3, 2, 1, 2
    assert a[c, i, j, k] == k + nz * (j + ny * (i + nx * c))


# ------------------------------------------------------------------ interpolation
def test_tricubic_coeff_reproduces_cubic_polynomial():
    """COEFF (constants.jl:24-89, derived in constants.tricubic_coeff) must make
    interpolate_grid exact for any tricubic polynomial."""
    C = tricubic_coeff()
    assert C.shape == (64, 64) and np.all(C == np.round(C))
    rng = np.random.default_rng(1)
    a = rng.standard_normal((4, 4, 4))                      # a[i,j,k] x^i y^j z^k
    def deriv(x, y, z, ox, oy, oz):
        tot = 0.0
        for i in range(4):
            for j in range(4):
                for k in range(4):
                    def d(e, o, t):
                        if o > e:
                            return 0.0
                        return (e if o else 1) * t ** (e - o)
                    tot += a[i, j, k] * d(i, ox, x) * d(j, oy, y) * d(k, oz, z)
        return tot
    chans = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
    X = np.array([deriv(c & 1, (c >> 1) & 1, (c >> 2) & 1, *ch) for ch in chans for c in range(8)])
    for r in ((0.3, 0.7, 0.2), (0.0, 0.0, 0.0), (0.99, 0.5, 0.01)):
        assert G.interpolate_from_corners(X, r, False) == pytest.approx(deriv(*r, 0, 0, 0), rel=1e-12, abs=1e-12)


def test_interpolate_vdw_blocked_corner():
    X = np.zeros(64)
    X[3] = 6e6
    assert G.interpolate_from_corners(X, (0.5, 0.5, 0.5), True) == 1e100          # grids.jl:245-248
    assert G.interpolate_from_corners(X, (0.5, 0.5, 0.5), False) != 1e100


def test_zero_and_invalid_grids():
    assert G.interpolate_grid(G.EnergyGrid.trivial(True), [0, 0, 0]) == 0.0
    with pytest.raises(ValueError):
        G.interpolate_grid(G.EnergyGrid.trivial(False), [0, 0, 0])


# ------------------------------------------------------------------ sharding
def test_slab_range_partition():
    from ceg_hip.distributed import slab_range
    for nx in (1, 7, 218, 256):
        for world in (1, 2, 3, 4, 8):
            spans = [slab_range(nx, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nx
            assert all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_roofline_workload_shape():
    w = W.roofline_workload("Ar", 255)
    assert w.natoms == 11664 and w.cset.npoints == (256, 256, 256) and w.npoints == 16777216
    assert w.probe_vdw.num_supercell == (1, 1, 1)
    np.testing.assert_array_equal(w.probe_vdw.positions, w.probe_coulomb.positions)
    assert abs(w.probe_coulomb.charges.sum() - 12 * (-122.769)) < 1e-6


def test_grid_file_headers_match_committed_fixture(tmp_path):
    """tests/golden/grid_headers.json ("headers of the corresponding .grid files"): the bytes around
    the payload for the fixtures' grids -- spacing, dims (217,203,189 for CHA at 0.15 A),
    size, shift, delta, unit-cell lengths, num_unitcell, Ewald precision, cell-matrix trailer."""
    import json
    fx = json.loads((ROOT / "tests" / "golden" / "grid_headers.json").read_text())
    assert fx["CHA_1.4_3b4eeb96/0.15/vdw"]["dims"] == [217, 203, 189]
    assert fx["CIT-7/0.15/vdw"]["num_unitcell"] == [2, 3, 3]
    for key, rec in fx.items():
        fwname, spacing, kind = key.split("/")
        fw = ceg.load_framework_RASPA(fwname, FFNAME)
        cset = ceg.GridCoordinatesSetup.from_cell(fw.mat, float(spacing))
        grid = np.zeros((8, 0), dtype=np.float32)                 # header + trailer only: empty payload
        f = tmp_path / "h.grid"
        G.write_grid_file(f, cset, tuple(rec["num_unitcell"]), grid, 1e-6 if kind == "coulomb" else None)
        raw = f.read_bytes()
        nh = 136 if kind == "coulomb" else 128
        assert len(raw) == nh + 72
        assert raw[:nh].hex() == rec["header_hex"] and raw[nh:].hex() == rec["trailer_hex"], key
        assert rec["payload_bytes"] == 32 * int(np.prod(np.asarray(cset.dims) + 1))


def test_scripts_compile_and_pair_work_count():
    """bench.py / scripts only run on the GPU box: at least their syntax is checked here (Python 3.10).  The counted
    minimum work of the roofline workload (bench.py's flop count) is reproducible: CHA density x cutoff sphere."""
    import py_compile
    root = Path(__file__).resolve().parent.parent
    for f in [root / "bench.py", root / "__graft_entry__.py", *sorted((root / "scripts").glob("*.py")), *sorted((root / "tests" / "perf").glob("*.py"))]:
        py_compile.compile(str(f), doraise=True)
    from ceg_hip import workloads as W
    w = W.roofline_workload("Ar", 255)
    pw = W.count_pair_work(w, planes=3, stride=16)
    rho = 972 / 22669.1405
    assert abs(pw["in_cutoff_per_point"] - rho * 4 / 3 * math.pi * 12 ** 3) < 3.0          # ~310
    assert abs(pw["lj_per_point"] / pw["in_cutoff_per_point"] - 648 / 972) < 0.01            # only the O atoms carry an Ar rule
    assert pw["buckingham_per_point"] == 0.0
    pn = W.count_pair_work(W.roofline_workload("Na", 255), planes=3, stride=16)
    assert pn["lj_per_point"] == 0.0 and abs(pn["buckingham_per_point"] - pw["lj_per_point"]) < 1e-9


def test_bench_cpu_baseline_leg(monkeypatch):
    """bench.py's cpu_baseline (the oracle timed on a bounded sample; runs on the GPU box at N = 1 only) on a small workload:
    automatic row count, real thread count, preallocated output, the per-thread rates it reports."""
    import importlib.util
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("bench_module", root / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv("CEG_BENCH_CPU_SECONDS", "0.5")
    w = W.fixture_workload("CIT-7", "Ar", 0.0, dims=(23, 23, 23))
    for mode in ("fused", "vdw"):
        cb = bench.cpu_baseline(w, mode, -1)
        assert cb["kind"] == "port" and cb["unit"] == "grid-points/s" and cb["value"] > 0
        from oracle import oracle as O
        assert cb["cores"] == O.usable_cpus() and 0.01 < cb["parallel_efficiency"] < 4.0      # not 1 after the single-thread probe
        assert cb["pair_checks_per_s_per_thread"] == pytest.approx(cb["pair_checks_per_s"] / cb["cores"])
        assert "preallocated" in cb["sample"]


def test_interpolate_grid_without_derivatives_host_mirror_vs_oracle(oracle):
    """The "no derivatives" branch of interpolate_grid (grids.jl:259-269) in the host mirror and in the oracle: same values where
    the reference's [x, y, z] addressing of its [z, y, x] array stays in bounds, IndexError / NaN where Julia raises BoundsError."""
    import math
    from ceg_hip import grids as G, workloads as W
    rng = np.random.default_rng(21)
    for dims in ((21, 21, 21), (25, 17, 21)):
        mat = np.diag([20.0, 20.0, 20.0]) + np.array([[0, 1.5, -0.7], [0, 0, 2.1], [0, 0, 0]])
        cset = W.grid_setup_with_dims(mat, dims)
        nx, ny, nz = cset.npoints
        eg = G.EnergyGrid(cset, (1, 1, 1), math.inf, False, rng.normal(size=(8, nx, ny, nz)).astype(np.float32))
        pts = rng.uniform(-30, 50, (400, 3))
        ref = oracle.interpolate_points(eg, pts)
        ok = ~np.isnan(ref)
        assert ok.all() == (len(set(dims)) == 1) and ok.any()
        for q in range(60):
            if ok[q]:
                assert G.interpolate_grid(eg, pts[q]) == pytest.approx(ref[q], rel=1e-10, abs=1e-12)
            else:
                with pytest.raises(IndexError):
                    G.interpolate_grid(eg, pts[q])


def test_round3_entry_points_validate_and_have_no_cpu_path(tmp_path):
    """The multi-probe build, the .grid -> device loader and the higher-order switch validate their arguments before touching a
    device and, with valid arguments on a box without a GPU, fail with CEG_ERR_NO_DEVICE -- never a CPU result."""
    import ctypes as C
    from ceg_hip.plan import MultiGridPlan
    lib = _abi.load_library()
    w1 = W.fixture_workload("CIT-7", "C_co2", 2.0)
    w2 = W.fixture_workload("CIT-7", "O_co2", 2.0)
    # host-side checks of the wrapper: probe count, one framework for all probes
    with pytest.raises(ValueError):
        MultiGridPlan(w1.cset, [], w1.probe_coulomb, w1.alpha)
    other = W.fixture_workload("CHA_1.4_3b4eeb96", "Ar", 2.0)
    with pytest.raises(ValueError):
        MultiGridPlan(w1.cset, [w1.probe_vdw, other.probe_vdw], None, 0.0)
    # C side: nprobes out of range / NULL tables are CEG_ERR_INVALID whatever the machine
    h = C.c_void_p()
    dims, size, shift, delta = G._grid_args(w1.cset)
    pos = np.ascontiguousarray(w1.probe_vdw.positions, dtype=np.float64)
    kinds = np.ascontiguousarray(w1.probe_vdw.atomkinds, dtype=np.int64)
    mat, inv = G._matT(w1.probe_vdw.mat), G._matT(w1.probe_vdw.invmat)
    tabs = [p.forcefield.rule_table(p.probe) for p in (w1.probe_vdw, w2.probe_vdw)]
    rules_pp = (C.c_void_p * 2)(*[t[0].ctypes.data for t in tabs])
    offs_pp = (C.c_void_p * 2)(*[t[1].ctypes.data for t in tabs])
    geo = (_abi.i32ptr(dims), _abi.dptr(size), _abi.dptr(shift), _abi.dptr(delta))
    args = lambda n, rp, op: (C.byref(h), 0, _abi.dptr(pos), _abi.i64ptr(kinds), None, len(kinds), _abi.dptr(mat), _abi.dptr(inv), 0, 1e9, 144.0,
                              n, rp, op, w1.forcefield.nkinds, 0.0, *geo)
    assert lib.ceg_plan_create_multi(*args(0, rules_pp, offs_pp)) == -1
    assert lib.ceg_plan_create_multi(*args(5, rules_pp, offs_pp)) == -1 and b"nprobes" in lib.ceg_last_error()
    assert lib.ceg_plan_create_multi(*args(2, None, offs_pp)) == -1
    assert lib.ceg_plan_build_multi(None, 1.0, 1.0, 1.0, 1.0, 0, 1, None, None, 1, 0, None) == -1
    assert lib.ceg_plan_num_probes(None) == 0
    assert lib.ceg_interp_set_higherorder(None, 0) == -1
    assert lib.ceg_interp_create_from_file(C.byref(h), 0, None, 0, 1.0, None, None, None) == -1
    assert lib.ceg_interp_create_from_file(C.byref(h), 0, b"/nonexistent.grid", 0, 1.0, _abi.dptr(mat), None, None) == -1      # mat without invmat
    if lib.ceg_device_count() > 0:
        return
    assert lib.ceg_plan_create_multi(*args(2, rules_pp, offs_pp)) == -2 and b"no HIP device" in lib.ceg_last_error()
    with pytest.raises(_abi.CegError) as ei:
        G.build_multi_arrays([w1.probe_vdw, w2.probe_vdw], w1.probe_coulomb, w1.alpha, w1.cset)
    assert ei.value.code == -2
    (tmp_path / "x.grid").write_bytes(b"\0" * 400)
    assert lib.ceg_interp_create_from_file(C.byref(h), 0, str(tmp_path / "x.grid").encode(), 0, 1.0, None, None, None) == -2


def test_pmc_record_matches_the_grid_kernel_sources():
    """bench.py takes roofline.frac from profiles/pmc_summary.json; a record collected on OTHER grid-kernel sources is flagged stale in
    the line the driver records.  A round must not end on a stale headline record: this fails until scripts/pmc.sh +
    scripts/pmc_merge.py have been re-run on the current csrc/ (the hash covers bench.GRID_KERNEL_SOURCES only -- edits of the consumer
    kernels do not invalidate it)."""
    import json
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    import bench
    rec = json.loads((root / "profiles" / "pmc_summary.json").read_text())["fused/Ar/255/1"]
    assert rec["csrc_sha256"] == bench.csrc_sha256(), "profiles/pmc_summary.json is stale: re-run scripts/pmc.sh fused_ar fused/Ar/255/1 and scripts/pmc_merge.py"
    assert rec["source"].startswith("profiles/")


def test_erfc_table_of_the_pair_kernels(tmp_path):
    """ceg_pairfrac::build_erfc_table (host code of csrc/ceg_pairfrac.h): the r^2-indexed records the pair kernels read erfc(alpha r)/r
    from, evaluated here with the kernel's Horner form against scipy's erfc -- 1e-13 of the value + 1e-15 of the function at r = 1."""
    import shutil, subprocess
    from scipy.special import erfc
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "erfc_table_dump"
    src = Path(__file__).parent / "cpp" / "erfc_table_dump.hip"
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-o", str(exe), str(src)], check=True, capture_output=True)
    for alpha, s_max in ((0.26505830360350674, 144.0), (0.5, 100.0), (0.08, 196.0)):
        out = subprocess.run([str(exe), repr(alpha), "1.0", repr(s_max)], check=True, capture_output=True, text=True).stdout.split()
        ok, base, ni, worst, nrec = int(out[0]), int(out[1]), int(out[2]), float(out[3]), int(out[4])
        assert ok == 1 and worst < 1e-13 and ni > 100
        rec = np.array(out[5:], dtype=np.float64).reshape(ni, nrec)
        rng = np.random.default_rng(3)
        s = np.concatenate([rng.uniform(1.0, s_max, 20000), [1.0, s_max, np.nextafter(2.0, 0), 2.0, np.nextafter(128.0, 0)]])
        s = s[s <= s_max]
        hi = (s.view(np.uint64) >> np.uint64(32)).astype(np.int64)
        key = (hi >> 15) - base
        assert key.min() >= 0 and key.max() < ni
        s_lo = ((hi >> 15) << 15).astype(np.uint64) << np.uint64(32)
        t = s - s_lo.view(np.float64)
        c = rec[key]
        v = c[:, 6]
        for k in range(5, -1, -1):
            v = v * t + c[:, k]
        ref = erfc(alpha * np.sqrt(s)) / np.sqrt(s)
        top = erfc(alpha)
        assert np.all(np.abs(v - ref) <= 2e-13 * np.abs(ref) + 2e-15 * top), float(np.max(np.abs(v - ref) / (np.abs(ref) + 1e-2 * top)))
    # an alpha for which the fit cannot hold the tolerance is refused (the kernels then keep the exp / erfcx polynomials)
    out = subprocess.run([str(exe), "3.0", "1.0", "144.0"], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) in (0, 1)
