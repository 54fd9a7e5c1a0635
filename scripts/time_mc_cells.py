"""Guest-guest term of the device-resident MC state (ceg_mc_trial, no framework grids, no Ewald summation) with neighbour
cells against the exhaustive loop, in the north-star MC cell (CHA fixture tiled 2 x 2 x 3: 56.8 x 56.8 x 85.1 A) and in the CHA
fixture cell itself (28.4 A: where the library keeps the exhaustive loop), for growing CO2 loadings.
usage: time_mc_cells.py [bin widths, A, comma separated]"""
import ctypes as C, os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd')]
import numpy as np
import ceg_hip as ceg
from ceg_hip import _abi
from ceg_hip.hostmirror.constants import COULOMBIC_CONVERSION_FACTOR
from ceg_hip.hostmirror.utils import mat_from_parameters

ceg.setdir_RASPA(os.path.join(here, '..', 'tests', 'golden', 'raspa'))
FF = "BoulfelfelSholl2021"
ff = ceg.parse_forcefield_RASPA(FF)
co2 = ceg.load_molecule_RASPA("CO2", "TraPPE", FF)
base = np.asarray(co2.position, dtype=np.float64).reshape(-1, 3)
ids = np.array([ff.sdict[a] - 1 for a in co2.atomic_symbol], dtype=np.int32)
rules, offsets = ff.pair_table()
offsets = np.ascontiguousarray(offsets, dtype=np.int32)
lib = _abi.load_library()
bins = [float(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [3.0, 4.0, 6.0]


def handle(mat, env):
    for k in ("CEG_HIP_MC_CELLS", "CEG_HIP_MC_BIN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    matT = np.ascontiguousarray(mat.T.reshape(9))
    invT = np.ascontiguousarray(np.linalg.inv(mat).T.reshape(9))
    h = C.c_void_p()
    charge = np.zeros(ff.nkinds)
    _abi.check(lib, lib.ceg_mc_create(C.byref(h), 0, None, None, _abi.dptr(charge), ff.nkinds, _abi.dptr(matT), _abi.dptr(invT), ff.cutoff ** 2,
                                      rules.ctypes.data, _abi.i32ptr(offsets), COULOMBIC_CONVERSION_FACTOR, None, None, None, None, 0, None, None))
    return h


def run(h, pos, kinds, first, nbatch, reps, rng, mat):
    _abi.check(lib, lib.ceg_mc_set_guests(h, _abi.dptr(pos.reshape(-1)), _abi.i32ptr(kinds), _abi.i32ptr(first), len(first) - 1))
    j = 17
    cur = pos[3 * j:3 * j + 3]
    out = np.empty((nbatch + 1, 4))
    trial = np.ascontiguousarray(cur[None] + rng.uniform(-0.5, 0.5, (nbatch, 1, 3)))
    call = lambda: _abi.check(lib, lib.ceg_mc_trial(h, j, _abi.dptr(trial.reshape(-1)), nbatch, _abi.dptr(out.reshape(-1))))
    call()
    t = time.perf_counter()
    for k in range(reps):
        call()
        if k % 2 == 0:
            _abi.check(lib, lib.ceg_mc_accept(h, j, _abi.dptr(trial[k % nbatch].reshape(-1))))
    return (time.perf_counter() - t) / reps, out[:, 2].copy()


cha = mat_from_parameters((28.377, 28.377, 28.377), (94.07, 94.07, 94.07))
for label, mat, loads in (("north-star cell 2x2x3 CHA", cha @ np.diag([2.0, 2.0, 3.0]), (1000, 3000, 10000)), ("CHA fixture cell", cha, (64, 1000))):
    print(f"== {label}: perpendicular widths {np.round(1 / np.linalg.norm(np.linalg.inv(mat), axis=1), 1)} A")
    for nmol in loads:
        rng = np.random.default_rng(nmol)
        centers = rng.uniform(0, 1, (nmol, 3)) @ mat.T
        pos = np.ascontiguousarray((centers[:, None, :] + base[None]).reshape(-1, 3))
        kinds = np.ascontiguousarray(np.tile(ids, nmol))
        first = np.arange(0, 3 * nmol + 1, 3, dtype=np.int32)
        variants = [("exhaustive", {"CEG_HIP_MC_CELLS": "0"})] + [(f"cells {b:g} A", {"CEG_HIP_MC_CELLS": "1", "CEG_HIP_MC_BIN": str(b)}) for b in bins] + [("library default", {})]
        ref = {}
        for name, env in variants:
            h = handle(mat, env)
            nb = np.zeros(3, dtype=np.int32); cap = C.c_int32(0)
            on = lib.ceg_mc_neighbour_cells(h, _abi.i32ptr(nb), C.byref(cap))
            line = f"{3 * nmol:6d} guest atoms  {name:16s} {('bins ' + 'x'.join(map(str, nb))) if on else 'no cells':20s}"
            # the stateless row-f3 handle (ceg_pairs_*: atoms uploaded sorted by cell) on the same system, 16384 placements
            hp = C.c_void_p()
            _abi.check(lib, lib.ceg_pairs_create(C.byref(hp), 0, _abi.dptr(np.ascontiguousarray(mat.T.reshape(9))), _abi.dptr(np.ascontiguousarray(np.linalg.inv(mat).T.reshape(9))),
                                                 ff.cutoff ** 2, rules.ctypes.data, _abi.i32ptr(offsets), ff.nkinds, COULOMBIC_CONVERSION_FACTOR))
            molid = np.ascontiguousarray(np.repeat(np.arange(nmol), 3), dtype=np.int32)
            t = time.perf_counter()
            _abi.check(lib, lib.ceg_pairs_set_atoms(hp, _abi.dptr(pos.reshape(-1)), _abi.i32ptr(kinds), _abi.i32ptr(molid), len(pos)))
            t_set = time.perf_counter() - t
            tr = np.ascontiguousarray((np.random.default_rng(3).uniform(0, 1, (16384, 3)) @ mat.T)[:, None, :] + base[None])
            eo = np.empty(len(tr))
            lib.ceg_pairs_energy(hp, _abi.dptr(tr.reshape(-1)), _abi.i32ptr(ids), 3, len(tr), 17, _abi.dptr(eo))
            t = time.perf_counter()
            for _ in range(5):
                _abi.check(lib, lib.ceg_pairs_energy(hp, _abi.dptr(tr.reshape(-1)), _abi.i32ptr(ids), 3, len(tr), 17, _abi.dptr(eo)))
            dtp = (time.perf_counter() - t) / 5
            lib.ceg_pairs_destroy(hp)
            if "pairs" in ref:
                assert np.allclose(eo, ref["pairs"], rtol=1e-9, atol=1e-6), name
            ref.setdefault("pairs", eo.copy())
            line += f"  ceg_pairs: set_atoms {t_set * 1e6:7.0f} us, 16384 placements {dtp * 1e6:8.1f} us ({dtp * 1e9 / len(tr):6.1f} ns) |  ceg_mc:"
            for nbatch, reps in ((1, 1000), (1024, 50), (16384, 5)):
                dt, e = run(h, pos, kinds, first, nbatch, reps, np.random.default_rng(7), mat)
                if nbatch in ref:
                    assert np.allclose(e, ref[nbatch], rtol=1e-9, atol=1e-6), (name, nbatch)
                ref.setdefault(nbatch, e)
                line += f"  batch {nbatch:5d}: {dt * 1e6:8.1f} us ({dt * 1e9 / nbatch:8.1f} ns)"
            lib.ceg_mc_neighbour_cells(h, _abi.i32ptr(nb), C.byref(cap))
            print(line + (f"  capacity {cap.value}" if on else ""), flush=True)
            lib.ceg_mc_destroy(h)
