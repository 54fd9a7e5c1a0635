"""The guest-guest term of the device-resident MC state in a LARGE cell (72 x 66 x 84 A, 1500 CO2 + 24 Na: the system of
test_mc_neighbour_cells): 20 000-row displacement batches through the wave kernels, neighbour cells on / off, pair tests on fractional
coordinates (k_mcw_pairs_frac) / Cartesian (k_mcw_pairs).  No grids, no Ewald sums: the pairs launch alone.
    python tests/perf/time_mc_cells.py"""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..'), os.path.join(here, '..')]
import numpy as np
import ceg_hip as ceg
from ceg_hip import _abi
from ceg_hip.hostmirror.constants import COULOMBIC_CONVERSION_FACTOR
from test_gpu_consumers import _RawMc, _rotation

ceg.setdir_RASPA(os.path.join(here, '..', 'golden', 'raspa'))
ff = ceg.parse_forcefield_RASPA("BoulfelfelSholl2021")
lib = _abi.load_library()
rng = np.random.default_rng(11)
co2 = ceg.load_molecule_RASPA("CO2", "TraPPE", "BoulfelfelSholl2021")
base = np.asarray(co2.position, dtype=np.float64).reshape(-1, 3)
ids = [ff.sdict[a] - 1 for a in co2.atomic_symbol]
na_id = [ff.sdict["Na"] - 1]
mat = np.array([[72.0, 0, 0], [9.0, 66.0, 0], [-7.0, 11.0, 84.0]]).T
rules, offsets = ff.pair_table()
table = (mat, ff.cutoff ** 2, rules, offsets, ff.nkinds, COULOMBIC_CONVERSION_FACTOR)
mols = [(ids, c + base @ _rotation(rng).T) for c in (rng.uniform(0, 1, (1500, 3)) @ mat.T)] + [(na_id, c[None].copy()) for c in (rng.uniform(0, 1, (24, 3)) @ mat.T)]
pos = np.concatenate([p for _k, p in mols])
kinds = np.array([k for ks, _p in mols for k in ks], dtype=np.int32)
first = np.concatenate([[0], np.cumsum([len(ks) for ks, _p in mols])]).astype(np.int32)
n = 20000
ks, cur = mols[700]
trial = (rng.uniform(0, 1, (n, 3)) @ mat.T)[:, None, :] + (cur - cur[1])[None]
ref = None
for cells in ("0", "1"):
    os.environ["CEG_HIP_MC_CELLS"] = cells
    h = _RawMc(lib, *table)
    h.set_guests(pos, kinds, first)
    for fr in ("1", "0"):
        os.environ["CEG_HIP_MC_FRAC"] = fr
        rows = h.trial(700, trial)
        t = time.perf_counter()
        for _ in range(10):
            rows = h.trial(700, trial)
        dt = (time.perf_counter() - t) / 10
        if ref is None:
            ref = rows
        fin = np.isfinite(ref)
        dev = float(np.max(np.abs(rows[fin] - ref[fin]) / (np.abs(ref[fin]) + 1e-3)))
        print(f"neighbour cells {'on ' if cells == '1' else 'off'}, {'fractional' if fr == '1' else 'Cartesian '} pair tests: {dt * 1e6:9.1f} us per {n}-row call "
              f"(host arrays in and out), deviation from the first {dev:.1e}", flush=True)
    h.close()
