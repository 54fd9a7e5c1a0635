"""One guest atom, one trial atom at a distance r (0.4 ... 12 A): ceg_pairs_energy through the fractional-coordinate kernel (records of
ceg_pairfrac.h), through the Cartesian kernel (CEG_HIP_PAIRS_FRAC=0) and the oracle -- where do they differ, pair kind by pair kind"""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..'), os.path.join(here, '..')]
import numpy as np
import ceg_hip as ceg
from ceg_hip import _abi
from ceg_hip.hostmirror.constants import COULOMBIC_CONVERSION_FACTOR
from oracle import oracle as O
from test_gpu_consumers import _pairs_gpu
ceg.setdir_RASPA(os.path.join(here, '..', 'golden', 'raspa'))
ff = ceg.parse_forcefield_RASPA("BoulfelfelSholl2021")
rules, offsets = ff.pair_table()
lib = _abi.load_library()
mat = np.diag([40.0, 40.0, 40.0])
inv = np.linalg.inv(mat)
r = np.concatenate([np.linspace(0.4, 2.0, 400), np.linspace(2.0, 11.999, 600)])
atom = np.array([[20.0, 20.0, 20.0]])
trial = (atom + r[:, None] * np.array([[0.6, 0.48, 0.64]]))[:, None, :]
for a, b in (("O_co2", "O_co2"), ("C_co2", "O_co2"), ("Na", "O_co2"), ("Ar", "Ar"), ("C_co2", "Na")):
    ka, kb = ff.sdict[a] - 1, ff.sdict[b] - 1
    args = (mat, ff.cutoff ** 2, rules, offsets, ff.nkinds, COULOMBIC_CONVERSION_FACTOR, atom, [ka], [0], trial, [kb], -1)
    ref = O.single_contribution_vdw_raw(mat, inv, *args[1:])
    os.environ.pop("CEG_HIP_PAIRS_FRAC", None)
    frac = _pairs_gpu(lib, *args)
    os.environ["CEG_HIP_PAIRS_FRAC"] = "0"
    cart = _pairs_gpu(lib, *args)
    os.environ.pop("CEG_HIP_PAIRS_FRAC", None)
    fin = np.isfinite(ref) & (ref != 0)
    ef = np.abs(frac[fin] - ref[fin]) / np.abs(ref[fin]); ec = np.abs(cart[fin] - ref[fin]) / np.abs(ref[fin])
    i = int(np.argmax(ef))
    print(f"{a}-{b}: worst relative deviation from the oracle: fractional {ef.max():.2e} at r = {r[fin][i]:.4f} (E = {ref[fin][i]:.4e}), Cartesian {ec.max():.2e}")
