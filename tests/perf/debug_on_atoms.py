"""Diagnostic for tests/test_gpu_parity.py::test_points_on_atoms_nan_inf_patterns: where do the stored values of the GPU grid and
the oracle differ, and how far is each atom from those points (in units that show a cutoff decision)?"""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..'), os.path.join(here, '..')]
import numpy as np
from ceg_hip import grids as G, workloads as W
from oracle import oracle as O
from util import synthetic_probes, grid_points

L = 30.0
mat = np.diag([L, L, L])
cset = W.grid_setup_with_dims(mat, (15, 15, 15))
pos = np.array([[4.0, 6.0, 8.0], [10.0, 10.0, 10.0], [20.0, 2.0, 28.0], [11.3, 17.7, 5.1], [0.0, 0.0, 0.0]])
kinds = np.array([1, 2, 4, 2, 1])
q = np.array([1.0, -1.0, 0.5, 0.0, -0.7])
pv, pc = synthetic_probes(mat, pos, kinds, q)
lam, thr = G.vdw_scaling()
got = G.build_vdw_array(pv, cset)
ref = O.grid_vdw(pv, cset, lam, thr)[0]
pts = grid_points(cset).reshape(16, 16, 16, 3)
bad = np.argwhere((got[0] != ref[0]) & ~(np.isnan(got[0]) & np.isnan(ref[0])))
print("differing channel-0 values:", len(bad))
for i, j, k in bad[:20]:
    p = pts[i, j, k]
    d = p - pos
    d -= L * np.round(d / L)
    r2 = (d ** 2).sum(1)
    print((i, j, k), p, "got", got[0, i, j, k], "ref", ref[0, i, j, k], " r2 - 144 per atom:", r2 - 144.0, "kinds", kinds)
lamc, thrc = G.coulomb_scaling()
gotc = G.build_coulomb_array(pc, 0.265, cset)
refc = O.grid_coulomb(pc, 0.265, cset, lamc, thrc)[0]
bad = np.argwhere((gotc[0] != refc[0]) & ~(np.isnan(gotc[0]) & np.isnan(refc[0])))
print("coulomb: differing channel-0 values:", len(bad), " median |ref|", np.median(np.abs(refc[0][np.isfinite(refc[0])])))
rawc_ref = O.points_coulomb(pc, 0.265, pts.reshape(-1, 3)).reshape(16, 16, 16, 8)
for i, j, k in bad[:24]:
    p = pts[i, j, k]
    d = p - pos
    d -= L * np.round(d / L)
    r2 = (d ** 2).sum(1)
    print((i, j, k), "got", gotc[0, i, j, k], "ref", refc[0, i, j, k], "raw ref", rawc_ref[i, j, k, 0], " r2 per atom:", r2, "q", q)
from ceg_hip.plan import GridPlan
plan = GridPlan(cset, pv, pc, 0.265)
for algo, name in ((1, "bruteforce"), (2, "culled")):
    raw = plan.eval_points("vdw", pts.reshape(-1, 3), algo)
    refraw = O.points_vdw(pv, pts.reshape(-1, 3))
    dd = np.abs(raw[:, 0] - refraw[:, 0])
    fin = np.isfinite(dd)
    w = np.argsort(-np.where(fin, dd, 0))[:5]
    print(name, "raw channel 0 worst abs diffs:", [(int(x), float(raw[x, 0]), float(refraw[x, 0])) for x in w])
plan.close()
