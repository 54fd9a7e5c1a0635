import sys, os
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import numpy as np
from ceg_hip import grids as G, workloads as W, _abi
from ceg_hip.plan import GridPlan
from oracle import oracle as O
np.set_printoptions(precision=17, linewidth=200)
w = W.fixture_workload("CHA_1.4_3b4eeb96_Na_11812", "Ar", 0.7)
lam, thr = G.vdw_scaling()
ref, raw = O.grid_vdw(w.probe_vdw, w.cset, lam, thr, want_raw=True)
got = G.build_vdw_array(w.probe_vdw, w.cset)
bad = np.argwhere(~np.isclose(got, ref, rtol=1e-5, atol=1e-3, equal_nan=True))
print(len(bad), "bad entries")
plan = GridPlan(w.cset, w.probe_vdw, None, 0.0)
seen = set()
for c, i, j, k in bad[:400]:
    if (i, j, k) in seen: continue
    seen.add((i, j, k))
    if len(seen) > 6: break
    pt = np.array([i * w.cset.size[0] / w.cset.dims[0] + w.cset.shift[0], j * w.cset.size[1] / w.cset.dims[1] + w.cset.shift[1], k * w.cset.size[2] / w.cset.dims[2] + w.cset.shift[2]])
    print("point", (i, j, k), pt)
    print(" ref f32", ref[:, i, j, k]); print(" got f32", got[:, i, j, k])
    print(" oracle raw", raw[i, j, k])
    print(" gpu brute ", plan.eval_points("vdw", pt[None], 1)[0])
    print(" gpu culled", plan.eval_points("vdw", pt[None], 2)[0])
    # nearest atoms (min image over 27)
    P = w.probe_vdw.positions; M = w.probe_vdw.mat
    best = []
    for a in range(-1, 2):
        for b in range(-1, 2):
            for cc in range(-1, 2):
                d = pt - (P + M @ np.array([a, b, cc], float))
                r = np.linalg.norm(d, axis=1)
                q = np.argmin(r)
                best.append((r[q], q, (a, b, cc), d[q]))
    best.sort(key=lambda t: t[0])
    for r, q, n, d in best[:2]:
        print("  near atom", q, "kind", w.probe_vdw.atomkinds[q], "n", n, "r", r, "d", d, "pos", P[q])
