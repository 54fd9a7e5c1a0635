"""Parity at scale: stored Float32 grids of the HIP path vs the CPU oracle on whole x-planes of the
BASELINE configurations; counts identical values, worst ULP distance and worst relative error per
channel group.  Writes one line per (config, grid).  Run on a GPU box: python tests/perf/parity_report.py"""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..')]
import numpy as np
from ceg_hip import workloads as W, grids as G
from oracle import oracle as O

def ulp_distance(a, b):
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia); ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)

def report(name, got, ref):
    both_nan = np.isnan(got) & np.isnan(ref)
    same = (got == ref) | both_nan
    fin = np.isfinite(ref) & np.isfinite(got)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref)), name
    ulp = ulp_distance(got[fin], ref[fin])
    med = np.median(np.abs(ref[fin])) if fin.any() else 0.0
    rel = np.abs(got[fin].astype(np.float64) - ref[fin]) / np.maximum(np.abs(ref[fin].astype(np.float64)), 1e-9 * med)
    where = ""
    if rel.size and rel.max() > 2e-7:          # beyond one Float32 ULP: show how small the value is against its channel
        q = int(np.argmax(rel))
        ch = int(np.unravel_index(np.flatnonzero(fin)[q], got.shape)[1])          # arrays are [plane, channel, y, z]
        col = ref[:, ch]
        cmed = np.median(np.abs(col[np.isfinite(col)]))
        where = f"  (worst at |value| = {abs(float(ref[fin][q])) / cmed:.1e} x the median of channel {ch}: a sum that cancels)"
    print(f"{name:78s} values {got.size:10d}  identical {same.sum() / got.size * 100:9.5f} %  differing {int((~same).sum()):6d}  "
          f"max ulp {int(ulp.max()) if ulp.size else 0:3d}  max rel {rel.max() if rel.size else 0:.2e}  non-finite {int((~fin).sum())}{where}", flush=True)

def planes(w, n):
    nx = w.cset.npoints[0]
    return sorted(set(int(x) for x in np.linspace(0, nx - 1, n)))

configs = [("config 1  CHA 0.5 A", W.fixture_workload("CHA_1.4_3b4eeb96", "Na", 0.5), None),
           ("config 2  CHA 0.1 A, Na", W.fixture_workload("CHA_1.4_3b4eeb96", "Na", 0.1), 6),
           ("config 2  CHA 0.1 A, Ar", W.fixture_workload("CHA_1.4_3b4eeb96", "Ar", 0.1), 3),
           ("CIT-7 0.15 A (2x3x3, triclinic), Na", W.fixture_workload("CIT-7", "Na", 0.15), None),
           ("config 3  R: 11664 atoms x 256^3, Ar", W.roofline_workload("Ar", 255), 2),
           ("config 3  R: 11664 atoms x 256^3, Na", W.roofline_workload("Na", 255), 1)]
for name, w, npl in configs:
    nx, ny, nz = w.cset.npoints
    t = time.perf_counter()
    gv = G.build_vdw_array(w.probe_vdw, w.cset)
    gc = G.build_coulomb_array(w.probe_coulomb, w.alpha, w.cset)
    tg = time.perf_counter() - t
    sel = list(range(nx)) if npl is None else planes(w, npl)
    t = time.perf_counter()
    for kind, grid in (("vdw", gv), ("coulomb", gc)):
        refs, gots = [], []
        for i in sel:
            if kind == "vdw":
                lam, thr = G.vdw_scaling(); ref, _ = O.grid_vdw(w.probe_vdw, w.cset, lam, thr, i, i + 1)
            else:
                lam, thr = G.coulomb_scaling(); ref, _ = O.grid_coulomb(w.probe_coulomb, w.alpha, w.cset, lam, thr, i, i + 1)
            refs.append(ref[:, i]); gots.append(grid[:, i])
        report(f"{name} / {kind} ({len(sel)} of {nx} x-planes)", np.stack(gots), np.stack(refs))
    print(f"   (GPU one-shot builds {tg*1e3:.0f} ms, oracle {time.perf_counter() - t:.1f} s on {O.max_threads()} threads)", flush=True)
