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
    """got / ref: [plane, channel, y, z] stored Float32 values.  One summary line, then one line per channel: values that differ,
    worst ULP distance, worst relative error with NO absolute floor (|got - ref| / |ref| over ref != 0), and how small the worst
    value is against its channel (a sum that cancels shows as a tiny fraction of the channel median)."""
    both_nan = np.isnan(got) & np.isnan(ref)
    same = (got == ref) | both_nan
    fin = np.isfinite(ref) & np.isfinite(got)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref)), name
    ulp_all = ulp_distance(got[fin], ref[fin])
    print(f"{name:78s} values {got.size:10d}  identical {same.sum() / got.size * 100:9.5f} %  differing {int((~same).sum()):6d}  "
          f"max ulp {int(ulp_all.max()) if ulp_all.size else 0:3d}  non-finite {int((~fin).sum())}", flush=True)
    for ch in range(got.shape[1]):
        g, r = got[:, ch], ref[:, ch]
        f = np.isfinite(g) & np.isfinite(r)
        g, r = g[f], r[f]
        if not g.size:
            continue
        ulp = ulp_distance(g, r)
        nzr = r != 0
        rel = np.zeros(g.shape)
        rel[nzr] = np.abs(g[nzr].astype(np.float64) - r[nzr]) / np.abs(r[nzr].astype(np.float64))
        med = float(np.median(np.abs(r)))
        q = int(np.argmax(ulp))
        tail = f"  worst at |value| = {abs(float(r[q])) / med:.1e} x channel median" if ulp[q] > 1 else ""
        print(f"      channel {ch}: differing {int((g != r).sum()):5d}  max ulp {int(ulp.max()):3d}  max rel (no floor) {rel.max():.2e}{tail}", flush=True)
        WORST[ch] = max(WORST.get(ch, 0), int(ulp.max()))
        WORST_REL[ch] = max(WORST_REL.get(ch, 0.0), float(rel.max()))

WORST, WORST_REL = {}, {}

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
print("ALL CONFIGS  max ulp per channel:", " ".join(f"c{c}={WORST[c]}" for c in sorted(WORST)))
print("ALL CONFIGS  max rel (no floor) per channel:", " ".join(f"c{c}={WORST_REL[c]:.2e}" for c in sorted(WORST_REL)))
print(f"ALL CONFIGS  channel 0 (the energy): max ulp {WORST.get(0)}  max rel {WORST_REL.get(0):.2e}  (north_star tolerance 1e-6)")
