"""Diagnostic: the four concurrent one-shot builds of test_oneshot_reentrant_from_several_threads, repeated; reports where a
concurrent result differs from the sequential one (usage: reentrancy_probe.py [rounds])."""
import os, sys, threading
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..')]
import numpy as np
from ceg_hip import workloads as W, grids as G
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ws = [W.fixture_workload("CIT-7", "Na", 0.25), W.fixture_workload("CHA_1.4_3b4eeb96", "Ar", 0.4)]
jobs = [(lambda w=w: G.build_vdw_array(w.probe_vdw, w.cset)) for w in ws] + \
       [(lambda w=w: G.build_coulomb_array(w.probe_coulomb, w.alpha, w.cset)) for w in ws]
ref = [j() for j in jobs]
ref2 = [j() for j in jobs]
print("sequential repeat identical:", [bool(np.array_equal(a, b, equal_nan=True)) for a, b in zip(ref, ref2)])
bad = 0
for rnd in range(rounds):
    out = [None] * len(jobs)
    def run(t):
        out[t] = jobs[t]()
    th = [threading.Thread(target=run, args=(t,)) for t in range(len(jobs))]
    for x in th: x.start()
    for x in th: x.join()
    for t in range(len(jobs)):
        if not np.array_equal(out[t], ref[t], equal_nan=True):
            bad += 1
            d = np.argwhere(~((out[t] == ref[t]) | (np.isnan(out[t]) & np.isnan(ref[t]))))
            print(f"round {rnd} job {t}: {len(d)} values differ; first {d[:5].tolist()}; x-planes {sorted(set(d[:, 1].tolist()))[:20]}; "
                  f"channels {sorted(set(d[:, 0].tolist()))}; got {out[t][tuple(d[0])]!r} ref {ref[t][tuple(d[0])]!r}", flush=True)
print("mismatching results:", bad, "of", rounds * len(jobs))
