"""Long randomized parity run (not part of the suite): random cells / atoms / cutoffs / grids, both kernels
(brute force, culled) and the stored Float32 grids against the oracle.  usage: fuzz_soak.py [nconfigs] [seed] [max_atoms] [max_half_dim]"""
import os, sys, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..'), os.path.join(here, '..')]
import numpy as np
from ceg_hip import _abi, grids as G, workloads as W
from ceg_hip.plan import GridPlan
from ceg_hip.hostmirror.utils import mat_from_parameters, perpendicular_lengths
from oracle import oracle as O
from oracle.compare import compare_grids
from util import compare_raw, grid_points, random_atoms, synthetic_probes

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_atoms = int(sys.argv[3]) if len(sys.argv) > 3 else 260
max_half_dim = int(sys.argv[4]) if len(sys.argv) > 4 else 9
rng = np.random.default_rng(seed)
done = fails = 0
stats = {"ortho": 0, "stale": 0, "plain": 0, "generic": 0, "onatom": 0}
t0 = time.time()
while done < n_cfg:
    lengths = rng.uniform(24.2, 45.0, 3)
    angles = rng.uniform(56.0, 124.0, 3) if rng.random() < 0.75 else rng.uniform(88.5, 91.5, 3)
    try:
        mat = mat_from_parameters(tuple(lengths), tuple(angles))
    except Exception:
        continue
    if not np.all(np.isfinite(mat)) or np.linalg.det(mat) <= 0:
        continue
    cutoff = float(rng.choice([8.0, 9.5, 11.0, 12.0]))
    if perpendicular_lengths(mat).min() < 2 * cutoff:
        continue
    n = int(rng.integers(1, max_atoms))
    pos = random_atoms(mat, n, rng, min_sep=float(rng.uniform(0.9, 2.0)))
    generic = rng.random() < 0.2
    hs = float(rng.choice([1.5, 1.5, 2.6]))
    # kind palettes: everything mixed (per-candidate LJ / Buckingham classes), LJ kinds only, Buckingham + none only (the single
    # tabulated Buckingham class of the culled kernel)
    palette = [np.array([1, 2, 3, 4]), np.array([1, 3, 4]), np.array([2, 3])][int(rng.integers(0, 3))]
    stats["palette%d" % len(palette)] = stats.get("palette%d" % len(palette), 0) + 1
    kinds_used = palette[rng.integers(0, len(palette), n)]
    charges_used = rng.uniform(-1.5, 1.5, n)
    multi_ok = 2 not in palette                      # probe P (5) is Buckingham against kind 2: not a multi-probe candidate then
    pv, pc = synthetic_probes(mat, pos, kinds_used, charges_used, cutoff=cutoff, generic=generic, hs_radius=hs)
    ortho, safemin2 = pv.periodic_setup()
    stats["ortho" if ortho else ("stale" if safemin2 < cutoff ** 2 else "plain")] += 1
    stats["generic"] += int(generic)
    dims = tuple(int(x) for x in 2 * rng.integers(0, max_half_dim, 3) + 1)
    cset = W.grid_setup_with_dims(mat, dims)
    alpha = float(rng.uniform(0.18, 0.33))
    what = f"cfg{done} seed{seed}: L {np.round(lengths, 3)} A {np.round(angles, 2)} cutoff {cutoff} n {n} dims {dims} alpha {alpha:.4f} generic {generic} hs {hs}"
    try:
        plan = GridPlan(cset, pv, pc, alpha)
        assert plan.can_cull
        pts = grid_points(cset)
        if rng.random() < 0.3:                      # some points exactly on atoms / very close
            k = min(len(pts), n, 5)
            pts[:k] = np.clip(pos[:k] + rng.choice([0.0, 1e-9, 0.3], (k, 1)), cset.shift, cset.shift + cset.size)
            stats["onatom"] += 1
        ref_v = O.points_vdw(pv, pts); ref_c = O.points_coulomb(pc, alpha, pts)
        for algo in (_abi.ALGO_BRUTEFORCE, _abi.ALGO_CULLED):
            compare_raw(plan.eval_points("vdw", pts, algo), ref_v, what + f"/vdw/algo{algo}")
            compare_raw(plan.eval_points("coulomb", pts, algo), ref_c, what + f"/coulomb/algo{algo}")
        plan.close()
        lam, thr = G.vdw_scaling(); ref, _ = O.grid_vdw(pv, cset, lam, thr)
        compare_grids(G.build_vdw_array(pv, cset), ref, what + "/grid vdw")
        lam, thr = G.coulomb_scaling(); ref, _ = O.grid_coulomb(pc, alpha, cset, lam, thr)
        refc = ref
        compare_grids(G.build_coulomb_array(pc, alpha, cset), ref, what + "/grid coulomb")
        if multi_ok and not generic:
            # round 3: the same framework through a multi-probe pass (2-3 Lennard-Jones probes + the Coulomb grid in one call)
            nprobe = int(rng.integers(2, 4))
            order = [int(x) for x in rng.permutation([5, 6, 7])[:nprobe]]
            probes, _pc = synthetic_probes(mat, pos, kinds_used, charges_used, cutoff=cutoff, probes=order, hs_radius=hs)
            vg, cg = G.build_multi_arrays(probes, pc if rng.random() < 0.8 else None, alpha, cset)
            lam, thr = G.vdw_scaling()
            for k, pr in enumerate(probes):
                compare_grids(vg[k], O.grid_vdw(pr, cset, lam, thr)[0], what + f"/multi probe {order[k]}")
            if cg is not None:
                compare_grids(cg, refc, what + "/multi coulomb")
            stats["multi"] = stats.get("multi", 0) + 1
        if rng.random() < 0.5:
            # round 3: ONE probe of whatever rule class this configuration has + the Coulomb grid through the one-probe multi call
            (v1,), c1 = G.build_multi_arrays([pv], pc, alpha, cset)
            lam, thr = G.vdw_scaling()
            compare_grids(v1, O.grid_vdw(pv, cset, lam, thr)[0], what + "/one-probe multi vdw")
            compare_grids(c1, refc, what + "/one-probe multi coulomb")
            stats["oneprobe"] = stats.get("oneprobe", 0) + 1
    except AssertionError as e:
        fails += 1
        print("FAIL", what, "::", str(e)[:300], flush=True)
    done += 1
    if done % 50 == 0:
        print(f"{done} configs, {fails} failures, {time.time() - t0:.0f} s, {stats}", flush=True)
print(f"done: {done} configs, {fails} failures, {stats}")
sys.exit(1 if fails else 0)
