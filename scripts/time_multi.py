"""Kernel times of the multi-probe build (ceg_plan_build_multi) against the separate builds it replaces:
    python scripts/time_multi.py [reps]
CO2 (C_co2 + O_co2 + Coulomb) in the CHA fixture at 0.15 A (the reference's default spacing) and on the roofline workload
(11 664 atoms x 256^3); three and four probes on the latter; round 4: probes of SEVERAL rule classes in one plan -- Na
(Buckingham + hard sphere) + the C and O of CO2 in CIT-7 (the setup of runtests.jl:240-258) and on the roofline workload, CO2 + Ar in CHA."""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import torch
from ceg_hip import workloads as W
from ceg_hip.plan import GridPlan, MultiGridPlan

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)


def timed(fn):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


def case(label, atoms, make):
    ws = [make(a) for a in atoms]
    w = ws[0]
    nx, ny, nz = w.cset.npoints
    cs = nx * ny * nz
    K = len(atoms)
    bufs = [torch.empty((8, nx, ny, nz), dtype=torch.float32, device=dev) for _ in range(K + 1)]
    s = torch.cuda.current_stream().cuda_stream
    # (a) what the reference's call pattern costs today: one VdW-only plan per probe + one Coulomb plan
    sep = []
    for q, wq in enumerate(ws):
        p = GridPlan(wq.cset, wq.probe_vdw, None, 0.0)
        sep.append(timed(lambda: p.build_vdw(bufs[q].data_ptr(), cs, 0, nx, 0, 0, s)))
        p.close()
    p = GridPlan(w.cset, None, w.probe_coulomb, w.alpha)
    t_c = timed(lambda: p.build_coulomb(bufs[K].data_ptr(), cs, 0, nx, 0, 0, s))
    p.close()
    # (b) round 2's best: fused (probe 0 + Coulomb) + VdW-only plans for the others
    p = GridPlan(w.cset, w.probe_vdw, w.probe_coulomb, w.alpha)
    t_f = timed(lambda: p.build_fused(bufs[0].data_ptr(), bufs[K].data_ptr(), cs, 0, nx, 0, 0, s))
    p.close()
    # (c) the multi-probe plan, in its groupings
    mp = MultiGridPlan(w.cset, [x.probe_vdw for x in ws], w.probe_coulomb, w.alpha)
    ptrs = [b.data_ptr() for b in bufs[:K]]
    res = {}
    for env in ("2", "1", "0"):
        os.environ["CEG_HIP_MULTI_FUSED_NP"] = env
        res[env] = timed(lambda: mp.build(ptrs, bufs[K].data_ptr(), cs, 0, nx, 0, s))
    os.environ.pop("CEG_HIP_MULTI_FUSED_NP", None)
    t_vonly = timed(lambda: mp.build(ptrs, 0, cs, 0, nx, 0, s))
    mp.close()
    t_sep = sum(sep) + t_c
    print(f"{label}: {K} probes {atoms} + Coulomb, {nx}x{ny}x{nz} points, {w.natoms} atoms")
    print(f"   separate plans      : VdW {' + '.join(f'{t:.3f}' for t in sep)} + Coulomb {t_c:.3f} = {t_sep:.3f} ms")
    print(f"   fused(0) + VdW rest : {t_f:.3f} + {sum(sep[1:]):.3f} = {t_f + sum(sep[1:]):.3f} ms  ({(t_f + sum(sep[1:])) / t_sep:.3f} x separate)")
    for env, what in (("2", "multi: fused pair + VdW rest"), ("1", "multi: fused single + VdW rest"), ("0", "multi: Coulomb alone + VdW multi")):
        print(f"   {what:32s}: {res[env]:.3f} ms  ({res[env] / t_sep:.3f} x separate)")
    print(f"   multi, VdW grids only           : {t_vonly:.3f} ms  ({t_vonly / sum(sep):.3f} x the separate VdW builds)", flush=True)


case("CHA fixture @ 0.15 A", ("C_co2", "O_co2"), lambda a: W.fixture_workload("CHA_1.4_3b4eeb96", a, 0.15))
case("roofline workload", ("C_co2", "O_co2"), lambda a: W.roofline_workload(a, 255))
case("roofline workload", ("C_co2", "O_co2", "N_n2"), lambda a: W.roofline_workload(a, 255))
case("roofline workload", ("C_co2", "O_co2", "N_n2", "Ar"), lambda a: W.roofline_workload(a, 255))
# one probe of another rule class (Na: Buckingham + hard sphere): a one-probe plan shares the fused single-probe pass with the Coulomb grid
case("roofline workload", ("Na",), lambda a: W.roofline_workload(a, 255))
# round 4: mixed rule classes in one plan / one call (with CEG_HIP_MULTI_FUSED_NP = 2: Lennard-Jones pair fused with the Coulomb grid + the
# cation alone; 1: cation fused with the Coulomb grid + the Lennard-Jones pair in one VdW launch; 0: Coulomb alone + VdW launches)
case("CIT-7 fixture @ 0.15 A, Na + CO2 (runtests.jl:240-258)", ("Na", "C_co2", "O_co2"), lambda a: W.fixture_workload("CIT-7", a, 0.15))
case("roofline workload, Na + CO2", ("Na", "C_co2", "O_co2"), lambda a: W.roofline_workload(a, 255))
case("CHA fixture @ 0.15 A, CO2 + Ar", ("C_co2", "O_co2", "Ar"), lambda a: W.fixture_workload("CHA_1.4_3b4eeb96", a, 0.15))
case("roofline workload, CO2 + Ar", ("C_co2", "O_co2", "Ar"), lambda a: W.roofline_workload(a, 255))
