import sys, time
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)),'..','crystalenergygrids.jl_amd')); sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)),'..'))
import numpy as np
import ceg_hip
from ceg_hip import grids as G, workloads as W, _abi
from ceg_hip.plan import GridPlan
from oracle import oracle as O
from oracle.compare import compare_grids
for fwname, atom, sp in (("CIT-7","Ar",0.6), ("CIT-7","Na",0.6), ("CHA_1.4_3b4eeb96","Na",1.0)):
    w = W.fixture_workload(fwname, atom, sp)
    plan = GridPlan(w.cset, w.probe_vdw, w.probe_coulomb, w.alpha)
    print(fwname, atom, "natoms", w.natoms, "npts", w.cset.npoints, "can_cull", plan.can_cull, "images", plan.num_images)
    nx,ny,nz = w.cset.npoints
    ii,jj,kk = np.meshgrid(np.arange(nx),np.arange(ny),np.arange(nz),indexing='ij')
    pts = np.stack([ii*w.cset.size[0]/w.cset.dims[0]+w.cset.shift[0], jj*w.cset.size[1]/w.cset.dims[1]+w.cset.shift[1], kk*w.cset.size[2]/w.cset.dims[2]+w.cset.shift[2]],axis=-1).reshape(-1,3)
    for which, ofn in (("vdw", lambda: O.points_vdw(w.probe_vdw, pts)), ("coulomb", lambda: O.points_coulomb(w.probe_coulomb, w.alpha, pts))):
        t=time.time(); ref = ofn(); to=time.time()-t
        for algo,name in ((1,"brute"),(2,"culled")):
            t=time.time(); got = plan.eval_points(which, pts, algo); tg=time.time()-t
            fin = np.isfinite(ref)
            same_special = np.array_equal(np.isfinite(got), fin) and np.array_equal(got[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)])
            scale = np.median(np.abs(ref[fin]))
            err = np.abs(got[fin]-ref[fin])/(np.abs(ref[fin])+1e-6*scale)
            print(f"  {which:8s} {name:7s} special_ok={same_special} max_rel={err.max():.3e}  oracle {to:.2f}s gpu {tg:.3f}s")
    plan.close()
