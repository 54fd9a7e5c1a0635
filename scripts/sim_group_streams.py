"""VERDICT r2 item 5a, evaluated before building it: would per-half-wave (or per-quarter) candidate streams lift the lane
utilisation of the boundary candidates?  Exact geometry of the roofline workload (11 664 atoms, 256^3 grid, 4x4x4 tiles), 3000
random tiles: for every tile the candidates the kernel keeps are classified interior / boundary exactly like the staging loop does
(nearest and farthest corner of the tile box); a sub-wave stream of a half (2x4x4 points) or quarter (1x4x4) of the tile can only
drop a boundary candidate that is out of reach of the WHOLE half / quarter, and the loop then runs max over the groups of their
list lengths.  Output: boundary iterations relative to today's one list per tile.  CPU only:  python scripts/sim_group_streams.py"""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import numpy as np
from ceg_hip import workloads as W
from scipy.spatial import cKDTree
w=W.roofline_workload("Ar",255)
cs=w.cset
pos=w.probe_coulomb.positions; mat=np.array(w.probe_coulomb.mat)
# lattice images within box grown by cutoff
lo=cs.shift-12.5; hi=cs.shift+cs.size+12.5
imgs=[]
for a in range(-2,3):
  for b in range(-2,3):
    for c in range(-2,3):
      p=pos+mat@np.array([a,b,c])
      m=np.all((p>=lo)&(p<=hi),axis=1)
      imgs.append(p[m])
imgs=np.concatenate(imgs); print(len(imgs),'images')
tree=cKDTree(imgs)
d=cs.size/cs.dims
rng=np.random.default_rng(0)
rc2=144.0; rex2=4.0
res={}
def classify(box_lo,box_hi,P):
    c=0.5*(box_lo+box_hi); h=0.5*(box_hi-box_lo)
    q=np.maximum(0,np.abs(c-P)-h); dmin2=(q*q).sum(1)
    f=np.abs(c-P)+h; dmax2=(f*f).sum(1)
    keep=dmin2<rc2
    interior=keep&(dmin2>rex2)&(dmax2<rc2*(1-1e-9))
    return keep,interior
tot=dict(nb=0,ni=0,half_x=0,half_z=0,quart_x=0,half_y=0,oct=0)
nt=3000
for t in range(nt):
    i0=4*rng.integers(0,64); j0=4*rng.integers(0,64); k0=4*rng.integers(0,64)
    blo=cs.shift+np.array([i0,j0,k0])*d; bhi=blo+3*d
    idx=tree.query_ball_point(0.5*(blo+bhi),12+np.linalg.norm(1.5*d)+0.1)
    P=imgs[idx]
    keep,interior=classify(blo,bhi,P)
    bnd=keep&~interior
    tot['nb']+=bnd.sum(); tot['ni']+=interior.sum()
    Pb=P[bnd]
    def groups(axis,parts):
        ns=[]
        for g in range(parts):
            glo=blo.copy(); ghi=bhi.copy()
            n=4//parts
            glo[axis]=blo[axis]+g*n*d[axis]; ghi[axis]=glo[axis]+(n-1)*d[axis]
            k,_=classify(glo,ghi,Pb)
            ns.append(k.sum())
        return max(ns)
    tot['half_x']+=groups(0,2); tot['half_y']+=groups(1,2); tot['half_z']+=groups(2,2); tot['quart_x']+=groups(0,4)
print({k:v/nt for k,v in tot.items()})
print('boundary iterations relative to full-tile list: half_x %.3f half_y %.3f half_z %.3f quart_x %.3f'%tuple(tot[k]/tot['nb'] for k in ('half_x','half_y','half_z','quart_x')))
