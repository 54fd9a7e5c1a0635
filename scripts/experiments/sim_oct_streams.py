import os, sys
sys.path[:0] = ['/root/repo/crystalenergygrids.jl_amd', '/root/repo']
import numpy as np
from ceg_hip import workloads as W
w=W.roofline_workload("Ar",255)
cs=w.cset
pos=w.probe_coulomb.positions; mat=np.array(w.probe_coulomb.mat)
cut=12.0; margin=cut*(1+1e-6)+1e-6
lo=cs.shift-margin; hi=cs.shift+cs.size+margin
imgs=[]
for a in range(-2,3):
  for b in range(-2,3):
    for c in range(-2,3):
      p=pos+mat@np.array([a,b,c])
      m=np.all((p>=lo)&(p<=hi),axis=1)
      imgs.append(p[m])
imgs=np.concatenate(imgs)
target=np.array([4.5,4.5,1.5])
nb=np.maximum(1,np.floor((hi-lo)/target).astype(int)); binw=(hi-lo)/nb
b=np.minimum(nb-1,np.floor((imgs-lo)/binw).astype(int))
key=(b[:,0]*nb[1]+b[:,1])*nb[2]+b[:,2]
order=np.argsort(key,kind='stable'); imgs=imgs[order]; key=key[order]
bin_start=np.searchsorted(key,np.arange(nb.prod()+1))
d=cs.size/cs.dims
rng=np.random.default_rng(0)
rc2=144.0*(1+1e-9)+1e-9; rc=np.sqrt(rc2); rex2=4.0
hasv=rng.random(len(imgs))<0.667
def dmin2(blo,bhi,P):
    q=np.maximum(0,np.maximum(blo-P,P-bhi)); return (q*q).sum(1)
def dmax2(blo,bhi,P):
    c=0.5*(blo+bhi); h=0.5*(bhi-blo); f=np.abs(c-P)+h; return (f*f).sum(1)
def binof(x,ax): return int(min(nb[ax]-1,max(0,np.floor((x-lo[ax])/binw[ax]))))
def run(rowmode, nt=400, twoclass=True):
    tot=dict(scanned=0,kept=0,interior=0,bnd=0,it_now=0,it_oct=0,chunks=0,it_oct_ideal=0)
    rng=np.random.default_rng(1)
    for t in range(nt):
        i0=4*rng.integers(0,64); j0=4*rng.integers(0,64); k0=4*rng.integers(0,64)
        blo=cs.shift+np.array([i0,j0,k0])*d; bhi=blo+3*d
        bx0,bx1=binof(blo[0]-rc,0),binof(bhi[0]+rc,0); by0,by1=binof(blo[1]-rc,1),binof(bhi[1]+rc,1)
        rows=[(bx,by) for bx in range(bx0,bx1+1) for by in range(by0,by1+1)]
        n=len(rows)
        if rowmode=='interleave':
            rows=[rows[r>>1] if r%2==0 else rows[n-1-(r>>1)] for r in range(n)]
        elif rowmode=='stride':   # bit-reversal-like: stride permutation
            st=next(s for s in range(int(n*0.38),n) if np.gcd(s,n)==1); rows=[rows[(r*st)%n] for r in range(n)]
        lst=[]
        for bx,by in rows:
            colx0=lo[0]+bx*binw[0]; coly0=lo[1]+by*binw[1]
            gx=max(0,blo[0]-(colx0+binw[0]),colx0-bhi[0]); gy=max(0,blo[1]-(coly0+binw[1]),coly0-bhi[1])
            dxy2=gx*gx+gy*gy
            if dxy2<rc2:
                zr=np.sqrt(rc2-dxy2); bz0,bz1=binof(blo[2]-zr,2),binof(bhi[2]+zr,2)
                rb=(bx*nb[1]+by)*nb[2]
                lst.append(np.arange(bin_start[rb+bz0],bin_start[rb+bz1+1]))
        lst=np.concatenate(lst)
        P=imgs[lst]; hv=hasv[lst] if twoclass else np.ones(len(lst),bool)
        dm=dmin2(blo,bhi,P); keep=dm<rc2; interior=keep&(dm>rex2)&(dmax2(blo,bhi,P)<144*(1-1e-9)); bnd=keep&~interior
        octin=[]
        for gx in range(2):
          for gy in range(2):
            for gz in range(2):
                glo=blo+np.array([gx,gy,gz])*2*d; ghi=glo+d
                octin.append(dmin2(glo,ghi,P)<rc2)
        octin=np.array(octin)&bnd
        tot['scanned']+=len(lst); tot['kept']+=keep.sum(); tot['interior']+=interior.sum(); tot['bnd']+=bnd.sum()
        tot['it_oct_ideal']+=max((octin&hv).sum(1))+max((octin&~hv).sum(1))
        for c0 in range(0,len(lst),64):
            sl=slice(c0,c0+64); tot['chunks']+=1
            tot['it_now']+=bnd[sl].sum()
            tot['it_oct']+=max((octin[:,sl]&hv[sl]).sum(1))+max((octin[:,sl]&~hv[sl]).sum(1))
    return {k:v/nt for k,v in tot.items()}
for mode in ['rowmajor','interleave','stride']:
    print(mode, run(mode))
def run_pool(rowmode, K, nt=300):
    tot=dict(bnd=0,it=0)
    rng=np.random.default_rng(1)
    for t in range(nt):
        i0=4*rng.integers(0,64); j0=4*rng.integers(0,64); k0=4*rng.integers(0,64)
        blo=cs.shift+np.array([i0,j0,k0])*d; bhi=blo+3*d
        bx0,bx1=binof(blo[0]-rc,0),binof(bhi[0]+rc,0); by0,by1=binof(blo[1]-rc,1),binof(bhi[1]+rc,1)
        rows=[(bx,by) for bx in range(bx0,bx1+1) for by in range(by0,by1+1)]
        n=len(rows)
        if rowmode=='interleave':
            rows=[rows[r>>1] if r%2==0 else rows[n-1-(r>>1)] for r in range(n)]
        lst=[]
        for bx,by in rows:
            colx0=lo[0]+bx*binw[0]; coly0=lo[1]+by*binw[1]
            gx=max(0,blo[0]-(colx0+binw[0]),colx0-bhi[0]); gy=max(0,blo[1]-(coly0+binw[1]),coly0-bhi[1])
            dxy2=gx*gx+gy*gy
            if dxy2<rc2:
                zr=np.sqrt(rc2-dxy2); bz0,bz1=binof(blo[2]-zr,2),binof(bhi[2]+zr,2)
                rb=(bx*nb[1]+by)*nb[2]
                lst.append(np.arange(bin_start[rb+bz0],bin_start[rb+bz1+1]))
        lst=np.concatenate(lst)
        P=imgs[lst]; hv=hasv[lst]
        dm=dmin2(blo,bhi,P); keep=dm<rc2; interior=keep&(dm>rex2)&(dmax2(blo,bhi,P)<144*(1-1e-9)); bnd=keep&~interior
        Pb=P[bnd]; hvb=hv[bnd]
        octin=[]
        for gx in range(2):
          for gy in range(2):
            for gz in range(2):
                glo=blo+np.array([gx,gy,gz])*2*d; ghi=glo+d
                octin.append(dmin2(glo,ghi,Pb)<rc2)
        octin=np.array(octin)
        tot['bnd']+=len(Pb)
        for c0 in range(0,len(Pb),K):
            sl=slice(c0,c0+K)
            tot['it']+=max((octin[:,sl]&hvb[sl]).sum(1))+max((octin[:,sl]&~hvb[sl]).sum(1))
    return {k:v/nt for k,v in tot.items()}
for K in (32,64,128):
    for mode in ('rowmajor','interleave'):
        print('pool',K,mode,run_pool(mode,K))
