"""Why degree 7 on 32 intervals per octave for the tabulated Buckingham exponential: interpolation error by (intervals per octave, degree).  CPU only, mpmath."""
import numpy as np, mpmath as mp
mp.mp.dps = 40
A=5.581e7; B=3.985; C=9.167e5
def fit_err(logm, deg, smin=4.0, smax=144.0):
    worst=0; worst_s=0
    # intervals: octave [2^e, 2^(e+1)) split into 2^logm
    e0=int(np.floor(np.log2(smin))); e1=int(np.ceil(np.log2(smax)))
    nd=deg+1
    nodes=[mp.cos(mp.pi*(k+0.5)/nd) for k in range(nd)]
    for e in range(e0,e1):
        for m in range(0, 2**logm, max(1,2**logm//8)):   # sample intervals
            lo=mp.mpf(2)**e*(1+mp.mpf(m)/2**logm); hi=mp.mpf(2)**e*(1+mp.mpf(m+1)/2**logm)
            if hi<=smin or lo>=smax: continue
            mid=(lo+hi)/2; hh=(hi-lo)/2
            f=lambda s: (A/C)*mp.e**(-B*mp.sqrt(s))
            ys=[f(mid+u*hh) for u in nodes]
            # barycentric / use polyfit via mp
            V=mp.matrix(nd,nd)
            for r in range(nd):
                for c in range(nd): V[r,c]=nodes[r]**c
            co=mp.lu_solve(V, mp.matrix(ys))
            for q in range(33):
                u=mp.mpf(-1)+mp.mpf(2*q)/32
                s=mid+u*hh
                pv=sum(co[c]*u**c for c in range(nd))
                ref=f(s)
                scale=abs(ref)+1/(s**3)
                err=abs(pv-ref)/scale
                if err>worst: worst=err; worst_s=s
    return float(worst), float(mp.sqrt(worst_s))
for logm in (5,6,7):
    for deg in (5,6,7,8):
        print(logm,deg,fit_err(logm,deg))
