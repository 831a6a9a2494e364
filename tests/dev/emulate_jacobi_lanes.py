"""Lane-level emulation of jacobi_lds (tsf_newton_kernels.h) in numpy, compared with the oracle's
cn_jacobi bit for bit -- checks the kernel's control flow (odd n, dummy partner, in-place passes)."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import canon_lib as cl
W=64
def bfly(v):
    v=v.copy()
    for off in (1,2,4,8,16,32):
        v = v + v[np.arange(W)^off]
    return v[0]
def emul(A_in):
    n=A_in.shape[0]; PM=n|1; m=n+(n&1)
    Am=np.zeros((PM,PM)); Am[:n,:n]=A_in; Vm=np.zeros((PM,PM))
    lanes=np.arange(W); live=lanes<n
    for i in range(n):
        for l in range(n): Vm[i,l]=1.0 if i==l else 0.0
    for sweep in range(30):
        so=np.zeros(W); sd=np.zeros(W)
        for l in range(n):
            for i in range(n):
                v=Am[i,l]
                if i==l: sd[l]=v*v
                else: so[l]=fma(v,v,so[l])
        off2=bfly(so); dia2=bfly(sd)
        if off2 <= 1e-26*dia2: break
        for r in range(m-1):
            q=np.zeros(W,int); c=np.ones(W); kap=np.zeros(W)
            for l in range(W):
                if l==m-1: ql=r
                elif l==r: ql=m-1
                else:
                    ql=(2*r-l)%(m-1)   # python % is non-negative already
                if live[l] and ql<n:
                    lo,hi=min(l,ql),max(l,ql)
                    apq=Am[lo,hi]
                    if apq!=0.0:
                        tau=(Am[hi,hi]-Am[lo,lo])/(2.0*apq)
                        t=(1.0 if tau>=0 else -1.0)/(abs(tau)+np.sqrt(1.0+tau*tau))
                        c[l]=1.0/np.sqrt(1.0+t*t); s=t*c[l]; kap[l]=-s if l==lo else s
                if not live[l]: ql=l
                q[l]=ql
            for rr in range(n):
                ai=Am[rr,:].copy(); vi=Vm[rr,:].copy()
                for l in range(n):
                    mix = q[l]<n
                    aq = ai[q[l]] if mix else 0.0; vq = vi[q[l]] if mix else 0.0
                    Am[rr,l]=fma(aq,kap[l],ai[l]*c[l]); Vm[rr,l]=fma(vq,kap[l],vi[l]*c[l])
            for i in range(n):
                qi=q[i]
                if qi>=n or qi<i: continue
                bi=Am[i,:].copy(); bq=Am[qi,:].copy()
                for l in range(n):
                    Am[i,l]=fma(kap[i],bq[l],c[i]*bi[l]); Am[qi,l]=fma(kap[qi],bi[l],c[qi]*bq[l])
    return np.array([Am[l,l] for l in range(n)]), Vm[:n,:n].copy()
import math, ctypes
libm=ctypes.CDLL('libm.so.6'); libm.fma.restype=ctypes.c_double; libm.fma.argtypes=[ctypes.c_double]*3
def fma(a,b,c): return libm.fma(float(a),float(b),float(c))
rng=np.random.default_rng(1)
for n in (2,3,5,8,29,33,34):
    M=rng.normal(size=(n,n)); A=M+M.T
    lam,V=emul(A)
    lo,Vo,sw=cl.jacobi_eigh(A)
    print(n,'sweeps',sw,'lam bits equal',np.array_equal(lam,lo),'V bits equal',np.array_equal(V,Vo))
