"""exp_fullcov_f16.py -- host emulation of the factor-row (full-covariance) contraction y = R^-1 (x - mu) with both operands as
two fp16 terms (3 products), three bf16 terms (6 products) and plain f32, judged on visible state log-likelihoods against
the double reference, over pools of growing conditioning (Sigma scaled down by `shrink`; kappa = max_g |R^-1 (mu - pivot)|^2 in
the rows' log2 scaling).  The numbers behind FULL_KAPPA_LIMIT_F16 (gmm.h).  No device needed.

    python tools/exp_fullcov_f16.py
"""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from aaltoasr_amd import synth
LOG2E=1.4426950408889634
def split16(x, n=2):
    out=[]; r=x.astype(np.float64)
    for i in range(n):
        h=r.astype(np.float16).astype(np.float64); out.append(h.astype(np.float32)); r=r-h
    return out
def bf16(x):
    u=x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r=((u+0x7FFF+((u>>16)&1))>>16)<<16
    return r.astype(np.uint32).view(np.float32)
def split_bf(x):
    a1=bf16(x); r=(x-a1).astype(np.float32); a2=bf16(r); a3=bf16((r-a2).astype(np.float32)); return a1,a2,a3
D,G,S,comps=13,48,12,4
for shrink in (1.0,1e-1,1e-2,1e-3):
    rng=np.random.default_rng(7)
    mean=rng.standard_normal((G,D))*1.5
    cov=np.empty((G,D,D))
    for g in range(G):
        a=rng.standard_normal((D,D))*0.35
        cov[g]=(a@a.T+0.1*np.eye(D)+np.diag(rng.uniform(0.2,1.0,D)))*shrink
    _,_,off,idx,w=synth.make_model(D=D,G=G,S=S,comps=comps,seed=3)
    pick=rng.integers(0,G,400)
    L=np.linalg.cholesky(cov[pick]); z=rng.standard_normal((400,D)); z*=rng.uniform(2,9,(400,1))/np.linalg.norm(z,axis=1,keepdims=True)
    frames=(mean[pick]+np.einsum("nij,nj->ni",L,z)).astype(np.float32)
    F=len(frames)
    pivot=mean.mean(0)
    Lc=np.linalg.cholesky(cov); W=np.linalg.inv(Lc)*np.sqrt(0.5*LOG2E); b=-np.einsum('gij,gj->gi',W,mean-pivot)
    gc=(-np.log(np.diagonal(Lc,axis1=1,axis2=2)).sum(1))*LOG2E
    A=np.concatenate([W,b[:,:,None]],2).reshape(G*D,D+1)
    X=np.concatenate([frames.astype(np.float64)-pivot,np.ones((F,1))],1)
    def ll_from_y(y): return (gc[:,None]-(y.reshape(G,D,F)**2).sum(1))/LOG2E
    ref=ll_from_y(A@X.T)
    def state(llg):
        llc=llg[idx]+np.log(w)[:,None]
        m=np.maximum.reduceat(llc,off[:-1],axis=0)
        e=np.exp(llc-np.repeat(m,np.diff(off),axis=0))
        return m+np.log(np.add.reduceat(e,off[:-1],axis=0))
    sref=np.maximum(state(ref),np.log(1e-50)); vis=sref>-103.97
    res=[]
    A32=A.astype(np.float32); X32=X.astype(np.float32)
    for name,Ap,Xp,pairs in (("f32",[A32],[X32],[(0,0)]),("bf16x3",split_bf(A32),split_bf(X32),[(0,2),(1,1),(2,0),(0,1),(1,0),(0,0)]),("fp16x2",split16(A32),split16(X32),[(0,1),(1,0),(0,0)])):
        y=np.zeros((G*D,F),np.float32)
        for (i,j) in pairs: y=y+(Ap[i].astype(np.float32)@Xp[j].astype(np.float32).T)
        st=np.maximum(state(ll_from_y(y.astype(np.float64))),np.log(1e-50))
        res.append(np.abs(st-sref)[vis].max())
    print("shrink %.0e visible %d kappa %.0f max|A| %.1f: f32 %.3g bf16x3 %.3g fp16x2 %.3g"%(shrink,vis.sum(),(b**2).sum(1).max(),np.abs(A).max(),*res))
