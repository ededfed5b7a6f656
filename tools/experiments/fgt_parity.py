"""Does it matter for parity HOW the row sums of the broad levels are formed?  The 10-level schedule in float32 numpy with the sums of
chosen levels taken exactly (float64) instead of in float32: entries of `match` outside abs 1e-6 + rel 1e-4 of the oracle.  CPU only."""
import sys, numpy as np, time
sys.path.insert(0, __file__.rsplit('/', 3)[0])
from oracle.oracle import Oracle
orc=Oracle()
rng=np.random.RandomState(100)
a=(rng.random_sample((32,2048,3))-0.5).astype(np.float32)
c=(rng.random_sample((32,2048,3))-0.5).astype(np.float32)
x1=a[3]; x2=c[3]
t=time.time(); om=orc.approx_match(x1[None],x2[None])[0]; print("oracle",time.time()-t, om.shape)
n=m=2048
levels=[-4.0**j for j in range(7,-2,-1)]+[0.0]
log2e=np.float32(1.44269502)
d=(x2[:,None,:]-x1[None,:,:]).astype(np.float32)   # [l][k]
D2=(d[...,2]*d[...,2] + (d[...,0]*d[...,0] + d[...,1]*d[...,1])).astype(np.float32)
def run(exact_levels):
    remainL=np.ones(n,np.float32); remainR=np.ones(m,np.float32)
    match=np.zeros((m,n),np.float32)
    for v,lv in enumerate(levels):
        cc=np.float32(lv)*log2e
        if v in exact_levels:
            E64=np.exp2(D2.astype(np.float64)*np.float64(cc))
            E=E64.astype(np.float32)
        else:
            E=np.exp2((D2*cc).astype(np.float32)).astype(np.float32); E64=None
        # P1: suml[k] = 1e-9 + sum_l e[l,k]*remainR[l]
        if E64 is not None: suml=(1e-9+ (E64*remainR[:,None].astype(np.float64)).sum(0)).astype(np.float32)
        else: suml=(np.float32(1e-9)+(E*remainR[:,None]).sum(0,dtype=np.float32)).astype(np.float32)
        ratioL=(remainL/suml).astype(np.float32)
        # P2
        if E64 is not None: sumr=(E64*ratioL[None,:].astype(np.float64)).sum(1).astype(np.float32)
        else: sumr=(E*ratioL[None,:]).sum(1,dtype=np.float32)
        tt=(sumr*remainR).astype(np.float32)
        cons=np.minimum(remainR/(tt+np.float32(1e-9)),np.float32(1)).astype(np.float32)
        ratioR=(remainR*cons).astype(np.float32)
        remainR=np.maximum(np.float32(0),remainR-tt).astype(np.float32)
        # P3
        W=((ratioL[None,:]*E).astype(np.float32)*ratioR[:,None]).astype(np.float32)
        match=(match+W).astype(np.float32)
        if E64 is not None: s3=((ratioL[None,:].astype(np.float64)*E64)*ratioR[:,None].astype(np.float64)).sum(0).astype(np.float32)
        else: s3=W.sum(0,dtype=np.float32)
        remainL=np.maximum(np.float32(0),remainL-s3).astype(np.float32)
    return match
for name,ex in (("fp32 pairwise sums",()),("levels 7,8 exact sums",(7,8)),("levels 6,7,8 exact",(6,7,8)),("all exact sums",tuple(range(10)))):
    mt=run(ex)
    err=np.abs(mt-om); tol=1e-6+1e-4*np.abs(om)
    print(f"{name:28s}: outside strict bar {int((err>tol).sum())} of {om.size}, max abs err {err.max():.3e}")
