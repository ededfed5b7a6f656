"""Numpy model of the split-bf16 MFMA filter costed in DESIGN.md 5.1 (not built): error of the K=15
approximate distance against 2^-14 (|a|^2+|b|^2) and the number of survivors per point."""
import numpy as np, sys
def bf16(x):
    x=np.asarray(x,np.float32); u=x.view(np.uint32).astype(np.uint64)
    r=((u + 0x7FFF + ((u>>16)&1)) >> 16).astype(np.uint32) << 16   # RNE
    return r.view(np.float32)
def split2(x):
    h=bf16(x); m=bf16((x-h).astype(np.float32)); return h,m
def approx_d2(a,b):
    # a (n,3), b (m,3); emulate K=15: -2*(ah.bh + ah.bm + am.bh) + na(3 terms) + nb(3 terms); fp32 accumulate
    ah,am=split2(a); bh,bm=split2(b)
    na=(a.astype(np.float64)**2).sum(1).astype(np.float32); nb=(b.astype(np.float64)**2).sum(1).astype(np.float32)
    def split3(v):
        h=bf16(v); m=bf16((v-h).astype(np.float32)); l=bf16((v-h-m).astype(np.float32)); return h,m,l
    nah,nam,nal=split3(na); nbh,nbm,nbl=split3(nb)
    # B side carries -2*b (exact scaling by 2)
    acc=np.zeros((a.shape[0],b.shape[0]),np.float32)
    for (x,y) in ((ah,bh),(ah,bm),(am,bh)):
        for c in range(3):
            acc=(acc + (x[:,c:c+1]*(-2*y[:,c])[None,:]).astype(np.float32)).astype(np.float32)
    for t in (nah,nam,nal): acc=(acc+t[:,None]).astype(np.float32)
    for t in (nbh,nbm,nbl): acc=(acc+t[None,:]).astype(np.float32)
    return acc,na,nb
def exact_d2(a,b):
    dx=(b[None,:,0]-a[:,None,0]).astype(np.float32); dy=(b[None,:,1]-a[:,None,1]).astype(np.float32); dz=(b[None,:,2]-a[:,None,2]).astype(np.float32)
    # fma chain emulated in float64 then rounded (close enough for statistics)
    t=(dy.astype(np.float64)*dy).astype(np.float32); u=(dx.astype(np.float64)*dx+t).astype(np.float32); return (dz.astype(np.float64)*dz+u).astype(np.float32)
for name,gen in (("randn",lambda r,n: r.randn(n,3).astype(np.float32)),("unif",lambda r,n:(r.rand(n,3)-0.5).astype(np.float32))):
  for (n,m) in ((2048,16384),(16384,2048),(4096,4096)):
    r=np.random.RandomState(1); a=gen(r,n); b=gen(r,m)
    ad,na,nb=approx_d2(a,b); ed=exact_d2(a,b)
    eps=(na[:,None]+nb[None,:])*np.float32(2.0**-14)
    viol=(np.abs(ad.astype(np.float64)-ed)>eps).sum()
    ratio=(np.abs(ad.astype(np.float64)-ed)/(na[:,None]+nb[None,:])).max()
    # survivors row direction: approx min per row, thr = m~ + 2*eps_row_max? use per-element eps: survivor if ad - eps <= min_j(ad + eps)
    ub=(ad+eps).min(1); surv=((ad-eps)<=ub[:,None]); 
    ubc=(ad+eps).min(0); survc=((ad-eps)<=ubc[None,:])
    print(name,n,m,"eps violations",viol,"max |err|/(na+nb) = 2^%.1f"%np.log2(ratio),"row survivors/row %.2f max %d"%(surv.sum(1).mean(),surv.sum(1).max()),"col survivors/col %.2f max %d"%(survc.sum(0).mean(),survc.sum(0).max()), "union density %.2e"%((surv|survc).mean()))
    # exactness: argmin among survivors == true argmin
    tr=ed.argmin(1); ok=surv[np.arange(n),tr].all(); tc=ed.argmin(0); okc=survc[tc,np.arange(m)].all()
    print("   true NN always among survivors:",ok,okc)
