"""The per-row certificate of the expanded EMD levels (emd_fgt.hip): true truncation error of the degree-P series against the bound
E(x) = exp(-a|x|^2) (g|x|)^(P+1)/(P+1)! exp(g|x|R_cols) Q on box-filling clouds, corner clusters and mixes.  numpy only.
usage: python tools/experiments/fgt_row_bound.py"""
import numpy as np, math
rng=np.random.RandomState(100)
n=2048
A=(rng.random_sample((n,3))-0.5); B=(rng.random_sample((n,3))-0.5)
def run(A,B,a,P,w,label):
    lo=np.minimum(A.min(0),B.min(0)); hi=np.maximum(A.max(0),B.max(0)); O=0.5*(lo+hi)
    x=A-O; y=B-O
    g=2*a
    rx=np.linalg.norm(x,axis=1); ry=np.linalg.norm(y,axis=1)
    Ra=rx.max(); Rb=ry.max()
    W=w*np.exp(-a*ry**2)
    t=g*x@y.T
    # truncated series
    ser=np.zeros_like(t); term=np.ones_like(t)
    for k in range(P+1):
        ser+=term; term=term*t/(k+1)
    S_true=np.exp(-a*rx**2)*((np.exp(t))@W)
    S_ser=np.exp(-a*rx**2)*(ser@W)
    rel=np.abs(S_ser-S_true)/S_true
    # bound with 4-bin Q
    nb=4
    rho=Ra*(np.arange(nb)+1)/nb
    Q=np.array([(W*ry**(P+1)*np.exp(g*r*ry)).sum() for r in rho])
    j=np.minimum((rx/Ra*nb).astype(int),nb-1)
    # tighter: use actual rx in power, Q bin in exp
    bound=np.exp(-a*rx**2)*(g*rx)**(P+1)/math.factorial(P+1)*Q[j]
    relb=bound/S_ser
    print(f"{label}: gRaRb={g*Ra*Rb:.3f} true rel err max {rel.max():.2e}  bound max {relb.max():.2e} median {np.median(relb):.2e}; frac rows bound>1e-7: {(relb>1e-7).mean():.4f} >3e-8: {(relb>3e-8).mean():.4f} >1e-8 {(relb>1e-8).mean():.4f}")
w=np.ones(n)/n
run(A,B,1.0,10,w,"uniform a=1 P10")
run(A,B,0.25,6,w,"uniform a=.25 P6")
run(A,B,0.25,8,w,"uniform a=.25 P8")
run(A,B,0.25,10,w,"uniform a=.25 P10")
# clustered opposite corners
A2=np.clip(0.45+0.03*rng.randn(n,3),-0.5,0.5); B2=np.clip(-0.45+0.03*rng.randn(n,3),-.5,.5)
run(A2,B2,1.0,10,w,"corners a=1 P10")
run(A2,B2,0.25,6,w,"corners a=.25 P6")
run(A2,B2,0.25,10,w,"corners a=.25 P10")
# partial vs complete: A cluster at corner, B uniform
run(A2,B,1.0,10,w,"corner-vs-uniform a=1 P10")
run(B,A2,1.0,10,w,"uniform-vs-corner a=1 P10")
w2=rng.random_sample(n)**4; w2/=w2.sum()
run(A,B,1.0,10,w2,"uniform skewed w a=1")
print("---- single-moment variant")
def run2(A,B,a,P,w,label):
    lo=np.minimum(A.min(0),B.min(0)); hi=np.maximum(A.max(0),B.max(0)); O=0.5*(lo+hi)
    x=A-O; y=B-O; g=2*a
    rx=np.linalg.norm(x,axis=1); ry=np.linalg.norm(y,axis=1); Rb=ry.max()
    W=w*np.exp(-a*ry**2)
    t=g*x@y.T
    ser=np.zeros_like(t); term=np.ones_like(t)
    for k in range(P+1):
        ser+=term; term=term*t/(k+1)
    S_true=np.exp(-a*rx**2)*((np.exp(t))@W); S_ser=np.exp(-a*rx**2)*(ser@W)
    rel=np.abs(S_ser-S_true)/S_true
    Q0=(W*ry**(P+1)).sum()
    bound=np.exp(-a*rx**2)*(g*rx)**(P+1)/math.factorial(P+1)*np.exp(g*rx*Rb)*Q0
    relb=bound/S_ser
    print(f"{label}: true max {rel.max():.2e} bound max {relb.max():.2e} med {np.median(relb):.2e}; frac>1e-7 {(relb>1e-7).mean():.4f} >3e-8 {(relb>3e-8).mean():.4f}; true err of rows passing 1e-7: {rel[relb<=1e-7].max() if (relb<=1e-7).any() else 0:.2e}")
for nm,(X,Y) in {"uniform":(A,B),"corners":(A2,B2),"corner-vs-uniform":(A2,B),"uniform-vs-corner":(B,A2)}.items():
    run2(X,Y,1.0,10,w,nm+" a=1 P10")
    run2(X,Y,0.25,6,w,nm+" a=.25 P6")
    run2(X,Y,0.25,8,w,nm+" a=.25 P8")
for s in range(5):
    r=np.random.RandomState(s)
    run2(r.random_sample((n,3))-.5, r.random_sample((n,3))-.5,1.0,10,w,f"uniform seed {s}")
