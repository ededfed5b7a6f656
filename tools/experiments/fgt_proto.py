"""One truncated Taylor expansion about the common centre for sum_col w exp(-a |row - col|^2) (emd_fgt.hip): relative error against
the direct fp64 sum, by sharpness a and total degree P, on C4-like clouds.  CPU only."""
import numpy as np, math, itertools
rng=np.random.RandomState(1)
n=m=2048
X=rng.rand(n,3)-0.5; Y=rng.rand(m,3)-0.5; w=rng.rand(n)
O=0.5*(np.minimum(X.min(0),Y.min(0))+np.maximum(X.max(0),Y.max(0)))
Xc=X-O; Yc=Y-O
RA=np.sqrt((Xc**2).sum(1).max()); RB=np.sqrt((Yc**2).sum(1).max())
for a,P in ((1.0,14),(1.0,12),(1.0,10),(0.25,8),(0.25,6),(4.0,14)):
    g=2*a
    D2=((Yc[:,None,:]-Xc[None,:,:])**2).sum(-1)
    S=(np.exp(-a*D2)*w[None,:]).sum(1)
    monos=[(i,j,k) for i in range(P+1) for j in range(P+1-i) for k in range(P+1-i-j)]
    wx=w*np.exp(-a*(Xc**2).sum(1))
    px=[Xc[:,0]**i for i in range(P+1)]; py=[Xc[:,1]**i for i in range(P+1)]; pz=[Xc[:,2]**i for i in range(P+1)]
    qx=[Yc[:,0]**i for i in range(P+1)]; qy=[Yc[:,1]**i for i in range(P+1)]; qz=[Yc[:,2]**i for i in range(P+1)]
    acc=np.zeros(m)
    for (i,j,k) in monos:
        M=(wx*px[i]*py[j]*pz[k]).sum()
        coef=g**(i+j+k)/(math.factorial(i)*math.factorial(j)*math.factorial(k))
        acc+=coef*M*qx[i]*qy[j]*qz[k]
    Sf=np.exp(-a*(Yc**2).sum(1))*acc
    print(f"a={a} P={P} monomials={len(monos)} gamma*RA*RB={g*RA*RB:.3f}: max rel err {np.abs(Sf-S).max()/S.min():.2e} (rel to each: {np.abs(Sf/S-1).max():.2e})")
print("---- higher a")
def test(a,P):
    g=2*a
    D2=((Yc[:,None,:]-Xc[None,:,:])**2).sum(-1)
    S=(np.exp(-a*D2)*w[None,:]).sum(1)
    wx=w*np.exp(-a*(Xc**2).sum(1))
    px=[Xc[:,0]**i for i in range(P+1)]; py=[Xc[:,1]**i for i in range(P+1)]; pz=[Xc[:,2]**i for i in range(P+1)]
    qx=[Yc[:,0]**i for i in range(P+1)]; qy=[Yc[:,1]**i for i in range(P+1)]; qz=[Yc[:,2]**i for i in range(P+1)]
    acc=np.zeros(m); cnt=0
    for i in range(P+1):
        for j in range(P+1-i):
            for k in range(P+1-i-j):
                M=(wx*px[i]*py[j]*pz[k]).sum()
                coef=g**(i+j+k)/(math.factorial(i)*math.factorial(j)*math.factorial(k))
                acc+=coef*M*qx[i]*qy[j]*qz[k]; cnt+=1
    Sf=np.exp(-a*(Yc**2).sum(1))*acc
    print(f"a={a} P={P} monomials={cnt}: max rel err {np.abs(Sf/S-1).max():.2e}")
for a,P in ((4.0,18),(4.0,22),(4.0,26),(16.0,40)):
    test(a,P)
