import numpy as np, torch, sys
sys.path.insert(0,'.')
from rfnet_amd import _raw as R
rng=np.random.RandomState(0)
for (n,m) in ((300,700),(4096,4096)):
    a=rng.randn(2,n,3).astype(np.float32); c=rng.randn(2,m,3).astype(np.float32)
    a[0,5,1]=np.nan; c[0,0,0]=np.nan; c[1,600,2]=np.nan; a[1,7]=np.inf; c[1,9]=np.inf
    for mode in ("dense","culled"):
        d1,i1,d2,i2=[t.cpu().numpy() for t in R.nn_distance(torch.from_numpy(a).cuda(),torch.from_numpy(c).cuda(),mode=mode)]
        print(n,m,mode,"nan query a[0,5]:",d1[0,5],i1[0,5]," inf query a[1,7]:",d1[1,7],i1[1,7]," nan cand c[0,0] as query:",d2[0,0],i2[0,0], " inf c[1,9]:", d2[1,9], i2[1,9], " any idx1==0 in b0:", (i1[0]==0).sum(), "nan count d1:", np.isnan(d1).sum(), np.isnan(d2).sum())
p=rng.rand(1,500,3).astype(np.float32); q=p[:,:20].copy(); q[0,3,0]=np.nan; p[0,100,1]=np.nan
idx,cnt=R.query_ball_point(0.2,8,torch.from_numpy(p).cuda(),torch.from_numpy(q).cuda())
print("qb nan query row:", idx[0,3].tolist(), cnt[0,3].item(), " row0:", idx[0,0].tolist(), cnt[0,0].item(), "contains 100:", (idx[0]==100).any().item())
