import sys, numpy as np, torch
sys.path.insert(0,'.')
from rfnet_amd import _raw as R
for (B,N,M) in ((32,2048,16384),(32,16384,16384)):
    rng=np.random.RandomState(100)
    a=torch.from_numpy(rng.randn(B,N,3).astype(np.float32)).cuda(); c=torch.from_numpy(rng.randn(B,M,3).astype(np.float32)).cuda()
    st=[]; R.nn_distance(a,c,mode="culled",stats=st)
    tot=sum(st[10:14])
    print((B,N,M),"scans by active lanes <=4,<=16,<=32,>32:",[f"{x} ({100*x/max(tot,1):.1f}%)" for x in st[10:14]],"total",tot, st[3]+st[7])
