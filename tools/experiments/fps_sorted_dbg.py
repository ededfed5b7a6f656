import sys, numpy as np, torch
sys.path.insert(0,'.')
from rfnet_amd import _raw as R
rng=np.random.RandomState(5)
x=torch.from_numpy(rng.randint(0,6,size=(2,16384,3)).astype(np.float32)).cuda()
ref=R.farthest_point_sample(300,x); got,nx=R.farthest_point_sample_sorted(300,x,with_xyz=True)
d=(ref!=got).nonzero()
print("mismatches",len(d), d[:5].tolist())
if len(d):
    b,j=d[0].tolist()
    print("first mismatch at",b,j,"ref",ref[b,j].item(),"got",got[b,j].item())
    print("coords ref",x[b,ref[b,j]].tolist(),"got",x[b,got[b,j]].tolist())
    print("ranks: ref k mod 512",ref[b,j].item()%512, "got", got[b,j].item()%512)
print("nx equal", torch.equal(nx, R.gather_point(x, got)))
