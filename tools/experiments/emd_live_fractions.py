"""How many columns / rows of set 2 are still LIVE (remainR, ratioR not exactly +0) at every level of the reference schedule on
one C4 sample: the chain in float32 numpy (matrix form of tf_approxmatch.cu:21-177).  These are the constants of bench.py
EMD_LAUNCH_MIX's round-6 rows and the reason for approxmatch.hip's live-column sweeps (am_compact_kernel, am_p2_live_kernel).
usage: python tools/experiments/emd_live_fractions.py"""
import numpy as np
rng=np.random.RandomState(100)
n=2048
A=(rng.random_sample((32,n,3))-0.5).astype(np.float32)[3]; B=(rng.random_sample((32,n,3))-0.5).astype(np.float32)[3]
D2=((A[:,None,:]-B[None,:,:])**2).sum(-1).astype(np.float32)
levels=[-4.0**j for j in range(7,-2,-1)]+[0.0]
remL=np.ones(n,np.float32); remR=np.ones(n,np.float32)
f32=np.float32
for lv,level in enumerate(levels):
    E=np.exp(f32(level)*D2).astype(np.float32)
    print(f"level {lv} ({level}): before: remainL: ==0 {np.mean(remL==0):.3f} <1e-7 {np.mean(remL<1e-7):.3f} <1e-5 {np.mean(remL<1e-5):.3f} | remainR: ==0 {np.mean(remR==0):.3f} <1e-7 {np.mean(remR<1e-7):.3f} <1e-5 {np.mean(remR<1e-5):.3f}  mass L {remL.sum():.3f} R {remR.sum():.3f}")
    suml=(E@remR+f32(1e-9)).astype(np.float32)
    ratioL=(remL/suml).astype(np.float32)
    sumr=(E.T@ratioL).astype(np.float32)
    s=(sumr*remR).astype(np.float32)
    cons=np.minimum(remR/(s+f32(1e-9)),f32(1)).astype(np.float32)
    ratioR=(remR*cons).astype(np.float32)
    remR=np.maximum(f32(0),remR-s).astype(np.float32)
    sl=(ratioL*(E@ratioR)).astype(np.float32)
    remL=np.maximum(f32(0),remL-sl).astype(np.float32)
    nzL=(ratioL!=0).mean(); nzR=(ratioR!=0).mean()
    print(f"     ratioL nonzero {nzL:.3f}  ratioR nonzero {nzR:.3f}")
