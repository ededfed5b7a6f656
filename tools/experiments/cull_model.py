"""numpy model of the culled Chamfer sweep: how many 16-candidate blocks a wave of
64 sorted queries has to scan / test, and how many superblock steps it takes, for different point
orders (Morton, Hilbert, histogram-equalised Hilbert, sort-tile-recursive) and cloud kinds.  This is
what the block / superblock sizes and the Hilbert + equalisation choice were read from before any
HIP was written.  usage: python tools/experiments/cull_model.py [randn|outlier|sphere]"""
import sys

import numpy as np

from cull_model_orders import hilbert_index, sort_hilbert, sort_str  # noqa: F401
def morton(q):  # q: (n,3) ints < 1024
    def spread(v):
        v = v.astype(np.uint64)
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:,0]) | (spread(q[:,1])<<1) | (spread(q[:,2])<<2)
def sort_cloud(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    q = np.minimum(((p-lo)/(hi-lo+1e-30)*(1<<bits)).astype(np.int64), (1<<bits)-1)
    k = morton(q)
    o = np.argsort(k, kind='stable')
    return p[o]
def sim(Q, C, QW=64, BS=16, SB=4, nwaves=24, rng=None, mode='sb'):
    nb = len(C)//BS
    Cb = C[:nb*BS].reshape(nb, BS, 3)
    blo, bhi = Cb.min(1), Cb.max(1)
    nsb = nb//SB
    slo = blo.reshape(nsb,SB,3).min(1); shi = bhi.reshape(nsb,SB,3).max(1)
    nw = len(Q)//QW
    waves = rng.choice(nw, min(nwaves,nw), replace=False)
    scans=[]; tests=[]; steps=[]
    for w in waves:
        q = Q[w*QW:(w+1)*QW]
        qlo, qhi = q.min(0), q.max(0)
        best = np.full(QW, np.inf); ns=nt=st=0
        lbs = (np.maximum(0, np.maximum(slo-qhi, qlo-shi))**2).sum(1)
        for s in np.argsort(lbs, kind='stable'):
            st+=1
            if lbs[s] > best.max(): break
            for bidx in range(s*SB,(s+1)*SB):
                lbq = (np.maximum(0, np.maximum(blo[bidx]-q, q-bhi[bidx]))**2).sum(1); nt+=1
                if (lbq > best).all(): continue
                d = ((q[:,None,:]-Cb[bidx][None])**2).sum(2).min(1)
                best = np.minimum(best, d); ns+=1
        scans.append(ns); tests.append(nt); steps.append(st)
    return np.mean(scans), np.mean(tests), np.mean(steps)
def cost(s, BS, stepc): return s[0]*BS*7.5 + s[1]*14 + s[2]*stepc
def sort_hilbert_eq(p, bits=5, hb=256):
    lo, hi = p.min(0), p.max(0)
    q = np.zeros(p.shape, np.int64)
    for a in range(3):
        f = np.minimum(((p[:,a]-lo[a])/(hi[a]-lo[a]+1e-30)*hb).astype(np.int64), hb-1)
        h = np.bincount(f, minlength=hb); cdf = np.cumsum(h)-h  # exclusive
        q[:,a] = np.minimum((cdf[f] * (1<<bits)) // len(p), (1<<bits)-1)
    return p[np.argsort(hilbert_index(q,bits), kind='stable')]
if __name__ == '__main__':
    rng = np.random.RandomState(100)
    kind = sys.argv[1] if len(sys.argv) > 1 else 'randn'
    def gen(n):
        if kind=='randn': return rng.randn(n,3).astype(np.float32)
        if kind=='outlier':
            x = rng.randn(n,3).astype(np.float32); x[:4] *= 50; return x
        if kind=='sphere':
            x = rng.randn(n,3); return (x/np.linalg.norm(x,axis=1,keepdims=True)).astype(np.float32)
    A = gen(2048); B = gen(16384); B2 = gen(16384)
    orders = {'hilbert5': lambda p: sort_hilbert(p,5), 'hilbert5eq': lambda p: sort_hilbert_eq(p,5), 'hilbert6eq': lambda p: sort_hilbert_eq(p,6)}
    for name,f in orders.items():
      As,Bs,B2s = f(A),f(B),f(B2)
      for BS,SB in ((16,4),):
          s1 = sim(As,Bs,64,BS,SB,16,rng); s2 = sim(Bs,As,64,BS,SB,32,rng); s3 = sim(Bs,B2s,64,BS,SB,32,rng)
          c1,c2,c3 = cost(s1,BS,28),cost(s2,BS,22),cost(s3,BS,28)
          tot_c2 = (c1*32*32 + c2*256*32)/9.3e11*1e6; tot_ns = 2*c3*256*32/9.3e11*1e6
          print(f'{kind} {name} BS{BS} SB{SB}: A>B sc {s1[0]:.0f} t {s1[1]:.0f} st {s1[2]:.0f} cost {c1:.0f} | B>A sc {s2[0]:.1f} t {s2[1]:.0f} st {s2[2]:.0f} cost {c2:.0f} | B>B sc {s3[0]:.1f} t {s3[1]:.0f} st {s3[2]:.0f} cost {c3:.0f} || est C2 {tot_c2:.0f} us  NS {tot_ns:.0f} us')
